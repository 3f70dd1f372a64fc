// pt — command-line driver, the C++ stand-in for pt/src/main.rs:25-51: for every scene name on the command
// line: parse scenes/<name>/scene_v3.xml -> build scene -> render on the GPU through the C ABI -> time it ->
// save zout/<output>. Extra flags (not in the reference) override the XML <default> values.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pthost.h"

int main(int argc, char** argv) {
    std::string overrides, collection = "scenes";
    std::vector<std::string> scenes;
    int device = 0;
    bool pfm = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto take = [&](const char* key) {
            if (i + 1 >= argc) {
                fprintf(stderr, "missing value for %s\n", a.c_str());
                exit(2);
            }
            overrides += std::string(key) + "=" + argv[++i] + ";";
        };
        if (a == "--spp") take("spp");
        else if (a == "--resx") take("resx");
        else if (a == "--resy") take("resy");
        else if (a == "--max_depth") take("max_depth");
        else if (a == "--integrator") take("integrator");
        else if (a == "--scenes" && i + 1 < argc) collection = argv[++i];
        else if (a == "--device" && i + 1 < argc) device = atoi(argv[++i]);
        else if (a == "--pfm") pfm = true;
        else scenes.push_back(a);
    }
    if (scenes.empty()) {
        fprintf(stderr, "usage: pt [--spp N --resx N --resy N --max_depth N --integrator path|lightpath|guided --scenes DIR --device N --pfm] <scene>...\n");
        return 2;
    }
    ptgpu_ctx* ctx = nullptr;
    if (ptgpu_init(&ctx, device) != PTGPU_OK) {
        fprintf(stderr, "ptgpu_init: %s\n", ptgpu_last_error());
        return 1;
    }
    int rc = 0;
    for (auto& name : scenes) {
        pthost_scene* sc = pthost_scene_load(collection.c_str(), name.c_str(), overrides.empty() ? nullptr : overrides.c_str(), nullptr);
        if (!sc) {
            fprintf(stderr, "%s: %s\n", name.c_str(), pthost_last_error());
            rc = 1;
            continue;
        }
        const ptgpu_camera_desc* cam = pthost_camera_desc(sc);
        const ptgpu_integrator_desc* integ = pthost_integrator_desc(sc);
        if (ptgpu_scene_create(ctx, pthost_scene_desc(sc)) != PTGPU_OK || ptgpu_camera_set(ctx, cam) != PTGPU_OK) {
            fprintf(stderr, "%s: %s\n", name.c_str(), ptgpu_last_error());
            pthost_scene_free(sc);
            rc = 1;
            continue;
        }
        std::vector<double> film((size_t)cam->resx * cam->resy * 3);
        double scale = 1.0;
        ptgpu_stats st;
        const auto t0 = std::chrono::steady_clock::now();
        const int r = ptgpu_render(ctx, integ, pthost_spp(sc), film.data(), &scale, &st);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (r == PTGPU_W_NONFINITE) fprintf(stderr, "%s: warning: %s\n", name.c_str(), ptgpu_last_error());  // (the reference would have panicked)
        if (r < 0) {
            fprintf(stderr, "%s: render: %s\n", name.c_str(), ptgpu_last_error());
            rc = 1;
        } else {
            const char* out = pthost_output_name(sc);
            // dbg!(start_time.elapsed(), output_name) (main.rs:42-44)
            fprintf(stderr, "[pt] elapsed = %.3fs, output_name = \"%s\", %.2f Msamples/s, %.2f Mrays/s\n", secs, out, st.samples / secs * 1e-6,
                    (st.extend_rays + st.shadow_rays) / secs * 1e-6);
            pthost_save_image(film.data(), cam->resx, cam->resy, scale, integ->kind == PTGPU_INTEGRATOR_LIGHTPATH, "zout", out);
            if (pfm) pthost_save_pfm(film.data(), cam->resx, cam->resy, scale, (std::string("zout/") + out + ".pfm").c_str());
        }
        pthost_scene_free(sc);
    }
    ptgpu_destroy(ctx);
    return rc;
}
