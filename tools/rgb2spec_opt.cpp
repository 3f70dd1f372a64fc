// rgb2spec_opt — regenerates the coefficient table the reference embeds as pt/src/spectrum/srgb.coeffs (missing from its
// tree: .MISSING_LARGE_BLOBS) for rgb2spec.rs:8-54 to read. Build container only; the GPU box never needs it.
//
// Jakob & Hanika 2019: for every cell of a res^3 grid per dominant component l — rgb[l] = z = scale[k],
// rgb[(l+1)%3] = x z, rgb[(l+2)%3] = y z with x = i/(res-1), y = j/(res-1), scale[k] = smoothstep(smoothstep(k/(res-1)))
// — the three sigmoid-polynomial coefficients whose colour under D65 is that rgb (Gauss-Newton on the CIELAB residual,
// path-tracer_amd/host/rgb2spec_fit.h), walked along k from res/5 upwards and downwards with the previous cell's solution
// as the starting point, and re-expressed for wavelengths in nm.
//
// Layout written with --full (what rgb2spec.rs:9-10 slices): 4 bytes "SPEC", u32 res, f32 scale[res],
// f32 data[3][res][res][res][3] (l, k, j, i, coefficient): 9 437 448 bytes at res 64.
// Layout written with --slab (path-tracer_amd/data/srgb_z05.coeffs, read by pthost.cpp): "SPZS", u32 res, u32 first k,
// u32 number of k, f32 scale[res], f32 data[3][n_k][res][res][3] — the k-slices around z = 0.5 only:
// TristimulusSpectrum::new divides every colour by twice its largest component (spectrum/mod.rs:285-286), so the
// lookup's z is always exactly 0.5 and touches slices k = 31 and 32 of the 64 and no others.
//
// Usage: rgb2spec_opt [--res 64] [--full out.coeffs] [--slab out_z05.coeffs] [--threads N]
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../path-tracer_amd/host/rgb2spec_fit.h"

static double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }

int main(int argc, char** argv) {
    int res = 64, n_threads = (int)std::thread::hardware_concurrency();
    std::string full, slab;
    for (int a = 1; a < argc; a++) {
        const std::string k = argv[a];
        if (k == "--res" && a + 1 < argc) res = atoi(argv[++a]);
        else if (k == "--full" && a + 1 < argc) full = argv[++a];
        else if (k == "--slab" && a + 1 < argc) slab = argv[++a];
        else if (k == "--threads" && a + 1 < argc) n_threads = atoi(argv[++a]);
        else {
            fprintf(stderr, "usage: rgb2spec_opt [--res 64] [--full out.coeffs] [--slab out_z05.coeffs] [--threads N]\n");
            return 2;
        }
    }
    if (res < 4 || res > 256 || (full.empty() && slab.empty())) return 2;
    if (n_threads < 1) n_threads = 1;
    const rgb2spec::Rgb2SpecFitter fitter;
    std::vector<float> scale(res);
    for (int k = 0; k < res; k++) scale[k] = (float)smoothstep(smoothstep((double)k / (double)(res - 1)));
    std::vector<float> data((size_t)3 * res * res * res * 3);
    std::atomic<int> next{0}, failed{0};
    const double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
    auto work = [&]() {
        for (;;) {
            const int row = next.fetch_add(1);
            if (row >= 3 * res) break;
            const int l = row / res, j = row % res;
            const double y = (double)j / (double)(res - 1);
            for (int i = 0; i < res; i++) {
                const double x = (double)i / (double)(res - 1);
                const int start = res / 5;
                for (int dir = 0; dir < 2; dir++) {
                    double c[3] = {0.0, 0.0, 0.0};
                    for (int k = start; dir == 0 ? k < res : k >= 0; k += dir == 0 ? 1 : -1) {
                        const double b = (double)scale[k];
                        double rgb[3];
                        rgb[l] = b;
                        rgb[(l + 1) % 3] = x * b;
                        rgb[(l + 2) % 3] = y * b;
                        if (!fitter.gauss_newton(rgb, c)) failed++;
                        const double A = c[0], B = c[1], C = c[2];
                        float* o = &data[((((size_t)l * res + k) * res + j) * res + i) * 3];
                        o[0] = (float)(A * (c1 * c1));
                        o[1] = (float)(B * c1 - 2.0 * A * c0 * (c1 * c1));
                        o[2] = (float)(C - B * c0 * c1 + A * (c0 * c1) * (c0 * c1));
                    }
                }
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto& t : th) t.join();
    fprintf(stderr, "rgb2spec_opt: res %d, %d of %zu fits did not reach the tolerance\n", res, failed.load(), (size_t)3 * res * res * (res + res / 5 + 1));
    const uint32_t res_u = (uint32_t)res;
    if (!full.empty()) {
        FILE* f = fopen(full.c_str(), "wb");
        if (!f) return 1;
        fwrite("SPEC", 4, 1, f);
        fwrite(&res_u, 4, 1, f);
        fwrite(scale.data(), 4, scale.size(), f);
        fwrite(data.data(), 4, data.size(), f);
        fclose(f);
    }
    if (!slab.empty()) {
        // the interval find_interval(scale, 0.5) returns, and its upper neighbour
        int zi = 0;
        for (int k = 0; k + 1 < res; k++)
            if (scale[k] <= 0.5f) zi = k;
        if (zi > res - 2) zi = res - 2;
        const uint32_t first = (uint32_t)zi, count = 2;
        FILE* f = fopen(slab.c_str(), "wb");
        if (!f) return 1;
        fwrite("SPZS", 4, 1, f);
        fwrite(&res_u, 4, 1, f);
        fwrite(&first, 4, 1, f);
        fwrite(&count, 4, 1, f);
        fwrite(scale.data(), 4, scale.size(), f);
        for (int l = 0; l < 3; l++)
            for (uint32_t k = first; k < first + count; k++) fwrite(&data[(((size_t)l * res + k) * res) * res * 3], 4, (size_t)res * res * 3, f);
        fclose(f);
    }
    return 0;
}
