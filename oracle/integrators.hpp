// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header).
// CPU restatement of pt/src/integrator/{mod,path_tracer,light_path_tracer}.rs and path_guiding/mod.rs.
// Tile-parallel like the reference (16x16 tiles, camera.rs:152,167-183): one task per tile on a
// std::thread pool. Film accumulation: tiles own their pixels; splats use a per-film mutex-free
// atomic-free scheme only when n_threads==1, otherwise a striped lock (order-dependent f64 rounding
// is the reference's behaviour too).
#pragma once
#include <atomic>
#include <mutex>
#include <thread>

#include "scene.hpp"
#include "sdtree.hpp"

namespace orc {

enum class Mode : int32_t { PT = 0, NEE = 1, MIS = 2 };
enum class IntegratorKind : int32_t { LightPath = 0, Path = 1, Guided = 2 };

struct IntegratorDesc {
    IntegratorKind kind = IntegratorKind::Path;
    size_t max_depth = 16;
    Mode mode = Mode::MIS;
    float spatial_threshold = 1000.0f;
    float directional_threshold = 0.1f;  // parsed but ignored: render() hard-codes 0.01 (mod.rs:73)
    Filter spatial_filter = Filter::Stochastic, directional_filter = Filter::Box;
};

struct RayStats {
    std::atomic<uint64_t> extend_rays{0}, shadow_rays{0}, nodes{0}, tris{0}, segments{0};
};

struct Tile {
    size_t min[2], size[2];
};
inline std::vector<Tile> create_tiles(size_t w, size_t h, size_t ts = 16) {  // camera.rs:167-183
    std::vector<Tile> tiles;
    for (size_t y = 0; y < (h + ts - 1) / ts; y++)
        for (size_t x = 0; x < (w + ts - 1) / ts; x++) {
            Tile t;
            t.min[0] = x * ts;
            t.min[1] = y * ts;
            t.size[0] = std::min(ts, w - t.min[0]);
            t.size[1] = std::min(ts, h - t.min[1]);
            tiles.push_back(t);
        }
    return tiles;
}

template <class F>
inline void parallel_tiles(const std::vector<Tile>& tiles, int n_threads, F f) {
    if (n_threads <= 1) {
        for (size_t i = 0; i < tiles.size(); i++) f(tiles[i], i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                size_t i = next.fetch_add(1);
                if (i >= tiles.size()) break;
                f(tiles[i], i);
            }
        });
    for (auto& t : th) t.join();
}

// Samples that are not finite are dropped and counted on both sides (the reference asserts on them in `path`,
// path_tracer.rs:46, and would poison the pixel in the other integrators): see pt_kernels.h, film_add_rgb_wave.
inline std::atomic<uint64_t>& nonfinite_samples() {
    static std::atomic<uint64_t> n{0};
    return n;
}
inline bool rgb_finite_or_count(const double rgb[3]) {
    if (std::isfinite(rgb[0]) && std::isfinite(rgb[1]) && std::isfinite(rgb[2])) return true;
    nonfinite_samples()++;
    return false;
}
inline void film_add_pixel(Film& film, size_t ix, size_t iy, const SSpec& l, const SWL& swl) {  // camera.rs:138-140
    double rgb[3];
    spectrum_to_rgb(l, swl, rgb);
    if (!rgb_finite_or_count(rgb)) return;
    double* p = &film.pixels[3 * (ix + iy * film.w)];
    p[0] += rgb[0];
    p[1] += rgb[1];
    p[2] += rgb[2];
}
struct SplatLocks {
    std::mutex m[256];
};
inline void film_add_splat(Film& film, SplatLocks& locks, V2 p_raster, const SSpec& l, const SWL& swl) {  // camera.rs:161-166
    size_t ix = std::min(f32_as_usize(p_raster.x), film.w - 1);
    size_t iy = std::min(f32_as_usize(p_raster.y), film.h - 1);
    size_t idx = ix + iy * film.w;
    double rgb[3];
    spectrum_to_rgb(l, swl, rgb);
    if (!rgb_finite_or_count(rgb)) return;
    std::lock_guard<std::mutex> g(locks.m[idx & 255]);
    double* p = &film.splats[3 * idx];
    p[0] += rgb[0];
    p[1] += rgb[1];
    p[2] += rgb[2];
}

// ------------------------------------------------------------- path_tracer.rs:54-183
inline SSpec path_li(const Scene& scene, const IntegratorDesc& it, Ray ray, SWL& swl, Sampler& sampler, RayStats* st) {
    SSpec l = SSpec::zero(), beta = SSpec::one();
    bool specular_bounce = false;
    SceneInteraction prev{};
    float prev_bsdf_pdf = 0.0f;
    size_t depth = 0;
    while (depth < it.max_depth) {
        SceneInteraction in;
        if (st) st->extend_rays++;
        if (!scene.intersect(ray, &in)) break;  // envmap absent -> ZERO (scene.rs:301-306)
        V3 wo = -ray.d;
        BSDF bsdf;
        if (!scene.compute_bsdf(in, swl, &bsdf)) {
            ray = in.spawn_ray(ray.d);
            specular_bounce = true;
            continue;
        }
        if (st) st->segments++;
        bool light = scene.is_light(in);
        switch (it.mode) {
            case Mode::PT:
                if (light) l = l + scene.get_light(in).emitted(in.p(), ray.o, swl) * beta;
                break;
            case Mode::NEE:
                if ((depth == 0 || specular_bounce) && light) l = l + scene.get_light(in).emitted(in.p(), ray.o, swl) * beta;
                break;
            case Mode::MIS:
                if (light) {
                    AreaLight al = scene.get_light(in);
                    if (depth == 0 || specular_bounce) l = l + al.emitted(in.p(), ray.o, swl) * beta;
                    else {
                        float pb = prev_bsdf_pdf;
                        float pl = al.pdf_li_solid_angle(prev.p(), in.p());
                        float w = (pb * pb) / (pb * pb + pl * pl);
                        l = l + w * al.emitted(in.p(), ray.o, swl) * beta;
                    }
                }
                break;
        }
        if (it.mode != Mode::PT && !bsdf.is_specular()) {
            float u1 = sampler.sample1d();
            float u2[2];
            sampler.sample2d(u2);
            LiSample ls;
            float pmf;
            if (scene.sample_li(in.p(), swl, u1, u2, &ls, &pmf)) {
                SSpec f = bsdf.eval(in.bsdf_eval_ctx(), wo, ls.wl, TransportMode::Radiance) / (ls.pdf * pmf);
                if (!f.is_black()) {
                    if (st) st->shadow_rays++;
                    if (!scene.occluded_interaction(in, ls.p())) {
                        if (it.mode == Mode::NEE) l = l + f * beta * ls.l;
                        else {
                            float pl = ls.pdf;
                            float pb = bsdf.pdf(in.bsdf_eval_ctx(), wo, ls.wl);
                            float w = (pl * pl) / (pl * pl + pb * pb);
                            l = l + w * f * beta * ls.l;
                        }
                    }
                }
            }
        }
        float u1 = sampler.sample1d();
        float u2[2];
        sampler.sample2d(u2);
        BSDFSample s;
        if (!bsdf.sample(in.bsdf_eval_ctx(), wo, u1, u2, TransportMode::Radiance, &s)) break;
        beta = beta * (s.f / s.pdf);
        if (beta.is_black()) break;
        specular_bounce = s.specular;
        if (s.terminate_secondary) swl.terminate_secondary();
        ray = in.spawn_ray(s.wi);
        prev = in;
        prev_bsdf_pdf = s.pdf;
        depth += 1;
    }
    return l;
}

// One (pixel, sample) of PathIntegrator::render (path_tracer.rs:40-49). `arlds_u` = the pixel's i-th
// wavelength sample (the reference carries the ARLDS across the samples of a pixel).
inline void path_render(const Scene& scene, const Camera& cam, const IntegratorDesc& it, size_t spp, size_t sample_begin,
                        size_t sample_step, Film& film, int n_threads, RayStats* st) {
    auto tiles = create_tiles(film.w, film.h);
    parallel_tiles(tiles, n_threads, [&](const Tile& tile, size_t) {
        Sampler sampler;
        for (size_t oy = 0; oy < tile.size[1]; oy++)
            for (size_t ox = 0; ox < tile.size[0]; ox++) {
                size_t ix = tile.min[0] + ox, iy = tile.min[1] + oy;
                for (size_t i = 0; i < spp; i++) {
                    sampler.init_pixel(ix, iy, i);
                    float wu = sampler.wavelength_sample();  // sequential ARLDS: advanced for EVERY i, even skipped shards
                    if (i % sample_step != sample_begin) continue;
                    SWL swl = SWL::sample_visible(wu);
                    float u[2];
                    sampler.sample2d(u);
                    Ray ray = cam.sample_we(ix, iy, u);
                    SSpec l = path_li(scene, it, ray, swl, sampler, st);
                    if (!l.is_black()) film_add_pixel(film, ix, iy, l, swl);
                }
            }
    });
}

// ------------------------------------------------------------- light_path_tracer.rs:21-119
inline void lightpath_render(const Scene& scene, const Camera& cam, const IntegratorDesc& it, size_t spp, size_t sample_begin,
                             size_t sample_step, Film& film, int n_threads, RayStats* st) {
    auto tiles = create_tiles(film.w, film.h);
    SplatLocks locks;
    parallel_tiles(tiles, n_threads, [&](const Tile& tile, size_t) {
        Sampler sampler;
        for (size_t oy = 0; oy < tile.size[1]; oy++)
            for (size_t ox = 0; ox < tile.size[0]; ox++) {
                size_t ix = tile.min[0] + ox, iy = tile.min[1] + oy;
                for (size_t i = 0; i < spp; i++) {
                    sampler.init_pixel(ix, iy, i);
                    float wu = sampler.wavelength_sample();
                    if (i % sample_step != sample_begin) continue;
                    SWL swl = SWL::sample_visible(wu);
                    float u1 = sampler.sample1d();
                    float ua[2], ub[2];
                    sampler.sample2d(ua);
                    sampler.sample2d(ub);
                    LeSample le;
                    float pmf;
                    if (!scene.sample_le(swl, u1, ua, ub, &le, &pmf)) continue;
                    CameraWiSample wi;
                    if (cam.sample_wi(le.p(), &wi)) {
                        SSpec f = wi.importance * le.l * std::fabs(dot(le.n, wi.wc)) / (wi.pdf * le.pdf_pos * pmf);
                        if (!f.is_black()) {
                            if (st) st->shadow_rays++;
                            if (!scene.occluded_le_sample(le, cam.p())) film_add_splat(film, locks, wi.p_raster, f, swl);
                        }
                    }
                    SSpec beta = le.l * std::fabs(dot(le.n, le.dir)) / (le.pdf_pos * le.pdf_dir * pmf);
                    Ray ray = le.spawn_ray();
                    for (size_t depth = 0; depth < it.max_depth; depth++) {  // random_walk
                        SceneInteraction in;
                        if (st) st->extend_rays++;
                        if (!scene.intersect(ray, &in)) break;
                        V3 wo = -ray.d;
                        BSDF bsdf;
                        if (!scene.compute_bsdf(in, swl, &bsdf)) {
                            ray = in.spawn_ray(ray.d);
                            continue;
                        }
                        if (st) st->segments++;
                        CameraWiSample cw;
                        if (!bsdf.is_specular() && cam.sample_wi(in.p(), &cw)) {
                            SSpec f = bsdf.eval(in.bsdf_eval_ctx(), wo, cw.wc, TransportMode::Importance) * cw.importance * beta / cw.pdf;
                            if (!f.is_black()) {
                                if (st) st->shadow_rays++;
                                if (!scene.occluded_interaction(in, cam.p())) film_add_splat(film, locks, cw.p_raster, f, swl);
                            }
                        }
                        float s1 = sampler.sample1d();
                        float s2[2];
                        sampler.sample2d(s2);
                        BSDFSample s;
                        if (!bsdf.sample(in.bsdf_eval_ctx(), wo, s1, s2, TransportMode::Importance, &s)) break;
                        beta = beta * (s.f / s.pdf);
                        if (beta.is_black()) break;
                        if (s.terminate_secondary) swl.terminate_secondary();
                        ray = in.spawn_ray(s.wi);
                    }
                }
            }
    });
}

// ------------------------------------------------------------- path_guiding/mod.rs
inline std::vector<size_t> wave_sample_counts(size_t total) {  // mod.rs:386-402
    size_t added = 0, wave = 1;
    std::vector<size_t> counts;
    for (;;) {
        counts.push_back(wave);
        added += wave;
        size_t remaining = total - added;
        wave *= 2;
        if (remaining < wave) {
            counts.back() += remaining;
            break;
        }
    }
    return counts;
}

struct PathVertex {  // mod.rs:281-289
    V3 p, wi;
    float wi_pdf;
    SSpec f, emitted, li;
    bool is_specular;
};

struct Guider {  // mod.rs:291-384
    STree stree;
    float directional_threshold, spatial_threshold;
    size_t iteration = 0;
    bool has_started = false;
    bool sample(const BSDF& bsdf, const BSDFEvalCtx& ctx, V3 p, V3 wo, Sampler& sampler, TransportMode mode, BSDFSample* out) const {
        if (iteration == 0 || bsdf.is_specular()) {
            float u1 = sampler.sample1d();
            float u2[2];
            sampler.sample2d(u2);
            return bsdf.sample(ctx, wo, u1, u2, mode, out);
        }
        const float p_guided = 0.5f, p_bsdf = 1.0f - p_guided;
        const DTree& dtree = stree.sampler(p);
        if (sampler.sample1d() < p_guided) {
            V2 sq = dtree.sample(sampler);
            V3 wi = square_to_dir(sq);
            float bsdf_pdf = bsdf.pdf(ctx, wo, wi) * p_bsdf;
            if (bsdf_pdf == 0.0f) return false;
            out->pdf = bsdf_pdf + dtree.pdf(sq) * p_guided;
            out->f = bsdf.eval(ctx, wo, wi, mode);
            out->wi = wi;
            out->specular = false;
            out->terminate_secondary = false;
            return true;
        }
        float u1 = sampler.sample1d();
        float u2[2];
        sampler.sample2d(u2);
        if (!bsdf.sample(ctx, wo, u1, u2, mode, out)) return false;
        out->pdf = p_bsdf * out->pdf + p_guided * dtree.pdf(dir_to_square(out->wi));
        return true;
    }
    void new_iteration() {
        if (has_started) {
            stree.new_iteration(directional_threshold, spatial_threshold, iteration);
            iteration += 1;
        } else has_started = true;
    }
};

inline void guided_random_walk(const Scene& scene, const IntegratorDesc& it, Ray ray, SWL& swl, std::vector<PathVertex>& path,
                               const Guider& guider, Sampler& sampler, RayStats* st) {  // mod.rs:124-279
    SSpec beta = SSpec::one();
    size_t depth = 0;
    bool specular_bounce = false;
    SceneInteraction prev{};
    float prev_bsdf_pdf = 0.0f;
    while (depth < it.max_depth) {
        SceneInteraction in;
        if (st) st->extend_rays++;
        if (!scene.intersect(ray, &in)) {
            PathVertex v;
            v.p = ray.o + ray.d * 10000.0f;
            v.wi = ray.d;
            v.wi_pdf = 1.0f;
            v.f = SSpec::one();
            v.emitted = SSpec::zero();
            v.is_specular = false;
            v.li = SSpec::zero();
            path.push_back(v);
            break;
        }
        V3 wo = -ray.d;
        BSDF bsdf;
        if (!scene.compute_bsdf(in, swl, &bsdf)) {
            ray = in.spawn_ray(ray.d);
            continue;
        }
        if (st) st->segments++;
        bool light = scene.is_light(in);
        SSpec emitted = SSpec::zero();
        switch (it.mode) {
            case Mode::PT:
                if (light) emitted = scene.get_light(in).emitted(in.p(), ray.o, swl);
                break;
            case Mode::NEE:
                if ((depth == 0 || specular_bounce) && light) emitted = scene.get_light(in).emitted(in.p(), ray.o, swl);
                break;
            case Mode::MIS:
                if (light) {
                    AreaLight al = scene.get_light(in);
                    if (depth == 0 || specular_bounce) emitted = al.emitted(in.p(), ray.o, swl);
                    else {
                        float pb = prev_bsdf_pdf;
                        float pl = al.pdf_li_solid_angle(prev.p(), in.p());
                        float w = (pb * pb) / (pb * pb + pl * pl);
                        emitted = w * al.emitted(in.p(), ray.o, swl);
                    }
                }
                break;
        }
        SSpec contribution = SSpec::zero();
        if (it.mode != Mode::PT && !bsdf.is_specular()) {
            float u1 = sampler.sample1d();
            float u2[2];
            sampler.sample2d(u2);
            LiSample ls;
            float pmf;
            if (scene.sample_li(in.p(), swl, u1, u2, &ls, &pmf)) {
                SSpec f = bsdf.eval(in.bsdf_eval_ctx(), wo, ls.wl, TransportMode::Radiance) / (ls.pdf * pmf);
                if (!f.is_black()) {
                    if (st) st->shadow_rays++;
                    if (!scene.occluded_interaction(in, ls.p())) {
                        if (it.mode == Mode::NEE) contribution = f * ls.l;
                        else {
                            float pl = ls.pdf;
                            float pb = bsdf.pdf(in.bsdf_eval_ctx(), wo, ls.wl);
                            float w = (pl * pl) / (pl * pl + pb * pb);
                            contribution = w * f * ls.l;
                        }
                    }
                }
            }
        }
        BSDFSample s;
        if (!guider.sample(bsdf, in.bsdf_eval_ctx(), in.p(), wo, sampler, TransportMode::Radiance, &s)) break;
        specular_bounce = bsdf.is_specular();
        if (!bsdf.is_specular()) {
            PathVertex v;
            v.p = in.p();
            v.wi_pdf = s.pdf;
            v.wi = s.wi;
            v.f = s.f / s.pdf;
            v.is_specular = false;
            v.emitted = emitted;
            v.li = contribution;
            path.push_back(v);
        }
        beta = beta * (s.f / s.pdf);
        if (beta.is_black()) break;
        if (s.terminate_secondary) swl.terminate_secondary();
        ray = in.spawn_ray(s.wi);
        prev = in;
        prev_bsdf_pdf = s.pdf;
        depth += 1;
    }
}

struct GuidedReport {
    size_t stree_nodes = 0, stree_leaves = 0, dtree_nodes_total = 0;
};

// The sample passes [first_index, first_index + wave_size) of one wave (mod.rs:84-118): every pass visits every pixel
// once, tile-parallel, records each path back to front into the collectors and adds it to the film.
// shard_index / shard_count: only the tiles t of pass i with (t + i) % shard_count == shard_index are traced (the product's
// multi-GPU rule, pt_guided.h: k_raygen_guided; the reference itself is one process: 0 / 1).
inline void guided_wave_passes(const Scene& scene, const Camera& cam, const IntegratorDesc& it, Guider& guider, Film& film, size_t first_index,
                               size_t wave_size, int n_threads, RayStats* st, size_t shard_index = 0, size_t shard_count = 1) {
    auto tiles = create_tiles(film.w, film.h);
    for (size_t k = 0; k < wave_size; k++) {
        size_t i = first_index + k;
        parallel_tiles(tiles, n_threads, [&](const Tile& tile, size_t tile_index) {
            if ((tile_index + i) % shard_count != shard_index) return;
            std::vector<PathVertex> path;
            Sampler sampler;
            for (size_t oy = 0; oy < tile.size[1]; oy++)
                for (size_t ox = 0; ox < tile.size[0]; ox++) {
                    size_t ix = tile.min[0] + ox, iy = tile.min[1] + oy;
                    SWL swl = SWL::sample_visible(sampler.wavelength_sample());  // BEFORE init_pixel (mod.rs:92-94)
                    sampler.init_pixel(ix, iy, i);
                    float u[2];
                    sampler.sample2d(u);
                    Ray ray = cam.sample_we(ix, iy, u);
                    path.clear();
                    guided_random_walk(scene, it, ray, swl, path, guider, sampler, st);
                    SSpec l = SSpec::zero();
                    for (size_t vi = path.size(); vi-- > 0;) {
                        const PathVertex& v = path[vi];
                        if (!v.is_specular) {
                            float r = l.average() / v.wi_pdf;
                            record_key() = RecordKey{(uint32_t)(ix + iy * film.w), (uint32_t)i, (uint32_t)vi};
                            guider.stree.record(v.p, v.wi, r, sampler);
                        }
                        l = l * v.f;
                        l = l + v.li;
                        l = l + v.emitted;
                    }
                    film_add_pixel(film, ix, iy, l, swl);
                }
        });
    }
}

// PathGuidingIntegrator::render (mod.rs:70-122).
inline void guided_render(const Scene& scene, const Camera& cam, const IntegratorDesc& it, size_t spp, Film& film, int n_threads,
                          RayStats* st, GuidedReport* rep, size_t max_waves = ~size_t(0)) {
    Guider guider;
    guider.stree.init(scene.bounds(), it.spatial_filter, it.directional_filter);
    guider.directional_threshold = 0.01f;  // hard-coded (mod.rs:73)
    guider.spatial_threshold = it.spatial_threshold;
    size_t wave_index = 0, waves_done = 0;
    for (size_t wave_size : wave_sample_counts(spp)) {
        if (waves_done++ >= max_waves) break;
        film.clear();
        guider.new_iteration();
        guided_wave_passes(scene, cam, it, guider, film, wave_index, wave_size, n_threads, st);
        wave_index += wave_size;
    }
    if (rep) {
        rep->stree_nodes = guider.stree.nodes.size();
        for (auto& n : guider.stree.nodes)
            if (n.is_leaf()) {
                rep->stree_leaves++;
                rep->dtree_nodes_total += n.sampler.nodes.size() - n.sampler.free_slots.size();  // live nodes (freed slots stay in the Vec)
            }
    }
}

inline double film_scale(const IntegratorDesc& it, size_t spp) {  // integrator/mod.rs:47-55
    if (it.kind == IntegratorKind::Guided) return 1.0 / (double)wave_sample_counts(spp).back();
    return 1.0 / (double)spp;
}

}  // namespace orc
