// pt_kernels.h — wavefront stages for the `path` and `lightpath` integrators (gfx950, wave64).
// Every stage is a grid-stride kernel over a dense range whose length lives in device memory, so the host
// enqueues a whole batch of bounces without reading anything back.
#pragma once
#include "pt_device.h"
#include "pt_state.h"

namespace pt {

// kBlock (threads per workgroup) is defined in pt_types.h
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 2  // minimum waves per SIMD the shade kernels are compiled for (register budget 512 / N)
#endif
#ifndef PT_WIDE_WAVES
#define PT_WIDE_WAVES 6  // waves per SIMD the four-wide walkers are compiled for (a wide node is 32 registers of loads in flight: at 8 they spill)
#endif
#ifndef PT_SORT_SHADE
#define PT_SORT_SHADE 1  // shade kernels with more than one BSDF regroup a workgroup's lanes by material kind (block_sort_by_class)
#endif
#ifndef PT_PATH_DIFFUSE_WAVES
#define PT_PATH_DIFFUSE_WAVES 3  // k_shade_path without conductors: 3 waves per SIMD fit (171 registers otherwise, one too many)
#endif
#ifndef PT_PATH_CONDUCTOR_WAVES
#define PT_PATH_CONDUCTOR_WAVES 3  // k_shade_path with conductors: 213 registers at 2 waves; at 3 it spills 160 B per lane and is still faster
                                   // (veach-bidir shade 17.9 -> 16.4 ms per 64 spp, veach-mis 21.9 -> 21.8; k_shade_guided loses: 5.7 -> 6.4)
#endif

// One atomic per wave: lanes with `want` get consecutive indices from *counter.
PT_D uint32_t wave_alloc(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask == 0ull) return 0;
    const uint32_t lane = __lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
// Slot allocation for a whole workgroup with ONE global atomic per loop iteration: the (n_active, n_shadow) pair of
// a RoundCounters is bumped as a single 64-bit word, lanes are ranked with ballots and an LDS add per wave. Every
// thread of the block must call it (same trip count). Per-wave global atomics on one word saturate near 100 per us
// (MI355X_MICROARCH.md, "dequeue"), which is what a 4 M-path round would otherwise spend its time on.
struct BlockAlloc {
    uint32_t cnt[2];
    uint32_t base[2];
};
// Workgroup barrier for data exchanged through LDS only. __syncthreads() also drains the vector-memory counter, i.e.
// it makes every wave wait for the acknowledgement of the global stores it has just issued (a full memory round trip
// per loop iteration in the shade kernels); the slot exchange below needs LDS visibility and nothing else.
PT_D void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
PT_D void block_alloc2(BlockAlloc& sm, RoundCounters* rc, bool want_a, bool want_b, uint32_t& slot_a, uint32_t& slot_b) {
    if (threadIdx.x == 0) sm.cnt[0] = sm.cnt[1] = 0;
    lds_barrier();
    const uint32_t lane = __lane_id();
    const unsigned long long ma = __ballot(want_a), mb = __ballot(want_b);
    uint32_t wa = 0, wb = 0;
    if (lane == 0) {
        if (ma) wa = atomicAdd(&sm.cnt[0], (uint32_t)__popcll(ma));
        if (mb) wb = atomicAdd(&sm.cnt[1], (uint32_t)__popcll(mb));
    }
    wa = __shfl(wa, 0);
    wb = __shfl(wb, 0);
    lds_barrier();
    if (threadIdx.x == 0) {
        const unsigned long long add = (unsigned long long)sm.cnt[0] | ((unsigned long long)sm.cnt[1] << 32);
        unsigned long long old = 0;
        if (add) old = atomicAdd(reinterpret_cast<unsigned long long*>(&rc->n_active), add);
        sm.base[0] = (uint32_t)old;
        sm.base[1] = (uint32_t)(old >> 32);
    }
    lds_barrier();
    const unsigned long long below = (1ull << lane) - 1ull;
    slot_a = sm.base[0] + wa + (uint32_t)__popcll(ma & below);
    slot_b = sm.base[1] + wb + (uint32_t)__popcll(mb & below);
}
// block_alloc2 with the survivors in two classes: `want_a` lanes take slots from the bottom of the state set
// (rc->n_active), `want_t` lanes from the top (*top, a counter on a cache line of its own: the two atomics are in flight
// together and do not queue behind one another at the memory side); `want_b` lanes take shadow-queue slots as before.
struct BlockAlloc3 {
    uint32_t cnt[3];
    uint32_t base[3];
};
PT_D void block_alloc3(BlockAlloc3& sm, RoundCounters* rc, uint32_t* top, uint32_t capacity, bool want_a, bool want_t, bool want_b, uint32_t& slot_a, uint32_t& slot_b) {
    if (threadIdx.x == 0) sm.cnt[0] = sm.cnt[1] = sm.cnt[2] = 0;
    lds_barrier();
    const uint32_t lane = __lane_id();
    const unsigned long long ma = __ballot(want_a), mt = __ballot(want_t), mb = __ballot(want_b);
    uint32_t wa = 0, wt = 0, wb = 0;
    if (lane == 0) {
        if (ma) wa = atomicAdd(&sm.cnt[0], (uint32_t)__popcll(ma));
        if (mt) wt = atomicAdd(&sm.cnt[1], (uint32_t)__popcll(mt));
        if (mb) wb = atomicAdd(&sm.cnt[2], (uint32_t)__popcll(mb));
    }
    wa = __shfl(wa, 0);
    wt = __shfl(wt, 0);
    wb = __shfl(wb, 0);
    lds_barrier();
    if (threadIdx.x == 0) {
        const unsigned long long add = (unsigned long long)sm.cnt[0] | ((unsigned long long)sm.cnt[2] << 32);
        unsigned long long old = 0;
        uint32_t old_t = 0;
        if (add) old = atomicAdd(reinterpret_cast<unsigned long long*>(&rc->n_active), add);
        if (sm.cnt[1]) old_t = atomicAdd(top, sm.cnt[1]);
        sm.base[0] = (uint32_t)old;
        sm.base[2] = (uint32_t)(old >> 32);
        sm.base[1] = old_t;
    }
    lds_barrier();
    const unsigned long long below = (1ull << lane) - 1ull;
    slot_a = want_t ? capacity - 1u - (sm.base[1] + wt + (uint32_t)__popcll(mt & below)) : sm.base[0] + wa + (uint32_t)__popcll(ma & below);
    slot_b = sm.base[2] + wb + (uint32_t)__popcll(mb & below);
}
// Lanes of one workgroup trip regrouped by the code they are about to run. After a bounce the 256 entries a workgroup
// reads hold a mix of surfaces — Lambertian, rough conductor, dielectric, misses — and a wave runs every BSDF's code
// that one of its lanes needs (veach-mis, k_shade_path<7>: 0.34 of the VALU lanes active). The hit record carries the
// material's kind (pt_types.h: tag), so before anything else is loaded the workgroup sorts its 256 indices by class
// through LDS (a counting sort: ballots per class, one LDS word per (wave, class)) and lane t takes the t-th index of
// the sorted order: waves become (nearly) uniform in class. Results do not depend on it: an entry is shaded by another
// lane, nothing else. `cls` in [0, kSortClasses); returns the offset (0..255) of the entry this thread takes.
// Every thread of the workgroup calls it (two barriers).
constexpr uint32_t kSortClasses = 4;
struct BlockSort {
    uint32_t count[4][kSortClasses];  // [wave][class]
    uint32_t order[kBlock];
};
PT_D uint32_t block_sort_by_class(BlockSort& sm, uint32_t cls) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x / 64u;
    unsigned long long m[kSortClasses];
#pragma unroll
    for (uint32_t k = 0; k < kSortClasses; k++) m[k] = __ballot(cls == k);
    if (lane < kSortClasses) sm.count[wave][lane] = (uint32_t)__popcll(lane == 0 ? m[0] : (lane == 1 ? m[1] : (lane == 2 ? m[2] : m[3])));
    lds_barrier();
    uint32_t base = 0;
#pragma unroll
    for (uint32_t k = 0; k < kSortClasses; k++) {
#pragma unroll
        for (uint32_t w = 0; w < 4; w++) {
            const uint32_t c = sm.count[w][k];
            if (k < cls || (k == cls && w < wave)) base += c;
        }
    }
    const unsigned long long mine = cls == 0 ? m[0] : (cls == 1 ? m[1] : (cls == 2 ? m[2] : m[3]));
    sm.order[base + (uint32_t)__popcll(mine & ((1ull << lane) - 1ull))] = threadIdx.x;
    lds_barrier();
    return sm.order[threadIdx.x];
}
// Sum of a per-thread count over the block, added to *counter once (call after the loop, by every thread).
PT_D void block_count_flush(uint32_t* sm_word, uint32_t* counter, uint32_t mine) {
    if (threadIdx.x == 0) *sm_word = 0;
    __syncthreads();
    uint32_t v = mine;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if (__lane_id() == 0 && v) atomicAdd(sm_word, v);
    __syncthreads();
    if (threadIdx.x == 0 && *sm_word) atomicAdd(counter, *sm_word);
}

PT_D void wave_count(uint32_t* counter, bool want) {
    const unsigned long long mask = __ballot(want);
    if (mask != 0ull && (int)__lane_id() == __ffsll((long long)mask) - 1) atomicAdd(counter, (uint32_t)__popcll(mask));
}

// Traversal census (bench / roofline accounting only): per-lane node and triangle counts summed into
// census[base + 0] and census[base + 1]. Compiled out of the production instantiations.
PT_D void census_add(unsigned long long* census, int base, uint32_t nodes, uint32_t tris) {
    unsigned long long a = nodes, b = tris;
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        b += __shfl_down(b, off);
    }
    if (__lane_id() == 0) {
        atomicAdd(census + base, a);
        atomicAdd(census + base + 1, b);
    }
}

PT_D Spec load_spec(const float4* a, const float4* b, uint32_t i) {
    const float4 x = a[i], y = b[i];
    Spec s;
    s.v[0] = x.x; s.v[1] = x.y; s.v[2] = x.z; s.v[3] = x.w;
    s.v[4] = y.x; s.v[5] = y.y; s.v[6] = y.z; s.v[7] = y.w;
    return s;
}
PT_D Spec spec_from(float4 a, float4 b) {
    Spec s;
    s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w;
    s.v[4] = b.x; s.v[5] = b.y; s.v[6] = b.z; s.v[7] = b.w;
    return s;
}
PT_D void store_spec(float4* a, float4* b, uint32_t i, const Spec& s) {
    a[i] = make_float4(s.v[0], s.v[1], s.v[2], s.v[3]);
    b[i] = make_float4(s.v[4], s.v[5], s.v[6], s.v[7]);
}

// SamplePixel::add / SplatPixel::add (camera.rs:138-146): f64 RGB accumulation, hardware f64 atomics.
// Device film: kFilmStride doubles per pixel (RGB + one pad) so that a pixel never straddles a 64-byte atomic request;
// the C ABI hands out packed RGB (k_pack_film).
constexpr uint32_t kFilmStride = 4;
PT_D void film_add_rgb(double* film, uint32_t pixel, const double rgb[3]) {
    unsafeAtomicAdd(film + kFilmStride * (size_t)pixel + 0, rgb[0]);
    unsafeAtomicAdd(film + kFilmStride * (size_t)pixel + 1, rgb[1]);
    unsafeAtomicAdd(film + kFilmStride * (size_t)pixel + 2, rgb[2]);
}
// `n_copies` private films (light tracing, below) are summed in a fixed order.
__global__ __launch_bounds__(kBlock) void k_pack_film(const double* film, double* rgb_out, uint32_t n_pixels, uint32_t n_copies) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n_pixels) return;
    const size_t at = kFilmStride * (size_t)(i / 3u) + i % 3u;
    double v = film[at];
    for (uint32_t k = 1; k < n_copies; k++) v += film[at + (size_t)k * kFilmStride * n_pixels];
    rgb_out[i] = v;
}
// Light tracing connects every path's first vertex — a point ON the light — to the camera, so a few thousand pixels
// (the light's image) receive one splat per path each: same-address atomics serialise at the memory side and that
// contention, not the ray casting, set k_occlude_splat's time. kLightFilms private films, picked by workgroup, divide it.
constexpr uint32_t kLightFilms = 16;
// The Film add issued by the whole wave (every lane must get here; `want` marks the lanes with something to add).
// Float atomics run at the memory side at a fixed rate of 64-byte requests (MI355X_MICROARCH.md "Global float
// atomics"): one lane per pixel makes three instructions of 64 single-double requests each; dealing the three
// channels of a pixel to three neighbouring lanes makes each pixel's 24 bytes (and neighbouring pixels) one request.
// A non-finite sample is dropped and counted (kGuardNonFinite): the reference asserts on it in `path` (path_tracer.rs:46)
// and would poison the pixel for good in the other integrators. It happens: a bounce that lands on its own origin (an
// edge shared by two triangles, t ~ 0) makes AreaLight::pdf_li_solid_angle 0/0 in the MIS weight of the next emitter hit.
PT_D void film_add_rgb_wave(double* film, uint32_t pixel, const double rgb[3], bool want, uint32_t* guards) {
    if (want && !(isfinite(rgb[0]) && isfinite(rgb[1]) && isfinite(rgb[2]))) {
        want = false;
        atomicAdd(guards + kGuardNonFinite, 1u);
    }
    if (__ballot(want) == 0ull) return;
    const uint32_t lane = __lane_id();
#pragma unroll
    for (uint32_t pass = 0; pass < 3; pass++) {
        const uint32_t idx = pass * 64u + lane;
        const uint32_t src = idx / 3u, c = idx - 3u * src;
        const uint32_t px = (uint32_t)__shfl((int)pixel, (int)src);
        const int w = __shfl(want ? 1 : 0, (int)src);
        const double v0 = __shfl(rgb[0], (int)src), v1 = __shfl(rgb[1], (int)src), v2 = __shfl(rgb[2], (int)src);
        if (w) unsafeAtomicAdd(film + kFilmStride * (size_t)px + c, c == 0u ? v0 : (c == 1u ? v1 : v2));
    }
}
PT_D void film_add_wave(const SceneDev& sc, double* film, const float4* lambda_tab, uint32_t row, uint32_t pixel, const Spec& l, bool sec_term, bool want) {
    double rgb[3] = {0.0, 0.0, 0.0};
    if (want) {
        const Lambdas wl = load_lambdas(lambda_tab, row, true);
        spec_to_rgb(l, wl, sec_term, sc.cie_xyz, rgb);
    }
    film_add_rgb_wave(film, pixel, rgb, want, sc.guards);
}

struct RenderParams {
    uint32_t max_depth;
    int mode;
    uint32_t capacity;       // slots per state set
    uint32_t resx;           // image width (pixel index = x + y*resx)
    uint32_t n_pixels;
    const float4* lambda_tab;  // per wavelength-table entry: 8 lambdas + 8 pdfs (see ptgpu.hip: wavelength tables)
    SpecTab spec;            // the scene's spectra at every table entry's wavelengths
    double* film;            // resx*resy*kFilmStride
    uint32_t film_copies;    // 1, or kLightFilms private films back to back (lightpath)
};

__global__ __launch_bounds__(kBlock) void k_build_lambda_table(const float* u, float4* tab, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Lambdas w = sample_visible(u[i]);
    tab[4 * (size_t)i + 0] = make_float4(w.l[0], w.l[1], w.l[2], w.l[3]);
    tab[4 * (size_t)i + 1] = make_float4(w.l[4], w.l[5], w.l[6], w.l[7]);
    tab[4 * (size_t)i + 2] = make_float4(w.pdf[0], w.pdf[1], w.pdf[2], w.pdf[3]);
    tab[4 * (size_t)i + 3] = make_float4(w.pdf[4], w.pdf[5], w.pdf[6], w.pdf[7]);
}

// SpecTab builder: one thread per (spectrum, row).
__global__ __launch_bounds__(kBlock) void k_build_spectrum_table(SceneDev sc, const float4* lambda_tab, uint32_t n_rows, uint32_t n_materials, uint32_t n_spectra,
                                                              float4* out) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_spectra * n_rows) return;
    const uint32_t spectrum = g / n_rows, row = g % n_rows;
    const Lambdas wl = load_lambdas(lambda_tab, row, false);
    SpectrumDev sd;
    if (spectrum < 2 * n_materials) {
        const MaterialDev m = sc.materials[spectrum / 2];
        sd = (spectrum & 1u) ? m.s1 : m.s0;
    } else {
        sd = sc.emissions[spectrum - 2 * n_materials];
    }
    const Spec v = spectrum_eval(sd, wl, sc.d65);
    out[2 * (size_t)g] = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
    out[2 * (size_t)g + 1] = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
}

// ------------------------------------------------------------------ ray generation
// Sampler::init_pixel + wavelength_sample + Camera::sample_we (path_tracer.rs:40-43).
// Path g of the batch = pixel (g % n_pixels), sample slot (g / n_pixels); the sample index is
// first_sample + slot * sample_step (multi-GPU shards interleave sample indices).
__global__ __launch_bounds__(kBlock) void k_raygen_camera(SceneDev sc, CameraDev cam, RenderParams rp, PathCols st, RoundCounters* cnt,
                                                          uint32_t n_paths, uint32_t n_pixels, uint64_t first_sample, uint64_t sample_step) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t g = blockIdx.x * blockDim.x + threadIdx.x; g < n_paths; g += stride) {
        const uint32_t pixel = g % n_pixels, slot = g / n_pixels;
        const uint64_t si = first_sample + (uint64_t)slot * sample_step;
        const uint32_t ix = pixel % cam.resx, iy = pixel / cam.resx;
        Rng rng = rng_init_pixel(ix, iy, si);
        const float u0 = rng_f32(rng), u1 = rng_f32(rng);
        V3 o, d;
        camera_sample_we(cam, ix, iy, u0, u1, o, d);
        st.o_pdf[g] = make_float4(o.x, o.y, o.z, 0.0f);
        st.d_flags[g] = make_float4(d.x, d.y, d.z, u2f(FLAG_FRESH));  // beta = 1, L = 0, no previous vertex, path id = g: k_shade_path knows
        st.rng_px[g] = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), pixel, (uint32_t)si);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) cnt[0].n_active = n_paths;
}

// ------------------------------------------------------------------ extend (closest hit)
// Scene::intersect -> Bvh::intersect (scene.rs:184-186, bvh.rs:86-142) for slots [0, n_active).
// `top` (may be null) counts the entries queued from the top of the state set (slot_of_index).
template <bool CENSUS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? PT_WIDE_WAVES : 8) void k_extend(SceneDev sc, PathCols st, HitCols hit, const RoundCounters* cnt, unsigned long long* census, const uint32_t* top,
                                                      uint32_t capacity) {
    const uint32_t n_bottom = cnt->n_active;
    const uint32_t n = n_bottom + (top ? *top : 0u);
    if (blockIdx.x * blockDim.x >= n) return;  // late rounds: most workgroups have nothing to do, not even staging nodes
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t cn = 0, ct = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const uint32_t i = slot_of_index(j, n_bottom, capacity);
        float4 o = st.o_pdf[i], d = st.d_flags[i];
        asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(d.x), "+v"(d.y), "+v"(d.z));  // one load each (see bvh_intersect)
        const Hit h = bvh_intersect<CENSUS, false, WIDE>(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), f_inf(), &cn, &ct);
        hit.tb[i] = make_float4(h.th.t, h.th.b0, h.th.b1, h.th.b2);
        hit.prim[i] = make_uint2(h.prim, h.tag);
    }
    if (CENSUS) census_add(census, 0, cn, ct);
}

// The same for the four-wide walk over the queue as a stream (bvh_stream_wide): lanes take a new ray as they finish.
#ifndef PT_STREAM_WAVES
#define PT_STREAM_WAVES 6
#endif
template <bool CENSUS>
__global__ __launch_bounds__(kBlock, PT_STREAM_WAVES) void k_extend_stream(SceneDev sc, PathCols st, HitCols hit, const RoundCounters* cnt, unsigned long long* census, const uint32_t* top,
                                                                          uint32_t capacity) {
    const uint32_t n_bottom = cnt->n_active;
    const uint32_t n = n_bottom + (top ? *top : 0u);
    if (blockIdx.x * blockDim.x >= n) return;
    bvh_stage_nodes<true>(sc);
    uint32_t cn = 0, ct = 0;
    const auto fetch = [&](uint32_t j, V3& o, V3& d) {
        const uint32_t i = slot_of_index(j, n_bottom, capacity);
        float4 a = st.o_pdf[i], b = st.d_flags[i];
        asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z));
        o = mk3(a.x, a.y, a.z);
        d = mk3(b.x, b.y, b.z);
    };
    const auto emit = [&](bool done, uint32_t j, const Hit& h) {
        if (!done) return;
        const uint32_t i = slot_of_index(j, n_bottom, capacity);
        hit.tb[i] = make_float4(h.th.t, h.th.b0, h.th.b1, h.th.b2);
        hit.prim[i] = make_uint2(h.prim, h.tag);
    };
    if (sc.lds_wnodes) bvh_stream_wide<CENSUS, false, true>(sc, n, f_inf(), fetch, emit, &cn, &ct);
    else bvh_stream_wide<CENSUS, false, false>(sc, n, f_inf(), fetch, emit, &cn, &ct);
    if (CENSUS) census_add(census, 0, cn, ct);
}

// Parity hook: caller-supplied rays (ox,oy,oz,dx,dy,dz,tmax) -> ptgpu_hit-shaped records + traversal counts.
struct HitOut {
    uint32_t mesh_index, triangle_index, leaf_index, hit;
    float t, b0, b1, b2;
};
template <bool COUNT, bool WIDE>  // COUNT: the instrumented walk (nodes popped, triangles tested), as the census variants of the stages
__global__ __launch_bounds__(kBlock) void k_trace_batch(SceneDev sc, const float* rays, uint32_t n, HitOut* out, unsigned long long* counts) {
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned long long nn = 0, nt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* r = rays + 7 * (size_t)i;
        uint32_t a = 0, b = 0;
        const Hit h = bvh_intersect<COUNT, false, WIDE>(sc, mk3(r[0], r[1], r[2]), mk3(r[3], r[4], r[5]), r[6], &a, &b);
        nn += a;
        nt += b;
        HitOut o;
        if (h.prim != kNoHit) {
            o.mesh_index = f2u(sc.tri_v[3 * h.prim].w);
            o.triangle_index = f2u(sc.tri_v[3 * h.prim + 1].w);
            o.leaf_index = h.prim;
            o.hit = 1;
            o.t = h.th.t; o.b0 = h.th.b0; o.b1 = h.th.b1; o.b2 = h.th.b2;
        } else {
            o.mesh_index = o.triangle_index = o.leaf_index = o.hit = 0;
            o.t = o.b0 = o.b1 = o.b2 = 0.0f;
        }
        out[i] = o;
    }
    if (COUNT && counts) {
        atomicAdd(counts + 0, nn);
        atomicAdd(counts + 1, nt);
    }
}
template <bool WIDE>
__global__ __launch_bounds__(kBlock) void k_occluded_batch(SceneDev sc, const float* from, const float* to, uint32_t n, uint8_t* out) {
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* a = from + 3 * (size_t)i;
        const float* b = to + 3 * (size_t)i;
        out[i] = scene_occluded<false, WIDE>(sc, mk3(a[0], a[1], a[2]), mk3(b[0], b[1], b[2])) ? 1 : 0;
    }
}

// Small scenes (the Cornell box: 24 triangles, 2 of them lights, 5 materials) keep their triangles, their light table and
// their material heads in LDS for the shade kernels (SMALL): what this kernel is short of is the CU's vector-memory front
// end — every 16-byte-per-lane load holds the address pipeline a good 16 cycles whatever it hits — and LDS is a second,
// independent pipe. Same values from another place; a separate instantiation, so that no load becomes a flat one.
constexpr uint32_t kSmallTris = 64, kSmallLights = 16, kSmallMats = 16;
template <bool SMALL>
struct SmallSceneLds {
    float4 tri[3 * kSmallTris];
    float4 light[3 * kSmallLights];
    MatHead mat[kSmallMats];
};
template <>
struct SmallSceneLds<false> {};
PT_HD bool scene_is_small(uint32_t n_tris, uint32_t n_emissive_tris, uint32_t n_materials) { return n_tris <= kSmallTris && n_emissive_tris <= kSmallLights && n_materials <= kSmallMats; }

template <class Lds>
PT_D void stage_small_scene(const SceneDev& sc, Lds& sm) {  // every thread of the workgroup
    for (uint32_t k = threadIdx.x; k < 3u * sc.n_tris; k += kBlock) sm.tri[k] = sc.tri_v[k];
    for (uint32_t k = threadIdx.x; k < 3u * sc.n_emissive_tris; k += kBlock) sm.light[k] = sc.light_v[k];
    for (uint32_t k = threadIdx.x; k < sc.n_materials; k += kBlock) sm.mat[k] = MatHead{sc.materials[k].kind, sc.materials[k].roughness};
    __syncthreads();
}

// ------------------------------------------------------------------ shade: `path` integrator
// One iteration of PathIntegrator::li's loop body after the intersection (path_tracer.rs:69-180).
template <int KINDS, bool SMALL>
__global__ __launch_bounds__(kBlock, KINDS != 7 ? PT_PATH_DIFFUSE_WAVES : PT_PATH_CONDUCTOR_WAVES) void k_shade_path(SceneDev sc, RenderParams rp, PathCols in, HitCols hit, PathCols out, ShadowCols sh,
                                                       const RoundCounters* cur, RoundCounters* nxt) {
    __shared__ BlockAlloc sm_alloc;
    __shared__ uint32_t sm_count;
    __shared__ BlockSort sm_sort;
    constexpr bool kSort = PT_SORT_SHADE && KINDS == 7;  // measured: a gain where conductors mix with other surfaces (veach-mis shade -9 %, veach-bidir -7 %), none or a loss on the Cornell box
    __shared__ SmallSceneLds<SMALL> sm_scene;
    if constexpr (SMALL) stage_small_scene(sc, sm_scene);
    uint32_t n_shaded = 0;
    const uint32_t n = cur->n_active;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n + stride - 1) / stride;  // same trip count for every thread: allocation is block-uniform
    for (uint32_t it = 0; it < n_iter; it++) {
        uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        if (kSort) {
            // class of entry i: 0 Lambertian, 1 dielectric, 2 conductor, 3 nothing to shade (miss, pass-through, beyond the queue)
            uint32_t cls = 3;
            if (i < n) {
                const uint2 pt = hit.prim[i];
                if (pt.x != kNoHit && tag_material(pt.y) >= 0) cls = tag_kind(pt.y);
            }
            i = it * stride + blockIdx.x * blockDim.x + block_sort_by_class(sm_sort, cls);
        }
        const bool valid = i < n;
        bool alive = false, want_shadow = false, do_film = false, shaded = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), prev_p = mk3(0, 0, 0);
        float prev_pdf = 0.0f;
        uint32_t flags = 0, pid = 0;
        Spec beta = spec_splat(0.0f), L = spec_splat(0.0f), contrib = spec_splat(0.0f);
        uint4 rp4 = make_uint4(0, 0, 0, 0);
        V3 sh_from = mk3(0, 0, 0), sh_to = mk3(0, 0, 0);
        if (valid) {
            const float4 o4 = in.o_pdf[i], d4 = in.d_flags[i];
            o = mk3(o4.x, o4.y, o4.z);
            d = mk3(d4.x, d4.y, d4.z);
            prev_pdf = o4.w;
            flags = f2u(d4.w);
            const bool fresh = (flags & FLAG_FRESH) != 0u;  // straight out of k_raygen_camera: 80 bytes of state that need not be read
            flags &= ~FLAG_FRESH;
            rp4 = in.rng_px[i];
            if (!fresh) L = load_spec(in.L0, in.L1, i);
            const uint2 prim_tag = hit.prim[i];
            const uint32_t prim = prim_tag.x;
            if (prim == kNoHit) {
                do_film = true;  // eval_envmap == ZERO without an envmap (scene.rs:301-306)
            } else {
                const float4 tb = hit.tb[i];
                if (fresh) {
                    pid = i;
                    beta = spec_splat(1.0f);
                } else {
                    const float4 pp = in.prev_p[i];
                    prev_p = mk3(pp.x, pp.y, pp.z);
                    pid = f2u(pp.w);
                    beta = load_spec(in.beta0, in.beta1, i);
                }
                Surface sf;
                if constexpr (SMALL) sf = make_surface_from(sc, sm_scene.tri, prim, tb.y, tb.z, tb.w);
                else sf = make_surface(sc, prim, tb.y, tb.z, tb.w);
                const int material = tag_material(prim_tag.y);
                uint32_t depth = flags & FLAG_DEPTH_MASK;
                if (material < 0) {
                    // no BSDF: continue straight on, do not count a bounce (path_tracer.rs:76-80)
                    o = offset_ray_origin(sf.blo, sf.bhi, sf.ng, d);
                    flags |= FLAG_SPECULAR_BOUNCE;
                    alive = passthrough_step(flags);
                    if (!alive) atomicAdd(sc.guards + kGuardPassThrough, 1u);
                    do_film = !alive;
                } else {
                    shaded = true;
                    flags &= ~FLAG_PASS_MASK;
                    const V3 wo = -d;
                    const V3 p = surf_p(sf);
                    Bsdf bsdf;
                    if constexpr (SMALL) bsdf = make_bsdf_head<KINDS>(rp.spec, sm_scene.mat[material], material, rp4.w);
                    else bsdf = make_bsdf<KINDS>(sc, rp.spec, material, rp4.w);
                    const bool spec_bounce = (flags & FLAG_SPECULAR_BOUNCE) != 0;
                    // emitted radiance of the surface that was hit (path_tracer.rs:82-108)
                    if (sf.tri.mesh < sc.n_emissive_meshes) {
                        const bool direct = (depth == 0) || spec_bounce;
                        if (rp.mode == MODE_PT || direct) {
                            L = L + light_emitted(rp.spec, sf.tri, sf.ng, p, o, rp4.w) * beta;
                        } else if (rp.mode == MODE_MIS) {
                            const float pb = prev_pdf;
                            const float pl = light_pdf_li(sf.tri, sf.ng, prev_p, p);
                            const float w = (pb * pb) / (pb * pb + pl * pl);
                            L = L + w * light_emitted(rp.spec, sf.tri, sf.ng, p, o, rp4.w) * beta;
                        }
                    }
                    Rng rng{(uint64_t)rp4.x | ((uint64_t)rp4.y << 32), rng_inc_for_pixel(rp4.z % rp.resx, rp4.z / rp.resx)};
                    // direct lighting (path_tracer.rs:111-158); the three floats are drawn only for non-specular BSDFs
                    if (rp.mode != MODE_PT && !bsdf_is_specular(bsdf)) {
                        const float u1d = rng_f32(rng), ua = rng_f32(rng), ub = rng_f32(rng);
                        Tri lt;
                        float pmf;
                        LiSample ls;
                        bool have_light;
                        if constexpr (SMALL) have_light = choose_light_from(sc, sm_scene.light, u1d, lt, pmf);
                        else have_light = choose_light(sc, u1d, lt, pmf);
                        if (have_light && light_sample_li(lt, p, ua, ub, ls)) {
                            Spec f = bsdf_eval(bsdf, sf.frame, sf.ng, wo, ls.wl, true) / (ls.pdf * pmf);
                            if (!spec_is_black(f)) {
                                const V3 lp = (ls.blo + ls.bhi) / 2.0f;
                                const Spec le = spec_lookup(rp.spec, rp.spec.emission_base + lt.mesh, rp4.w);
                                if (rp.mode == MODE_NEE) contrib = f * beta * le;
                                else {
                                    const float pl = ls.pdf;
                                    const float pb = bsdf_pdf(bsdf, sf.frame, wo, ls.wl);
                                    const float w = (pl * pl) / (pl * pl + pb * pb);
                                    contrib = w * f * beta * le;
                                }
                                // Scene::occluded_interaction (scene.rs:219-223)
                                const V3 w = normalize(lp - p);
                                sh_from = offset_ray_origin(sf.blo, sf.bhi, sf.ng, w);
                                sh_to = lp;
                                want_shadow = true;
                            }
                        }
                    }
                    const float u1d = rng_f32(rng), ua = rng_f32(rng), ub = rng_f32(rng);
                    BsdfSample bs;
                    // At the last bounce the reference still samples the BSDF, but nothing of the sample outlives the loop
                    // (path_tracer.rs:160-181: beta, the ray and the pdf feed an iteration that `depth < max_depth` denies)
                    // except terminate_secondary, which only a dielectric can raise: other surfaces skip the dead sample.
                    const bool dead_sample = depth + 1 >= rp.max_depth && bsdf.kind != MAT_DIELECTRIC;
                    if (!dead_sample && bsdf_sample(bsdf, sf.frame, sf.ng, wo, u1d, ua, ub, true, bs)) {
                        const float inv_pdf = bs.pdf;
#pragma unroll
                        for (int k = 0; k < NSW; k++) beta.v[k] *= bs.f.v[k] / inv_pdf;
                        if (!spec_is_black(beta)) {
                            flags = bs.specular ? (flags | FLAG_SPECULAR_BOUNCE) : (flags & ~FLAG_SPECULAR_BOUNCE);
                            if (bs.terminate_secondary) flags |= FLAG_SECONDARY_TERMINATED;
                            o = offset_ray_origin(sf.blo, sf.bhi, sf.ng, bs.wi);
                            d = bs.wi;
                            prev_p = p;
                            prev_pdf = bs.pdf;
                            depth += 1;
                            flags = (flags & ~FLAG_DEPTH_MASK) | (depth & FLAG_DEPTH_MASK);
                            alive = depth < rp.max_depth;
                        }
                    }
                    rp4.x = (uint32_t)rng.state;
                    rp4.y = (uint32_t)(rng.state >> 32);
                    if (!alive && !want_shadow) do_film = true;
                }
            }
        }
        // ---- wave-uniform tail: slot allocation and stores
        const bool want_terminal = valid && !alive && want_shadow;
        uint32_t a_slot, s_slot;
        block_alloc2(sm_alloc, nxt, alive, want_shadow, a_slot, s_slot);
        const uint32_t t_slot = wave_alloc(&nxt->n_terminal, want_terminal);  // rare: path ended with a shadow ray pending
        n_shaded += shaded ? 1u : 0u;
        if (alive) {
            out.o_pdf[a_slot] = make_float4(o.x, o.y, o.z, prev_pdf);
            out.d_flags[a_slot] = make_float4(d.x, d.y, d.z, u2f(flags));
            store_spec(out.beta0, out.beta1, a_slot, beta);
            store_spec(out.L0, out.L1, a_slot, L);
            out.rng_px[a_slot] = rp4;
            out.prev_p[a_slot] = make_float4(prev_p.x, prev_p.y, prev_p.z, u2f(pid));
        }
        uint32_t target = a_slot;
        if (want_terminal) {
            const uint32_t slot = rp.capacity - 1u - t_slot;
            store_spec(out.L0, out.L1, slot, L);
            out.rng_px[slot] = rp4;
            out.d_flags[slot] = make_float4(0.0f, 0.0f, 0.0f, u2f(flags));
            target = slot | kTerminalBit;
        }
        if (want_shadow) {
            sh.q0[s_slot] = make_float4(sh_from.x, sh_from.y, sh_from.z, u2f(target));
            sh.q1[s_slot] = make_float4(sh_to.x, sh_to.y, sh_to.z, 0.0f);
            store_spec(sh.q2, sh.q3, s_slot, contrib);
        }
        film_add_wave(sc, rp.film, rp.lambda_tab, rp4.w, rp4.z, L, (flags & FLAG_SECONDARY_TERMINATED) != 0, do_film && !spec_is_black(L));
    }
    block_count_flush(&sm_count, &nxt->n_segments, n_shaded);
}

// ------------------------------------------------------------------ occlude: `path` integrator
// Scene::occluded for the NEE shadow rays of this round; adds the MIS-weighted contribution to the path's
// radiance and finishes paths that ended in shade with this ray still pending.
template <bool CENSUS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? PT_WIDE_WAVES : 8) void k_occlude_path(SceneDev sc, RenderParams rp, PathCols st, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;  // late rounds: most workgroups have nothing to do, not even staging nodes
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t cn = 0, ct = 0;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {  // block-uniform trip count
        const uint32_t j = base + threadIdx.x;
        bool to_film = false;
        uint32_t pixel = 0;
        double rgb[3] = {0.0, 0.0, 0.0};
        if (j < n) {
            const float4 a = sh.q0[j], b = sh.q1[j];
            const bool occ = scene_occluded<CENSUS, WIDE>(sc, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), &cn, &ct);
            const uint32_t target = f2u(a.w);
            const uint32_t slot = target & ~kTerminalBit;
            if (target & kTerminalBit) {
                Spec L = load_spec(st.L0, st.L1, slot);
                if (!occ) L = L + load_spec(sh.q2, sh.q3, j);
                if (!spec_is_black(L)) {
                    const uint4 rp4 = st.rng_px[slot];
                    const uint32_t flags = f2u(st.d_flags[slot].w);
                    const Lambdas wl = load_lambdas(rp.lambda_tab, rp4.w, true);
                    spec_to_rgb(L, wl, (flags & FLAG_SECONDARY_TERMINATED) != 0, sc.cie_xyz, rgb);
                    pixel = rp4.z;
                    to_film = true;
                }
            } else if (!occ) {
                const Spec L = load_spec(st.L0, st.L1, slot) + load_spec(sh.q2, sh.q3, j);
                store_spec(st.L0, st.L1, slot, L);
            }
        }
        film_add_rgb_wave(rp.film, pixel, rgb, to_film, sc.guards);
    }
    if (CENSUS) census_add(census, 2, cn, ct);
}

// The same over the shadow queue as a stream (bvh_stream_wide), for the four-wide walk of large scenes. Film adds of
// paths that end here go out as their rays finish, wave-wide as above (another order of the same atomic adds).
template <bool CENSUS>
__global__ __launch_bounds__(kBlock, PT_STREAM_WAVES) void k_occlude_path_stream(SceneDev sc, RenderParams rp, PathCols st, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;
    bvh_stage_nodes<true>(sc);
    uint32_t cn = 0, ct = 0;
    const auto fetch = [&](uint32_t j, V3& o, V3& d) {
        const float4 a = sh.q0[j], b = sh.q1[j];
        o = mk3(a.x, a.y, a.z);
        d = mk3(b.x, b.y, b.z) - o;  // Scene::occluded: unnormalised segment, t_max 0.9999
    };
    const auto emit = [&](bool done, uint32_t j, const Hit& h) {
        bool to_film = false;
        uint32_t pixel = 0;
        double rgb[3] = {0.0, 0.0, 0.0};
        if (done) {
            const bool occ = h.prim != kNoHit;
            const uint32_t target = f2u(sh.q0[j].w);
            const uint32_t slot = target & ~kTerminalBit;
            if (target & kTerminalBit) {
                Spec L = load_spec(st.L0, st.L1, slot);
                if (!occ) L = L + load_spec(sh.q2, sh.q3, j);
                if (!spec_is_black(L)) {
                    const uint4 rp4 = st.rng_px[slot];
                    const uint32_t flags = f2u(st.d_flags[slot].w);
                    const Lambdas wl = load_lambdas(rp.lambda_tab, rp4.w, true);
                    spec_to_rgb(L, wl, (flags & FLAG_SECONDARY_TERMINATED) != 0, sc.cie_xyz, rgb);
                    pixel = rp4.z;
                    to_film = true;
                }
            } else if (!occ) {
                const Spec L = load_spec(st.L0, st.L1, slot) + load_spec(sh.q2, sh.q3, j);
                store_spec(st.L0, st.L1, slot, L);
            }
        }
        film_add_rgb_wave(rp.film, pixel, rgb, to_film, sc.guards);
    };
    if (sc.lds_wnodes) bvh_stream_wide<CENSUS, true, true>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    else bvh_stream_wide<CENSUS, true, false>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    if (CENSUS) census_add(census, 2, cn, ct);
}

// ------------------------------------------------------------------ lightpath integrator
// LightPathIntegrator::render per-sample head (light_path_tracer.rs:27-56): emit from a light, connect the
// emission point to the pinhole, start the random walk.
__global__ __launch_bounds__(kBlock) void k_raygen_light(SceneDev sc, CameraDev cam, RenderParams rp, PathCols st, ShadowCols sh, RoundCounters* cnt,
                                                         uint32_t n_paths, uint32_t n_pixels, uint64_t first_sample, uint64_t sample_step) {
    __shared__ BlockAlloc sm_alloc;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n_paths + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; it++) {
        const uint32_t g = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool alive = false, want_splat = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), sh_from = mk3(0, 0, 0);
        Spec beta = spec_splat(0.0f);
        uint4 rp4 = make_uint4(0, 0, 0, 0);
        double rgb[3] = {0.0, 0.0, 0.0};
        uint32_t splat_px = 0;
        if (g < n_paths) {
            const uint32_t pixel = g % n_pixels, slot = g / n_pixels;
            const uint64_t si = first_sample + (uint64_t)slot * sample_step;
            const uint32_t ix = pixel % cam.resx, iy = pixel / cam.resx;
            Rng rng = rng_init_pixel(ix, iy, si);
            const Lambdas wl = load_lambdas(rp.lambda_tab, (uint32_t)si, true);
            const float u1d = rng_f32(rng), u00 = rng_f32(rng), u01 = rng_f32(rng), u10 = rng_f32(rng), u11 = rng_f32(rng);
            Tri lt;
            float pmf;
            LeSample le;
            if (choose_light(sc, u1d, lt, pmf) && light_sample_le(lt, u00, u01, u10, u11, le)) {
                const Spec L = spec_lookup(rp.spec, rp.spec.emission_base + lt.mesh, (uint32_t)si);
                const V3 lp = (le.blo + le.bhi) / 2.0f;
                CameraWi cw;
                if (camera_sample_wi(cam, lp, cw)) {
                    const Spec f = spec_splat(cw.importance) * L * fabsf(dot(le.n, cw.wc)) / (cw.pdf * le.pdf_pos * pmf);
                    if (!spec_is_black(f)) {
                        spec_to_rgb(f, wl, false, sc.cie_xyz, rgb);
                        splat_px = splat_pixel_index(cam, cw.rx, cw.ry);
                        const V3 w = normalize(cam.p - lp);  // Scene::occluded_le_sample (scene.rs:224-228)
                        sh_from = offset_ray_origin(le.blo, le.bhi, le.n, w);
                        want_splat = true;
                    }
                }
                beta = L * fabsf(dot(le.n, le.dir)) / (le.pdf_pos * le.pdf_dir * pmf);
                o = offset_ray_origin(le.blo, le.bhi, le.n, le.dir);
                d = le.dir;
                alive = rp.max_depth > 0;
            }
            rp4 = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), pixel, (uint32_t)si);
        }
        uint32_t a_slot, s_slot;
        block_alloc2(sm_alloc, &cnt[0], alive, want_splat, a_slot, s_slot);
        if (alive) {
            st.o_pdf[a_slot] = make_float4(o.x, o.y, o.z, 0.0f);
            st.d_flags[a_slot] = make_float4(d.x, d.y, d.z, u2f(0u));
            store_spec(st.beta0, st.beta1, a_slot, beta);
            st.rng_px[a_slot] = rp4;
        }
        if (want_splat) {
            sh.q0[s_slot] = make_float4(sh_from.x, sh_from.y, sh_from.z, u2f(splat_px));
            sh.q1[s_slot] = make_float4(cam.p.x, cam.p.y, cam.p.z, 0.0f);
            double* r = reinterpret_cast<double*>(sh.q2);
            r[3 * (size_t)s_slot + 0] = rgb[0];
            r[3 * (size_t)s_slot + 1] = rgb[1];
            r[3 * (size_t)s_slot + 2] = rgb[2];
        }
    }
}

// LightPathIntegrator::random_walk loop body (light_path_tracer.rs:74-118). The depth counter counts loop
// iterations (a pass-through surface burns one, unlike `path`).
template <int KINDS, bool SMALL>
__global__ __launch_bounds__(kBlock, PT_SHADE_WAVES) void k_shade_light(SceneDev sc, CameraDev cam, RenderParams rp, PathCols in, HitCols hit, PathCols out, ShadowCols sh,
                                                        const RoundCounters* cur, RoundCounters* nxt) {
    __shared__ BlockAlloc sm_alloc;
    __shared__ uint32_t sm_count;
    __shared__ SmallSceneLds<SMALL> sm_scene;
    if constexpr (SMALL) stage_small_scene(sc, sm_scene);
    uint32_t n_shaded = 0;
    const uint32_t n = cur->n_active;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; it++) {
        const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool alive = false, want_splat = false, shaded = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1), sh_from = mk3(0, 0, 0);
        uint32_t flags = 0, splat_px = 0;
        Spec beta = spec_splat(0.0f);
        uint4 rp4 = make_uint4(0, 0, 0, 0);
        double rgb[3] = {0.0, 0.0, 0.0};
        if (i < n && hit.prim[i].x != kNoHit) {
            const float4 o4 = in.o_pdf[i], d4 = in.d_flags[i];
            o = mk3(o4.x, o4.y, o4.z);
            d = mk3(d4.x, d4.y, d4.z);
            flags = f2u(d4.w);
            rp4 = in.rng_px[i];
            beta = load_spec(in.beta0, in.beta1, i);
            const float4 tb = hit.tb[i];
            const Lambdas wl = load_lambdas(rp.lambda_tab, rp4.w, true);
            const uint2 prim_tag = hit.prim[i];
            Surface sf;
            if constexpr (SMALL) sf = make_surface_from(sc, sm_scene.tri, prim_tag.x, tb.y, tb.z, tb.w);
            else sf = make_surface(sc, prim_tag.x, tb.y, tb.z, tb.w);
            const int material = tag_material(prim_tag.y);
            uint32_t depth = (flags & FLAG_DEPTH_MASK) + 1;  // iterations consumed including this one
            if (material < 0) {
                o = offset_ray_origin(sf.blo, sf.bhi, sf.ng, d);
                alive = true;
            } else {
                shaded = true;
                const V3 wo = -d;
                const V3 p = surf_p(sf);
                Bsdf bsdf;
                if constexpr (SMALL) bsdf = make_bsdf_head<KINDS>(rp.spec, sm_scene.mat[material], material, rp4.w);
                else bsdf = make_bsdf<KINDS>(sc, rp.spec, material, rp4.w);
                CameraWi cw;
                if (!bsdf_is_specular(bsdf) && camera_sample_wi(cam, p, cw)) {
                    Spec f = bsdf_eval(bsdf, sf.frame, sf.ng, wo, cw.wc, false) * spec_splat(cw.importance) * beta / cw.pdf;
                    if (!spec_is_black(f)) {
                        spec_to_rgb(f, wl, (flags & FLAG_SECONDARY_TERMINATED) != 0, sc.cie_xyz, rgb);
                        splat_px = splat_pixel_index(cam, cw.rx, cw.ry);
                        const V3 w = normalize(cam.p - p);
                        sh_from = offset_ray_origin(sf.blo, sf.bhi, sf.ng, w);
                        want_splat = true;
                    }
                }
                Rng rng{(uint64_t)rp4.x | ((uint64_t)rp4.y << 32), rng_inc_for_pixel(rp4.z % rp.resx, rp4.z / rp.resx)};
                const float u1d = rng_f32(rng), ua = rng_f32(rng), ub = rng_f32(rng);
                BsdfSample bs;
                // (the sample of the walk's last iteration feeds nothing, light_path_tracer.rs:100-117: skipped)
                if (depth < rp.max_depth && bsdf_sample(bsdf, sf.frame, sf.ng, wo, u1d, ua, ub, false, bs)) {
                    const float pdf = bs.pdf;
#pragma unroll
                    for (int k = 0; k < NSW; k++) beta.v[k] *= bs.f.v[k] / pdf;
                    if (!spec_is_black(beta)) {
                        if (bs.terminate_secondary) flags |= FLAG_SECONDARY_TERMINATED;
                        o = offset_ray_origin(sf.blo, sf.bhi, sf.ng, bs.wi);
                        d = bs.wi;
                        alive = true;
                    }
                }
                rp4.x = (uint32_t)rng.state;
                rp4.y = (uint32_t)(rng.state >> 32);
            }
            flags = (flags & ~FLAG_DEPTH_MASK) | (depth & FLAG_DEPTH_MASK);
            alive = alive && depth < rp.max_depth;
        }
        uint32_t a_slot, s_slot;
        block_alloc2(sm_alloc, nxt, alive, want_splat, a_slot, s_slot);
        n_shaded += shaded ? 1u : 0u;
        if (alive) {
            out.o_pdf[a_slot] = make_float4(o.x, o.y, o.z, 0.0f);
            out.d_flags[a_slot] = make_float4(d.x, d.y, d.z, u2f(flags));
            store_spec(out.beta0, out.beta1, a_slot, beta);
            out.rng_px[a_slot] = rp4;
        }
        if (want_splat) {
            sh.q0[s_slot] = make_float4(sh_from.x, sh_from.y, sh_from.z, u2f(splat_px));
            sh.q1[s_slot] = make_float4(cam.p.x, cam.p.y, cam.p.z, 0.0f);
            double* r = reinterpret_cast<double*>(sh.q2);
            r[3 * (size_t)s_slot + 0] = rgb[0];
            r[3 * (size_t)s_slot + 1] = rgb[1];
            r[3 * (size_t)s_slot + 2] = rgb[2];
        }
    }    block_count_flush(&sm_count, &nxt->n_segments, n_shaded);
}

// Camera-connection visibility + Film::add_splat (camera.rs:161-166) with f64 atomics.
template <bool CENSUS, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_occlude_splat(SceneDev sc, RenderParams rp, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;  // late rounds: most workgroups have nothing to do, not even staging nodes
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    const double* r = reinterpret_cast<const double*>(sh.q2);
    uint32_t cn = 0, ct = 0;
    for (uint32_t base = blockIdx.x * blockDim.x; base < n; base += stride) {  // block-uniform trip count
        const uint32_t j = base + threadIdx.x;
        bool visible = false;
        uint32_t pixel = 0;
        double rgb[3] = {0.0, 0.0, 0.0};
        if (j < n) {
            const float4 a = sh.q0[j], b = sh.q1[j];
            if (!scene_occluded<CENSUS, WIDE>(sc, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), &cn, &ct)) {
                rgb[0] = r[3 * (size_t)j];
                rgb[1] = r[3 * (size_t)j + 1];
                rgb[2] = r[3 * (size_t)j + 2];
                pixel = f2u(a.w);
                visible = true;
            }
        }
        film_add_rgb_wave(rp.film + (size_t)(blockIdx.x % rp.film_copies) * kFilmStride * rp.n_pixels, pixel, rgb, visible, sc.guards);
    }
    if (CENSUS) census_add(census, 2, cn, ct);
}
// The same over the connection queue as a stream (bvh_stream_wide), for the four-wide walk of large scenes.
template <bool CENSUS>
__global__ __launch_bounds__(kBlock, PT_STREAM_WAVES) void k_occlude_splat_stream(SceneDev sc, RenderParams rp, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;
    bvh_stage_nodes<true>(sc);
    const double* r = reinterpret_cast<const double*>(sh.q2);
    double* film = rp.film + (size_t)(blockIdx.x % rp.film_copies) * kFilmStride * rp.n_pixels;
    uint32_t cn = 0, ct = 0;
    const auto fetch = [&](uint32_t j, V3& o, V3& d) {
        const float4 a = sh.q0[j], b = sh.q1[j];
        o = mk3(a.x, a.y, a.z);
        d = mk3(b.x, b.y, b.z) - o;  // Scene::occluded: unnormalised segment, t_max 0.9999
    };
    const auto emit = [&](bool done, uint32_t j, const Hit& h) {
        const bool visible = done && h.prim == kNoHit;
        uint32_t pixel = 0;
        double rgb[3] = {0.0, 0.0, 0.0};
        if (visible) {
            rgb[0] = r[3 * (size_t)j];
            rgb[1] = r[3 * (size_t)j + 1];
            rgb[2] = r[3 * (size_t)j + 2];
            pixel = f2u(sh.q0[j].w);
        }
        film_add_rgb_wave(film, pixel, rgb, visible, sc.guards);
    };
    if (sc.lds_wnodes) bvh_stream_wide<CENSUS, true, true>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    else bvh_stream_wide<CENSUS, true, false>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    if (CENSUS) census_add(census, 2, cn, ct);
}

}  // namespace pt
