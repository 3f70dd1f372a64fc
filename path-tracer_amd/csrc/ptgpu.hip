// ptgpu.hip — C ABI of libptgpu.so (include/ptgpu.h): context, scene flattening, wavefront render loops.
#include "../../include/ptgpu.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "bvh_build.h"
#include "pt_bvh_build.h"
#include "cie_data.inc"
#include "pt_guided.h"
#include "pt_kernels.h"
#include "pt_pinned.h"

using namespace pt;

// Scratch of ptgpu_trace_batch_noalloc (include/ptgpu.h): the device-side staging of the rays and of the hit records that
// ptgpu_trace_batch allocates per call. In host code this name is only a stand-in for the symbol: the device address
// comes from hipGetSymbolAddress (ptgpu_init, once per context with its device current) and lives in ptgpu_ctx::scratch.
// Every GPU the library runs on holds its own copy (15 MiB) until the process ends.
constexpr uint32_t kNoAllocMaxRays = 1u << 18;
constexpr size_t kNoAllocRayBytes = (size_t)kNoAllocMaxRays * 7 * sizeof(float);
constexpr size_t kNoAllocHitBytes = (size_t)kNoAllocMaxRays * sizeof(ptgpu_hit);
static __device__ __align__(256) unsigned char scratch[kNoAllocRayBytes + kNoAllocHitBytes];

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e__ = (expr);                                                                          \
        if (e__ != hipSuccess) return fail(PTGPU_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

const float kD65[CIE_N] = {PT_CIE_D65_VALUES};
const double kXyz[CIE_N][3] = {PT_CIE_XYZ_VALUES};

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0, cap = 0;
    hipError_t alloc(size_t count) {
        release();
        n = count;
        cap = count;
        if (count == 0) return hipSuccess;
        const hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) {  // nothing is held: a later reserve() of less must not take the failed size for capacity
            p = nullptr;
            n = cap = 0;
        }
        return e;
    }
    hipError_t upload(const std::vector<T>& v, hipStream_t s) {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        e = hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        return hipStreamSynchronize(s);
    }
    hipError_t reserve(size_t count) {  // grow-only alloc: contents are not preserved
        if (count <= cap) {
            n = count;
            return hipSuccess;
        }
        return alloc(count);
    }
    // Re-used between guided waves: grows by half when too small, copies without waiting (the caller keeps `src`
    // alive and synchronises the stream before touching it again).
    hipError_t upload_async(const T* src, size_t count, hipStream_t s) {
        if (count > cap) {
            release();
            const size_t want = count + count / 2 + 64;
            hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
            if (e != hipSuccess) return e;
            cap = want;
        }
        n = count;
        if (count == 0) return hipSuccess;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        cap = 0;
    }
    ~DevBuf() { release(); }
};

struct StateSet {
    DevBuf<float4> cols;  // 8 columns * capacity
    PathCols view(uint32_t cap) {
        PathCols c;
        float4* b = cols.p;
        c.o_pdf = b;
        c.d_flags = b + (size_t)cap;
        c.beta0 = b + 2 * (size_t)cap;
        c.beta1 = b + 3 * (size_t)cap;
        c.L0 = b + 4 * (size_t)cap;
        c.L1 = b + 5 * (size_t)cap;
        c.rng_px = reinterpret_cast<uint4*>(b + 6 * (size_t)cap);
        c.prev_p = b + 7 * (size_t)cap;
        return c;
    }
};

}  // namespace

struct ptgpu_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;  // guided: the record pass of batch b runs here while `stream` walks batch b + 1
    // guided: a large batch's records go through the record pass in slices, pipelined over three streams: k_guide_scan of a slice
    // on stream2, its locate / count scan / scatter on stream3, its splat on stream4
    hipStream_t stream3 = nullptr, stream4 = nullptr;
    static constexpr int kMaxRecordSlices = 16;
    hipEvent_t ev_scanned[kMaxRecordSlices] = {}, ev_grouped[kMaxRecordSlices] = {}, ev_splatted = nullptr;
    int n_cus = 256;
    // scene
    bool have_scene = false, have_camera = false;
    SceneDev scene{};
    CameraDev camera{};
    DevBuf<float4> d_nodes, d_tri_v, d_tri_n;
    DevBuf<MeshInfo> d_mesh_info;
    DevBuf<float4> d_light_v;
    DevBuf<MaterialDev> d_materials;
    DevBuf<SpectrumDev> d_emissions;
    DevBuf<uint32_t> d_emissive;
    DevBuf<float> d_d65;
    DevBuf<double> d_xyz;
    std::vector<uint32_t> h_nodes_words;  // 8 words per node (ptgpu_scene_bvh)
    std::vector<uint32_t> h_leaf_tris;    // mesh, tri per leaf triangle
    // render workspace
    uint32_t batch_paths = 0;  // 0 = default
    uint32_t capacity = 0;
    uint64_t fallback_cap = 0, fallback_unit = 0;  // path / lightpath: the batch size that fit where the device had less memory left, and the pass size it was found for
    StateSet set_a, set_b;
    DevBuf<float4> d_hit_tb;
    DevBuf<uint2> d_hit_prim;
    DevBuf<float4> d_shadow;  // q0 | q1 | q2,q3 contiguous (4 * capacity)
    DevBuf<RoundCounters> d_counters;
    DevBuf<uint32_t> d_top_counters;  // per round: entries queued from the top of the state set (guided: predicted-specular survivors)
    PinnedVec<uint32_t> h_top_counters;  // (page-locked, like h_counters: the host's looks at the queue length wait for these copies)
    DevBuf<double> d_film;         // kFilmStride doubles per pixel, film_copies films back to back
    uint32_t film_copies = 1;
    DevBuf<double> d_film_packed;  // RGB, what the ABI returns
    DevBuf<float> d_lambda_u;
    DevBuf<float4> d_spec_tab;
    SpecTab spec{nullptr, 0, 0};
    uint32_t n_materials = 0, n_emissions = 0;  // entries of the device materials / emissions arrays
    DevBuf<float4> d_lambda_tab;  // 4 float4 per table entry: lambdas, pdfs
    PinnedVec<RoundCounters> h_counters;
    bool profiling = false;
    // HIP-event pool: each launch records a start/stop pair on the library's stream without waiting; the pairs are
    // resolved once per render call, so profiling does not serialise the host with the device.
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_stage;  // stage of pair i (events 2i, 2i+1)
    bool census_on = false;
    int kinds = 7;       // material kinds in use, rounded up to a compiled set (PT_LAUNCH_SHADE)
    std::map<const void*, int> resident_blocks;  // per shade kernel: workgroups resident per CU (resident_grid)
    size_t bvh_lds = 0;  // dynamic LDS bytes of every kernel that walks the BVH (pt_device.h: bvh_intersect), for the mode in use
    size_t bvh_lds_mode[2] = {0, 0};  // binary walk, wide walk
    bool wide = false;     // walk the four-wide image of the BVH (same hits; ptgpu_set_traversal)
    bool have_wide = false;
    DevBuf<float4> d_wnodes;
    DevBuf<unsigned long long> d_census;  // extend nodes, extend tris, occlude nodes, occlude tris
    DevBuf<uint32_t> d_guards;            // kGuard* counters (pt_types.h), cumulative since ptgpu_init
    double stage_ms[PTGPU_N_STAGES] = {0, 0, 0, 0, 0, 0};
    uint64_t stage_launches[PTGPU_N_STAGES] = {0, 0, 0, 0, 0, 0};
    uint64_t launches = 0;
    GuidedState* guided = nullptr;
    GuidedPersist* guided_persist = nullptr;
    int bvh_builder = 2;               // ptgpu_set_bvh_builder: 0 host, 1 device, 2 device for scenes of kDeviceBuildFrom triangles and more
    bool bvh_built_on_device = false;  // how the resident scene's tree was built
    bool has_passthrough = false;      // the resident scene has surfaces without a material (material_index < 0): walks may take more rounds than max_depth
    bool train_last_wave = true;       // ptgpu_set_final_training
    bool direct_splat = false;         // PTGPU_DIRECT_SPLAT, sampled at the start of a wave
    uint32_t nonfinite_seen = 0;       // kGuardNonFinite as of the last film download
    uint64_t nonfinite_last = 0;       // ... and how many of them the render that film belongs to dropped (ptgpu_stats::nonfinite_samples)
    bool capture_records = false;      // ptgpu_guided_capture_records (parity hook)
    std::vector<uint32_t> captured;    // 14 words per record of the last wave traced with the hook on
    bool record_overlap = getenv("PTGPU_NO_OVERLAP") == nullptr;  // guided: record pass on the second stream (ptgpu_set_record_overlap)
    bool film_ready = false;           // d_film_packed holds the film of the last completed render
    unsigned char* scratch = nullptr;  // device address of `scratch` on this context's GPU (hipGetSymbolAddress)
};

namespace {

// Shade kernels are compiled for three material-kind sets: diffuse only (1), diffuse + dielectric (3), all (7).
// The shade kernels hold 2-4 workgroups per CU (registers), not the 8 the stage grid is sized for: they are launched with
// exactly as many workgroups as are resident at once, so the grid-stride loop has no second, partly filled round (a 3-wave
// kernel runs 3 + 3 + 2 otherwise: measured 3-4 % of the kernel on the Cornell box, 4 blocks per CU 17 % slower than 3).
#define PT_LAUNCH_ONE_KIND(c, kernel, grid, ...)                                                                          \
    hipLaunchKernelGGL(kernel, dim3(resident_grid((c), (const void*)kernel, (grid))), dim3(kBlock), 0, (c)->stream, __VA_ARGS__)

int resident_grid(ptgpu_ctx* c, const void* kernel, int grid, size_t dynamic_lds = 0);
// A shade kernel by material-kind set and, where the scene is small enough, with the scene in LDS (pt_kernels.h: SmallSceneLds)
#define PT_LAUNCH_SHADE_S(c, kernel, SMALL, grid, ...)                                    \
    do {                                                                                   \
        auto k1_ = kernel<1, SMALL>;                                                       \
        auto k3_ = kernel<3, SMALL>;                                                       \
        auto k7_ = kernel<7, SMALL>;                                                       \
        if ((c)->kinds == 1) PT_LAUNCH_ONE_KIND(c, k1_, grid, __VA_ARGS__);                \
        else if ((c)->kinds == 3) PT_LAUNCH_ONE_KIND(c, k3_, grid, __VA_ARGS__);           \
        else PT_LAUNCH_ONE_KIND(c, k7_, grid, __VA_ARGS__);                                \
    } while (0)
#define PT_LAUNCH_SHADE(c, kernel, grid, ...)                                                                                \
    do {                                                                                                                      \
        static const bool no_small = getenv("PTGPU_NO_SMALL_SCENE") != nullptr; /* A/B */                                     \
        if (!no_small && scene_is_small((c)->scene.n_tris, (c)->scene.n_emissive_tris, (c)->scene.n_materials))               \
            PT_LAUNCH_SHADE_S(c, kernel, true, grid, __VA_ARGS__);                                                            \
        else PT_LAUNCH_SHADE_S(c, kernel, false, grid, __VA_ARGS__);                                                          \
    } while (0)
int resident_grid(ptgpu_ctx* c, const void* kernel, int grid, size_t dynamic_lds) {
    auto it = c->resident_blocks.find(kernel);
    if (it == c->resident_blocks.end()) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, dynamic_lds) != hipSuccess || per_cu < 1) per_cu = dynamic_lds ? 1 : 8;
        it = c->resident_blocks.emplace(kernel, std::min(per_cu, 8)).first;
    }
    return std::min(grid, c->n_cus * it->second);
}
// Kernels that walk the BVH keep their traversal stacks (and small node arrays) in dynamic LDS sized from the scene's
// tree depth (c->bvh_lds, 15 KB for the Cornell box, 35 KB at depth 25, 74 KB at the reference's limit of 64 levels): the
// grid is what is resident with THAT much LDS per workgroup (the cache is per scene: ptgpu_scene_create clears it).
// Every kernel that walks the BVH exists for the binary and for the four-wide image of the tree (ptgpu_set_traversal).
#define PT_WALKER(c, kernel, census) ((c)->wide ? kernel<census, true> : kernel<census, false>)
#define PT_LAUNCH_WALK(c, kernel, grid, ...) \
    hipLaunchKernelGGL(kernel, dim3(resident_grid((c), (const void*)kernel, (grid), (c)->bvh_lds)), dim3(kBlock), (c)->bvh_lds, (c)->stream, __VA_ARGS__)
// Above 64 KB per workgroup a kernel must be told so before its first launch (up to the CU's 160 KB).
constexpr size_t kLdsDefaultLimit = 64 * 1024, kLdsPerCu = 160 * 1024;
// Default traversal of a new scene unless PTGPU_BVH_WIDE says otherwise: the four-wide walk for trees whose binary nodes
// exceed this many bytes. Measured (profiles/r02_traversal_modes.txt): where the nodes sit in LDS or L1 the binary walk's
// cheaper steps win (Cornell box: extend 8.1 ms against 9.1), where every step is a round trip to L2 the wide walk's
// fewer steps do (veach-mis, 61 492 triangles: extend 25.2 -> 22.2 ms, occlusion 22.9 -> 20.0); veach-bidir (0.7 MB) was even
// then and is well ahead since the wide walkers take their queues as streams (path 211 -> 266, guided 140 -> 161 Msamples/s at
// 64 spp, tools/time_render.py --scene veach-bidir), Cornell still 2 % behind. Nothing in between was measured: the line sits at the L2 side of it.
constexpr size_t kWideAboveBytes = 1u << 18;

int grid_for(const ptgpu_ctx* c, uint64_t n) {
    const uint64_t want = (n + kBlock - 1) / kBlock;
    const uint64_t cap = (uint64_t)c->n_cus * 8;  // <= 8 blocks of 256 per CU, grid-stride the rest
    return (int)std::max<uint64_t>(1, std::min(want, cap));
}

SpectrumDev to_dev(const ptgpu_spectrum& s, int illuminant) { return SpectrumDev{s.c0, s.c1, s.c2, s.scale, illuminant}; }

int ensure_workspace(ptgpu_ctx* c, uint32_t cap) {
    if (c->capacity == cap && c->set_a.cols.p) return PTGPU_OK;
    HIP_TRY(c->set_a.cols.alloc(8 * (size_t)cap));
    HIP_TRY(c->set_b.cols.alloc(8 * (size_t)cap));
    HIP_TRY(c->d_hit_tb.alloc(cap));
    HIP_TRY(c->d_hit_prim.alloc(cap));
    HIP_TRY(c->d_shadow.alloc(4 * (size_t)cap));
    HIP_TRY(c->d_counters.alloc(kMaxRounds + 2));
    c->h_counters.resize(kMaxRounds + 2);
    HIP_TRY(c->d_top_counters.alloc(kMaxRounds + 2));
    c->h_top_counters.assign(kMaxRounds + 2, 0u);
    c->capacity = cap;
    return PTGPU_OK;
}
// ensure_workspace for `cap` paths or, where the device has not that much memory left, for the largest whole multiple
// of `unit` paths (one sample pass of the image) that it has, halving until it fits.
// `headroom`: bytes that must stay free after it (the film and the tables are allocated later; at least 2 GB).
int ensure_workspace_or_smaller(ptgpu_ctx* c, uint64_t cap, uint64_t unit, size_t headroom) {
    // a context that had to fall back starts the next render of the same image at the size that fit, instead of freeing
    // the working workspace and failing its way down from the top again (multi-GB hipFree / hipMalloc per render)
    if (c->fallback_unit == unit && c->fallback_cap != 0 && cap > c->fallback_cap) cap = std::max<uint64_t>(unit, (c->fallback_cap / unit) * unit);
    if (c->capacity == cap && c->set_a.cols.p) return PTGPU_OK;  // the workspace of the previous render
    headroom = std::max(headroom, (size_t)2 << 30);
    for (;;) {
        const int rc = ensure_workspace(c, (uint32_t)cap);
        size_t free_b = 0, total_b = 0;
        if (rc == PTGPU_OK && (cap <= unit || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b >= headroom)) return rc;
        const bool out_of_memory = rc == PTGPU_OK || hipGetLastError() == hipErrorOutOfMemory;
        c->capacity = 0;  // whatever was allocated is not a consistent workspace
        if (!out_of_memory || cap <= unit) return rc;
        cap = std::max(unit, (cap / 2 / unit) * unit);
        c->fallback_cap = cap;
        c->fallback_unit = unit;
    }
}
ShadowCols shadow_view(ptgpu_ctx* c) {
    ShadowCols s;
    const size_t cap = c->capacity;
    s.q0 = c->d_shadow.p;
    s.q1 = c->d_shadow.p + cap;
    s.q2 = c->d_shadow.p + 2 * cap;  // q2 and q3 are contiguous: lightpath keeps 3 doubles per entry there
    s.q3 = c->d_shadow.p + 3 * cap;
    return s;
}

// Brackets one kernel launch with a pair of HIP events on the library's stream when profiling is on.
struct StageTimer {
    ptgpu_ctx* c;
    int stage;
    size_t pair = 0;
    bool on = false;
    hipStream_t s;
    StageTimer(ptgpu_ctx* c_, int stage_, hipStream_t stream_ = nullptr) : c(c_), stage(stage_), s(stream_ ? stream_ : c_->stream) {
        if (!c->profiling) return;
        pair = c->ev_stage.size();
        while (c->ev_pool.size() < 2 * (pair + 1)) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->ev_pool.push_back(e);
        }
        c->ev_stage.push_back(stage);
        on = true;
        (void)hipEventRecord(c->ev_pool[2 * pair], s);
    }
    ~StageTimer() {
        c->launches++;
        c->stage_launches[stage]++;
        if (on) (void)hipEventRecord(c->ev_pool[2 * pair + 1], s);
    }
};
// Turns the recorded event pairs into per-stage totals; call after the stream has been synchronised.
void resolve_stage_times(ptgpu_ctx* c) {
    for (size_t i = 0; i < c->ev_stage.size(); i++) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_pool[2 * i], c->ev_pool[2 * i + 1]) == hipSuccess) c->stage_ms[c->ev_stage[i]] += ms;
    }
    c->ev_stage.clear();
}
void reset_stage_times(ptgpu_ctx* c) {
    for (int i = 0; i < PTGPU_N_STAGES; i++) {
        c->stage_ms[i] = 0.0;
        c->stage_launches[i] = 0;
    }
    c->launches = 0;
    c->ev_stage.clear();
}

// Wavelength tables. The reference draws the hero-wavelength sample from a per-pixel additive recurrence
// (Sampler::wavelength_sample, sampling.rs:162-193) whose state is carried sequentially: for `path` and
// `lightpath` sample i of every pixel sees the (i+1)-th iterate of frac(state + golden); for `guided` the
// value depends only on the pixel's position inside its 16x16 tile (path_guiding/mod.rs:89-94; see
// build_guided_lambda_table). The tables hold the same f32 iterates the sequential code produces.
std::vector<float> golden_iterates(size_t n) {
    const float nn = 1.0f;
    float inc = (nn + sqrtf(nn * nn + 4.0f)) / 2.0f;
    inc = inc - floorf(inc);
    std::vector<float> t(n);
    float state = 0.0f;
    for (size_t i = 0; i < n; i++) {
        state += inc;
        state -= floorf(state);
        t[i] = state;
    }
    return t;
}

}  // namespace

extern "C" {

const char* ptgpu_last_error(void) { return g_err.c_str(); }

int ptgpu_init(ptgpu_ctx** out, int device_id) {
    if (!out) return fail(PTGPU_E_INVALID, "ptgpu_init: out is NULL");
    *out = nullptr;
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device_id < 0 || device_id >= n_dev) return fail(PTGPU_E_DEVICE, "ptgpu_init: no such HIP device");
    HIP_TRY(hipSetDevice(device_id));
    ptgpu_ctx* c = new ptgpu_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) c->n_cus = prop.multiProcessorCount;
    // the main stream carries the critical path (the walk); the guided record pass fills in behind it at low priority
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_least);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_least);
    if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->stream4, hipStreamNonBlocking, prio_least);
    for (int k = 0; k < ptgpu_ctx::kMaxRecordSlices; k++) {
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_scanned[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_grouped[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_splatted, hipEventDisableTiming);
    if (e == hipSuccess) e = hipGetSymbolAddress((void**)&c->scratch, HIP_SYMBOL(scratch));  // this device's copy of the array
    if (e == hipSuccess) e = c->d_census.alloc(4);
    if (e == hipSuccess) e = hipMemset(c->d_census.p, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = c->d_guards.alloc(kNumGuards);
    if (e == hipSuccess) e = hipMemset(c->d_guards.p, 0, kNumGuards * sizeof(uint32_t));
    if (e == hipSuccess) e = c->d_d65.upload(std::vector<float>(kD65, kD65 + CIE_N), c->stream);
    if (e == hipSuccess) e = c->d_xyz.upload(std::vector<double>(&kXyz[0][0], &kXyz[0][0] + 3 * CIE_N), c->stream);
    if (e != hipSuccess) {
        delete c;
        return fail(PTGPU_E_DEVICE, std::string("ptgpu_init: ") + hipGetErrorString(e));
    }
    *out = c;
    return PTGPU_OK;
}

void ptgpu_destroy(ptgpu_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->stream3) (void)hipStreamSynchronize(c->stream3);
    if (c->stream4) (void)hipStreamSynchronize(c->stream4);
    guided_free(c->guided);
    guided_persist_free(c->guided_persist);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->stream4) (void)hipStreamDestroy(c->stream4);
    for (int k = 0; k < ptgpu_ctx::kMaxRecordSlices; k++) {
        if (c->ev_scanned[k]) (void)hipEventDestroy(c->ev_scanned[k]);
        if (c->ev_grouped[k]) (void)hipEventDestroy(c->ev_grouped[k]);
    }
    if (c->ev_splatted) (void)hipEventDestroy(c->ev_splatted);
    delete c;
}

namespace {
constexpr size_t kDeviceBuildFrom = 16384;  // below this the single-core host build takes a few milliseconds
// Bvh::new on the device (pt_bvh_build.h). `built` = false: the input is one the device builder leaves to the host
// (negative zeros, non-finite bounds, 2^24 triangles or more, an absurdly deep tree); `tris` is permuted as build_bvh does.
int build_bvh_device(ptgpu_ctx* c, std::vector<BuildTri>& tris, FlatBvh& out, bool& built, bool& leaf_too_big) {
    built = false;
    leaf_too_big = false;
    const size_t n = tris.size();
    if (n == 0 || n >= (1u << 24)) return PTGPU_OK;
    std::vector<float4> lo(n), hi(n);
    for (size_t i = 0; i < n; i++) {
        const Box3& b = tris[i].bounds;
        const float v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
        for (float x : v)
            if (!std::isfinite(x) || (x == 0.0f && std::signbit(x))) return PTGPU_OK;
        lo[i] = make_float4(b.lo.x, b.lo.y, b.lo.z, u2f(tris[i].id));
        hi[i] = make_float4(b.hi.x, b.hi.y, b.hi.z, 0.0f);
    }
    DevBuf<float4> d_lo, d_hi, d_out;
    DevBuf<BuildNode> d_nodes;
    DevBuf<uint32_t> d_a, d_b, d_words;  // d_words: node counter, error flag
    HIP_TRY(d_lo.upload(lo, c->stream));
    HIP_TRY(d_hi.upload(hi, c->stream));
    const uint32_t max_nodes = (uint32_t)(2 * n);
    HIP_TRY(d_nodes.alloc(max_nodes));
    HIP_TRY(d_a.alloc(n));
    HIP_TRY(d_b.alloc(n));
    HIP_TRY(d_words.alloc(2));
    HIP_TRY(hipMemsetAsync(d_nodes.p, 0, (size_t)max_nodes * sizeof(BuildNode), c->stream));
    BuildNode root{};
    root.begin = 0;
    root.end = (uint32_t)n;
    uint32_t words[2] = {1u, 0u};
    HIP_TRY(hipMemcpyAsync(d_nodes.p, &root, sizeof(root), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_words.p, words, sizeof(words), hipMemcpyHostToDevice, c->stream));
    BuildArgs a{d_lo.p, d_hi.p, d_nodes.p, d_words.p, max_nodes, d_a.p, d_b.p, d_words.p + 1};
    std::vector<std::pair<uint32_t, uint32_t>> levels;  // first node, nodes
    uint32_t first = 0, count = 1;
    while (count > 0) {
        if (levels.size() > 4096) return PTGPU_OK;  // the host builder's recursion is what such a tree gets
        levels.emplace_back(first, count);
        hipLaunchKernelGGL(k_bvh_split, dim3(count), dim3(kBuildBlock), 0, c->stream, a, first);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(words, d_words.p, sizeof(words), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (words[1] != 0) return fail(PTGPU_E_DEVICE, "scene_create: device BVH build failed (internal error " + std::to_string(words[1]) + ")");
        first += count;
        count = words[0] - first;
    }
    const uint32_t n_nodes = words[0];
    for (size_t l = levels.size(); l-- > 0;)
        hipLaunchKernelGGL(k_bvh_count, dim3((levels[l].second + 255) / 256), dim3(256), 0, c->stream, d_nodes.p, levels[l].first, levels[l].second);
    for (size_t l = 0; l < levels.size(); l++)
        hipLaunchKernelGGL(k_bvh_number, dim3((levels[l].second + 255) / 256), dim3(256), 0, c->stream, d_nodes.p, levels[l].first, levels[l].second);
    HIP_TRY(d_out.alloc(2 * (size_t)n_nodes));
    hipLaunchKernelGGL(k_bvh_emit, dim3((n_nodes + 255) / 256), dim3(256), 0, c->stream, d_nodes.p, n_nodes, d_out.p, d_words.p + 1);
    HIP_TRY(hipGetLastError());
    std::vector<float4> flat(2 * (size_t)n_nodes);
    HIP_TRY(hipMemcpyAsync(flat.data(), d_out.p, flat.size() * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(lo.data(), d_lo.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(hi.data(), d_hi.p, n * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(words, d_words.p, sizeof(words), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    built = true;
    if (words[1] == 3) {
        leaf_too_big = true;
        return PTGPU_OK;
    }
    out.nodes.resize(n_nodes);
    for (uint32_t i = 0; i < n_nodes; i++) {
        const float4 l = flat[2 * (size_t)i], h = flat[2 * (size_t)i + 1];
        const uint32_t meta = f2u(h.w);
        out.nodes[i] = FlatNode{Box3{mk3(l.x, l.y, l.z), mk3(h.x, h.y, h.z)}, f2u(l.w), (uint16_t)(meta & 0xffffu), (uint8_t)(meta >> 16)};
    }
    out.leaf_order.resize(n);
    for (size_t i = 0; i < n; i++) {
        out.leaf_order[i] = f2u(lo[i].w);
        tris[i] = BuildTri{Box3{mk3(lo[i].x, lo[i].y, lo[i].z), mk3(hi[i].x, hi[i].y, hi[i].z)}, f2u(lo[i].w)};
    }
    return PTGPU_OK;
}
}  // namespace

int ptgpu_set_bvh_builder(ptgpu_ctx* c, int mode) {
    if (!c || mode < 0 || mode > 2) return fail(PTGPU_E_INVALID, "set_bvh_builder: mode is 0 (host), 1 (device) or 2 (by scene size)");
    c->bvh_builder = mode;
    return PTGPU_OK;
}
int ptgpu_bvh_built_on_device(ptgpu_ctx* c) { return (c && c->have_scene && c->bvh_built_on_device) ? 1 : 0; }

int ptgpu_scene_create(ptgpu_ctx* c, const ptgpu_scene_desc* desc) {
    if (!c || !desc || !desc->meshes || desc->n_meshes == 0) return fail(PTGPU_E_INVALID, "scene_create: empty description");
    HIP_TRY(hipSetDevice(c->device));
    // Scene::new (scene.rs:94-171): emissive meshes first, each group in input order.
    std::vector<uint32_t> order;
    uint32_t n_emissive_meshes = 0;
    for (uint32_t i = 0; i < desc->n_meshes; i++)
        if (desc->meshes[i].emissive) order.push_back(i);
    n_emissive_meshes = (uint32_t)order.size();
    for (uint32_t i = 0; i < desc->n_meshes; i++)
        if (!desc->meshes[i].emissive) order.push_back(i);

    bool any_passthrough = false;
    std::vector<BuildTri> build;
    std::vector<uint32_t> tri_mesh, tri_index;  // per input triangle id
    std::vector<MeshInfo> mesh_info(order.size());
    std::vector<SpectrumDev> emissions;
    std::vector<uint32_t> emissive_ids;  // input triangle ids of emissive triangles, Scene::emissive_triangles order
    for (uint32_t m = 0; m < order.size(); m++) {
        const ptgpu_mesh& mesh = desc->meshes[order[m]];
        if (!mesh.points || !mesh.triangles) return fail(PTGPU_E_INVALID, "scene_create: mesh without points/triangles");
        if (mesh.material_index >= (int32_t)desc->n_materials) return fail(PTGPU_E_INVALID, "scene_create: material index out of range");
        mesh_info[m].material = mesh.material_index < 0 ? -1 : mesh.material_index;
        if (mesh.material_index < 0 && mesh.n_triangles > 0) any_passthrough = true;
        mesh_info[m].has_normals = mesh.normals ? 1 : 0;
        if (m < n_emissive_meshes) emissions.push_back(to_dev(mesh.emission, 1));
        for (uint32_t t = 0; t < mesh.n_triangles; t++) {
            const uint32_t i = mesh.triangles[3 * t], j = mesh.triangles[3 * t + 1], k = mesh.triangles[3 * t + 2];
            if (i >= mesh.n_points || j >= mesh.n_points || k >= mesh.n_points) return fail(PTGPU_E_INVALID, "scene_create: vertex index out of range");
            const float* P = mesh.points;
            const V3 p0 = mk3(P[3 * i], P[3 * i + 1], P[3 * i + 2]), p1 = mk3(P[3 * k], P[3 * k + 1], P[3 * k + 2]),
                     p2 = mk3(P[3 * j], P[3 * j + 1], P[3 * j + 2]);
            Box3 b{min3(p0, p1), max3(p0, p1)};
            b = box_join(b, p2);
            if (m < n_emissive_meshes) emissive_ids.push_back((uint32_t)build.size());
            build.push_back(BuildTri{b, (uint32_t)build.size()});
            tri_mesh.push_back(m);
            tri_index.push_back(t);
        }
    }
    if (build.empty()) return fail(PTGPU_E_INVALID, "scene_create: no triangles");
    FlatBvh bvh;
    bool on_device = false, leaf_too_big = false;
    if (c->bvh_builder == 1 || (c->bvh_builder == 2 && build.size() >= kDeviceBuildFrom)) {
        const int rc = build_bvh_device(c, build, bvh, on_device, leaf_too_big);
        if (rc != PTGPU_OK) return rc;
    }
    if (!on_device) leaf_too_big = !build_bvh(build, bvh);
    if (leaf_too_big) return fail(PTGPU_E_LIMIT, "scene_create: a BVH leaf exceeds 65535 triangles");
    c->bvh_built_on_device = on_device;

    const size_t n_tris = bvh.leaf_order.size(), n_nodes = bvh.nodes.size();
    // deepest leaf: the traversal keeps at most one waiting sibling per level
    uint32_t bvh_depth = 0;
    {
        std::vector<std::pair<uint32_t, uint32_t>> todo{{0u, 0u}};
        while (!todo.empty()) {
            const auto [ni, depth] = todo.back();
            todo.pop_back();
            bvh_depth = std::max(bvh_depth, depth);
            if (bvh.nodes[ni].n_tris == 0) {
                todo.emplace_back(ni + 1, depth + 1);
                todo.emplace_back(bvh.nodes[ni].offset, depth + 1);
            }
        }
    }
    if (bvh_depth + 1 > (uint32_t)kBvhStack) return fail(PTGPU_E_LIMIT, "scene_create: BVH deeper than the reference's 64-entry traversal stack (bvh.rs:94)");
    std::vector<float4> nodes(2 * n_nodes), tri_v(3 * n_tris), tri_n(3 * n_tris, make_float4(0, 0, 0, 0));
    c->h_nodes_words.assign(8 * n_nodes, 0);
    for (size_t i = 0; i < n_nodes; i++) {
        const FlatNode& n = bvh.nodes[i];
        const uint32_t meta = (uint32_t)n.n_tris | ((uint32_t)n.split_axis << 16);
        nodes[2 * i] = make_float4(n.bounds.lo.x, n.bounds.lo.y, n.bounds.lo.z, u2f(n.offset));
        nodes[2 * i + 1] = make_float4(n.bounds.hi.x, n.bounds.hi.y, n.bounds.hi.z, u2f(meta));
        uint32_t* w = &c->h_nodes_words[8 * i];
        w[0] = f2u(n.bounds.lo.x); w[1] = f2u(n.bounds.lo.y); w[2] = f2u(n.bounds.lo.z);
        w[3] = f2u(n.bounds.hi.x); w[4] = f2u(n.bounds.hi.y); w[5] = f2u(n.bounds.hi.z);
        w[6] = n.offset; w[7] = meta;
    }
    std::vector<uint32_t> leaf_of_input(n_tris);
    c->h_leaf_tris.assign(2 * n_tris, 0);
    bool any_normals = false;
    for (size_t k = 0; k < n_tris; k++) {
        const uint32_t id = bvh.leaf_order[k];
        leaf_of_input[id] = (uint32_t)k;
        const uint32_t m = tri_mesh[id], t = tri_index[id];
        const ptgpu_mesh& mesh = desc->meshes[order[m]];
        const uint32_t idx[3] = {mesh.triangles[3 * t], mesh.triangles[3 * t + 1], mesh.triangles[3 * t + 2]};
        // third tag: material + 1 (0 = pass-through surface) | kind << 28 | has_normals << 31, so that shading a hit needs no
        // dependent mesh_info load after the triangle fetch
        const int mat = mesh_info[m].material;
        const uint32_t kind = mat >= 0 ? (uint32_t)desc->materials[mat].kind & 3u : 0u;
        const uint32_t tag[3] = {m, t, (uint32_t)(mat + 1) | (kind << kTagKindShift) | (mesh.normals ? kTagNormalsBit : 0u)};
        for (int v = 0; v < 3; v++) {
            const float* p = mesh.points + 3 * (size_t)idx[v];
            tri_v[3 * k + v] = make_float4(p[0], p[1], p[2], u2f(tag[v]));
            if (mesh.normals) {
                const float* nn = mesh.normals + 3 * (size_t)idx[v];
                tri_n[3 * k + v] = make_float4(nn[0], nn[1], nn[2], 0.0f);
                any_normals = true;
            }
        }
        c->h_leaf_tris[2 * k] = m;
        c->h_leaf_tris[2 * k + 1] = t;
    }
    if (!any_normals) tri_n.assign(1, make_float4(0, 0, 0, 0));
    std::vector<uint32_t> emissive_leaf(emissive_ids.size());
    for (size_t i = 0; i < emissive_ids.size(); i++) emissive_leaf[i] = leaf_of_input[emissive_ids[i]];
    std::vector<MaterialDev> mats(desc->n_materials);
    for (uint32_t i = 0; i < desc->n_materials; i++) {
        const ptgpu_material& m = desc->materials[i];
        if (m.kind < 0 || m.kind > 2) return fail(PTGPU_E_INVALID, "scene_create: unknown material kind");
        mats[i] = MaterialDev{m.kind, m.roughness, to_dev(m.s0, 0), to_dev(m.s1, 0)};
    }
    {
        int used = 0;
        for (const MeshInfo& mi : mesh_info)
            if (mi.material >= 0) used |= 1 << mats[mi.material].kind;
        c->kinds = (used & (1 << MAT_CONDUCTOR)) ? 7 : ((used & (1 << MAT_DIELECTRIC)) ? 3 : 1);
    }
    if (mats.empty()) mats.push_back(MaterialDev{0, 0.0f, SpectrumDev{0, 0, 0, 0, 0}, SpectrumDev{0, 0, 0, 0, 0}});
    if (emissions.empty()) emissions.push_back(SpectrumDev{0, 0, 0, 0, 1});
    if (emissive_leaf.empty()) emissive_leaf.push_back(0);

    HIP_TRY(c->d_nodes.upload(nodes, c->stream));
    HIP_TRY(c->d_tri_v.upload(tri_v, c->stream));
    HIP_TRY(c->d_tri_n.upload(tri_n, c->stream));
    HIP_TRY(c->d_mesh_info.upload(mesh_info, c->stream));
    {
        std::vector<float4> light_v(3 * emissive_leaf.size());
        for (size_t i = 0; i < emissive_leaf.size(); i++)
            for (int v = 0; v < 3; v++) light_v[3 * i + v] = tri_v[3 * (size_t)emissive_leaf[i] + v];
        HIP_TRY(c->d_light_v.upload(light_v, c->stream));
    }
    HIP_TRY(c->d_materials.upload(mats, c->stream));
    HIP_TRY(c->d_emissions.upload(emissions, c->stream));
    c->n_materials = (uint32_t)mats.size();
    c->has_passthrough = any_passthrough;
    c->n_emissions = (uint32_t)emissions.size();
    HIP_TRY(c->d_emissive.upload(emissive_leaf, c->stream));
    SceneDev& s = c->scene;
    s.nodes = c->d_nodes.p;
    s.tri_v = c->d_tri_v.p;
    s.tri_n = c->d_tri_n.p;
    s.mesh_info = c->d_mesh_info.p;
    s.light_v = c->d_light_v.p;
    s.materials = c->d_materials.p;
    s.emissions = c->d_emissions.p;
    s.emissive_tris = c->d_emissive.p;
    s.d65 = c->d_d65.p;
    s.cie_xyz = c->d_xyz.p;
    s.guards = c->d_guards.p;
    s.n_nodes = (uint32_t)n_nodes;
    s.n_tris = (uint32_t)n_tris;
    s.n_emissive_tris = (uint32_t)emissive_ids.size();
    s.n_materials = (uint32_t)mats.size();
    s.n_emissive_meshes = n_emissive_meshes;
    s.bvh_stack = bvh_depth + 1;
    s.lds_nodes = (n_nodes * 32 <= kLdsNodeBudget) ? 1u : 0u;
    c->bvh_lds_mode[0] = (size_t)kBlock * (s.bvh_stack + 2 * kLeafQueue) * sizeof(uint32_t) + (s.lds_nodes ? n_nodes * 32 : 0);
    {
        // the four-wide image of the same tree (bvh_build.h)
        std::vector<WideNode> wide;
        uint32_t wide_depth = 0;
        c->have_wide = build_wide(bvh, wide, wide_depth);
        if (!c->have_wide) wide.assign(1, WideNode{});
        std::vector<float4> wn(8 * wide.size());
        memcpy(wn.data(), wide.data(), wide.size() * sizeof(WideNode));
        HIP_TRY(c->d_wnodes.upload(wn, c->stream));
        s.wnodes = c->d_wnodes.p;
        s.n_wnodes = (uint32_t)wide.size();
        s.wide_stack = 3 * wide_depth + 1;
        s.lds_wnodes = (wide.size() * sizeof(WideNode) <= kLdsNodeBudget) ? 1u : 0u;
        c->bvh_lds_mode[1] = (size_t)kBlock * (s.wide_stack + kLeafQueue) * sizeof(uint32_t) + (s.lds_wnodes ? wide.size() * sizeof(WideNode) : 0);
        if (c->bvh_lds_mode[1] > kLdsPerCu) c->have_wide = false;
    }
    static const char* env_wide = getenv("PTGPU_BVH_WIDE");  // default traversal of new scenes: 1 = four-wide, 0 = binary
    c->wide = c->have_wide && (env_wide ? atoi(env_wide) != 0 : n_nodes * 32 > kWideAboveBytes);
    c->bvh_lds = c->bvh_lds_mode[c->wide ? 1 : 0];
    c->resident_blocks.clear();
    if (c->bvh_lds_mode[0] > kLdsPerCu) return fail(PTGPU_E_LIMIT, "scene_create: traversal stacks exceed the CU's LDS");
    if (std::max(c->bvh_lds_mode[0], c->have_wide ? c->bvh_lds_mode[1] : 0) > kLdsDefaultLimit) {
        const int lds_b = (int)c->bvh_lds_mode[0], lds_w = c->have_wide ? (int)std::max(c->bvh_lds_mode[1], (size_t)1) : 1;  // (no wide image: its kernels are never launched)
#define PT_SET_LDS(kernel)                                                                                             \
    HIP_TRY(hipFuncSetAttribute((const void*)kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b)); \
    HIP_TRY(hipFuncSetAttribute((const void*)kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b)); \
    HIP_TRY(hipFuncSetAttribute((const void*)kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));  \
    HIP_TRY(hipFuncSetAttribute((const void*)kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        PT_SET_LDS(k_trace_batch)
        PT_SET_LDS(k_extend)
        PT_SET_LDS(k_occlude_path)
        PT_SET_LDS(k_occlude_splat)
        PT_SET_LDS(k_occlude_guided)
        PT_SET_LDS(k_walk_guided)
#undef PT_SET_LDS
        HIP_TRY(hipFuncSetAttribute((const void*)k_extend_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_extend_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_path_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_path_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_splat_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_splat_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_guided_stream<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occlude_guided_stream<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occluded_batch<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b));
        HIP_TRY(hipFuncSetAttribute((const void*)k_occluded_batch<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_w));
    }
    s.bounds_lo = bvh.nodes[0].bounds.lo;
    s.bounds_hi = bvh.nodes[0].bounds.hi;
    c->have_scene = true;
    return PTGPU_OK;
}

int ptgpu_camera_set(ptgpu_ctx* c, const ptgpu_camera_desc* cam) {
    if (!c || !cam || cam->resx == 0 || cam->resy == 0) return fail(PTGPU_E_INVALID, "camera_set: bad arguments");
    if ((uint64_t)cam->resx * cam->resy > (1ull << 31)) return fail(PTGPU_E_LIMIT, "camera_set: image too large");
    CameraDev& d = c->camera;
    for (int r = 0; r < 4; r++)
        for (int col = 0; col < 4; col++) d.w2c.c[col][r] = cam->world_to_camera[4 * r + col];
    // Matrix4::inverse via adjugate / determinant (math/src/matrix.rs:109-178), same expression order.
    {
        const float(*m)[4] = d.w2c.c;
        const float m00 = m[0][0], m01 = m[0][1], m02 = m[0][2], m03 = m[0][3], m10 = m[1][0], m11 = m[1][1], m12 = m[1][2], m13 = m[1][3];
        const float m20 = m[2][0], m21 = m[2][1], m22 = m[2][2], m23 = m[2][3], m30 = m[3][0], m31 = m[3][1], m32 = m[3][2], m33 = m[3][3];
        const float c00 = (m22 * m33) - (m32 * m23), c02 = (m12 * m33) - (m32 * m13), c03 = (m12 * m23) - (m22 * m13);
        const float c04 = (m21 * m33) - (m31 * m23), c06 = (m11 * m33) - (m31 * m13), c07 = (m11 * m23) - (m21 * m13);
        const float c08 = (m21 * m32) - (m31 * m22), c10 = (m11 * m32) - (m31 * m12), c11 = (m11 * m22) - (m21 * m12);
        const float c12 = (m20 * m33) - (m30 * m23), c14 = (m10 * m33) - (m30 * m13), c15 = (m10 * m23) - (m20 * m13);
        const float c16 = (m20 * m32) - (m30 * m22), c18 = (m10 * m32) - (m30 * m12), c19 = (m10 * m22) - (m20 * m12);
        const float c20 = (m20 * m31) - (m30 * m21), c22 = (m10 * m31) - (m30 * m11), c23 = (m10 * m21) - (m20 * m11);
        const float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
        const float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
        const float v0[4] = {m10, m00, m00, m00}, v1[4] = {m11, m01, m01, m01}, v2[4] = {m12, m02, m02, m02}, v3[4] = {m13, m03, m03, m03};
        const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
        float ad[4][4];
        for (int i = 0; i < 4; i++) {
            ad[0][i] = ((v1[i] * f0[i]) - (v2[i] * f1[i]) + (v3[i] * f2[i])) * sa[i];
            ad[1][i] = ((v0[i] * f0[i]) - (v2[i] * f3[i]) + (v3[i] * f4[i])) * sb[i];
            ad[2][i] = ((v0[i] * f1[i]) - (v1[i] * f3[i]) + (v3[i] * f5[i])) * sa[i];
            ad[3][i] = ((v0[i] * f2[i]) - (v1[i] * f4[i]) + (v2[i] * f5[i])) * sb[i];
        }
        const float d0 = m[0][0] * ad[0][0], d1 = m[0][1] * ad[1][0], d2 = m[0][2] * ad[2][0], d3 = m[0][3] * ad[3][0];
        const float recip = 1.0f / (d0 + d1 + d2 + d3);
        for (int col = 0; col < 4; col++)
            for (int r = 0; r < 4; r++) d.c2w.c[col][r] = ad[col][r] * recip;
    }
    const float half = (cam->fov_deg / 2.0f) * (kPi / 180.0f);  // f32::to_radians
    d.pixel_size = 2.0f * tanf(half) / (float)cam->resy;
    d.resx = cam->resx;
    d.resy = cam->resy;
    d.p = mat_point(d.c2w, mk3(0.0f, 0.0f, 0.0f));
    d.n = normalize(mat_vector(d.c2w, mk3(0.0f, 0.0f, 1.0f)));
    c->film_ready = false;
    c->have_camera = true;
    return PTGPU_OK;
}

int ptgpu_scene_info(ptgpu_ctx* c, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* n_emissive_tris, float bounds_out[6]) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "scene_info: no scene");
    if (n_nodes) *n_nodes = c->scene.n_nodes;
    if (n_tris) *n_tris = c->scene.n_tris;
    if (n_emissive_tris) *n_emissive_tris = c->scene.n_emissive_tris;
    if (bounds_out) {
        bounds_out[0] = c->scene.bounds_lo.x; bounds_out[1] = c->scene.bounds_lo.y; bounds_out[2] = c->scene.bounds_lo.z;
        bounds_out[3] = c->scene.bounds_hi.x; bounds_out[4] = c->scene.bounds_hi.y; bounds_out[5] = c->scene.bounds_hi.z;
    }
    return PTGPU_OK;
}
int ptgpu_scene_bvh(ptgpu_ctx* c, uint32_t* nodes_out, uint32_t* leaf_tris_out) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "scene_bvh: no scene");
    if (nodes_out) memcpy(nodes_out, c->h_nodes_words.data(), c->h_nodes_words.size() * 4);
    if (leaf_tris_out) memcpy(leaf_tris_out, c->h_leaf_tris.data(), c->h_leaf_tris.size() * 4);
    return PTGPU_OK;
}

int ptgpu_trace_batch(ptgpu_ctx* c, const float* rays, uint32_t n, ptgpu_hit* hits_out) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "trace_batch: no scene");
    if (n == 0) return PTGPU_OK;
    if (!rays || !hits_out) return fail(PTGPU_E_INVALID, "trace_batch: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    static_assert(sizeof(HitOut) == sizeof(ptgpu_hit), "hit layout");
    DevBuf<float> d_rays;
    DevBuf<HitOut> d_hits;
    HIP_TRY(d_rays.alloc(7 * (size_t)n));
    HIP_TRY(d_hits.alloc(n));
    HIP_TRY(hipMemcpyAsync(d_rays.p, rays, 7 * (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    // Also the extend micro-benchmark (tools/bench_extend.py): with profiling on, the walk's own time lands in
    // the EXTEND stage (ptgpu_kernel_times); with the census on, nodes popped / triangles tested land in ptgpu_census[0..1].
    reset_stage_times(c);
    {
        StageTimer st(c, PTGPU_STAGE_EXTEND);
        if (c->census_on)
            PT_LAUNCH_WALK(c, PT_WALKER(c, k_trace_batch, true), grid_for(c, n), c->scene, d_rays.p, n, d_hits.p, c->d_census.p);
        else
            PT_LAUNCH_WALK(c, PT_WALKER(c, k_trace_batch, false), grid_for(c, n), c->scene, d_rays.p, n, d_hits.p, (unsigned long long*)nullptr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hits_out, d_hits.p, (size_t)n * sizeof(HitOut), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    resolve_stage_times(c);
    return PTGPU_OK;
}

// ptgpu_trace_batch for callers that cannot allocate: same kernel, the staging buffers are the static scratch array.
// No hipMalloc / hipFree / event creation here (hence no stage timer and no census).
int ptgpu_trace_batch_noalloc(ptgpu_ctx* c, const float* rays, uint32_t n, ptgpu_hit* hits_out) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "trace_batch_noalloc: no scene");
    if (n == 0) return PTGPU_OK;
    if (!rays || !hits_out) return fail(PTGPU_E_INVALID, "trace_batch_noalloc: NULL buffer");
    if (n > kNoAllocMaxRays) return fail(PTGPU_E_LIMIT, "trace_batch_noalloc: at most 262144 rays per call (the static scratch buffer's size)");
    if (!c->scratch) return fail(PTGPU_E_DEVICE, "trace_batch_noalloc: scratch address unknown");
    HIP_TRY(hipSetDevice(c->device));
    float* d_rays = reinterpret_cast<float*>(c->scratch);
    HitOut* d_hits = reinterpret_cast<HitOut*>(c->scratch + kNoAllocRayBytes);
    // nothing in the scratch has to start at zero: the kernel writes every hit record in [0, n) and reads only rays [0, n)
    HIP_TRY(hipMemcpyAsync(d_rays, rays, 7 * (size_t)n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PT_LAUNCH_WALK(c, PT_WALKER(c, k_trace_batch, false), grid_for(c, n), c->scene, d_rays, n, d_hits, (unsigned long long*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hits_out, d_hits, (size_t)n * sizeof(HitOut), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

int ptgpu_occluded_batch(ptgpu_ctx* c, const float* from, const float* to, uint32_t n, uint8_t* out) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "occluded_batch: no scene");
    if (n == 0) return PTGPU_OK;
    if (!from || !to || !out) return fail(PTGPU_E_INVALID, "occluded_batch: NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf<float> d_from, d_to;
    DevBuf<uint8_t> d_out;
    HIP_TRY(d_from.alloc(3 * (size_t)n));
    HIP_TRY(d_to.alloc(3 * (size_t)n));
    HIP_TRY(d_out.alloc(n));
    HIP_TRY(hipMemcpyAsync(d_from.p, from, 3 * (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_to.p, to, 3 * (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    PT_LAUNCH_WALK(c, ((c)->wide ? k_occluded_batch<true> : k_occluded_batch<false>), grid_for(c, n), c->scene, d_from.p, d_to.p, n, d_out.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, d_out.p, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return PTGPU_OK;
}

int ptgpu_set_profiling(ptgpu_ctx* c, int enabled) {
    if (!c) return fail(PTGPU_E_INVALID, "set_profiling: NULL ctx");
    c->profiling = enabled != 0;
    return PTGPU_OK;
}
int ptgpu_set_final_training(ptgpu_ctx* c, int enabled) {
    if (!c) return fail(PTGPU_E_INVALID, "set_final_training: NULL ctx");
    if (c->guided) return fail(PTGPU_E_STATE, "set_final_training: not between guided_begin and guided_end");
    c->train_last_wave = enabled != 0;
    return PTGPU_OK;
}
int ptgpu_set_record_overlap(ptgpu_ctx* c, int enabled) {
    if (!c) return fail(PTGPU_E_INVALID, "set_record_overlap: NULL ctx");
    if (c->guided) return fail(PTGPU_E_STATE, "set_record_overlap: not between guided_begin and guided_end");
    c->record_overlap = enabled != 0;
    return PTGPU_OK;
}
int ptgpu_set_traversal(ptgpu_ctx* c, int wide) {
    if (!c || !c->have_scene) return fail(PTGPU_E_STATE, "set_traversal: no scene");
    if (wide && !c->have_wide) return fail(PTGPU_E_LIMIT, "set_traversal: this scene has no four-wide tree (a leaf with more than 128 triangles, or more than 2^24 triangles)");
    c->wide = wide != 0;
    c->bvh_lds = c->bvh_lds_mode[c->wide ? 1 : 0];
    c->resident_blocks.clear();
    return PTGPU_OK;
}
int ptgpu_get_traversal(ptgpu_ctx* c) { return (c && c->wide) ? 1 : 0; }
int ptgpu_kernel_times(ptgpu_ctx* c, double* ms_out, uint64_t* launches_out, uint32_t n) {
    if (!c) return fail(PTGPU_E_INVALID, "kernel_times: NULL ctx");
    for (uint32_t i = 0; i < n && i < PTGPU_N_STAGES; i++) {
        if (ms_out) ms_out[i] = c->stage_ms[i];
        if (launches_out) launches_out[i] = c->stage_launches[i];
    }
    return PTGPU_OK;
}
int ptgpu_set_census(ptgpu_ctx* c, int enabled) {
    if (!c) return fail(PTGPU_E_INVALID, "set_census: NULL ctx");
    HIP_TRY(hipSetDevice(c->device));
    c->census_on = enabled != 0;
    if (enabled) HIP_TRY(hipMemset(c->d_census.p, 0, 4 * sizeof(unsigned long long)));
    return PTGPU_OK;
}
int ptgpu_census(ptgpu_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(PTGPU_E_INVALID, "census: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long h[4];
    HIP_TRY(hipMemcpy(h, c->d_census.p, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out[i] = h[i];
    return PTGPU_OK;
}
int ptgpu_film_device(ptgpu_ctx* c, double** device_ptr_out, size_t* count_out) {
    if (!c || !c->film_ready) return fail(PTGPU_E_STATE, "film_device: no completed render");
    if (device_ptr_out) *device_ptr_out = c->d_film_packed.p;
    if (count_out) *count_out = 3 * (size_t)c->camera.resx * c->camera.resy;
    return PTGPU_OK;
}
int ptgpu_guard_counts(ptgpu_ctx* c, uint64_t out[4]) {
    if (!c || !out) return fail(PTGPU_E_INVALID, "guard_counts: NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    uint32_t h[kNumGuards];
    HIP_TRY(hipMemcpy(h, c->d_guards.p, sizeof(h), hipMemcpyDeviceToHost));
    for (int i = 0; i < kNumGuards; i++) out[i] = h[i];
    return PTGPU_OK;
}
int ptgpu_set_batch_paths(ptgpu_ctx* c, uint32_t n_paths) {
    if (!c) return fail(PTGPU_E_INVALID, "set_batch_paths: NULL ctx");
    c->batch_paths = n_paths;
    return PTGPU_OK;
}

}  // extern "C"

namespace {

// The four-wide walks take their queue as a stream (lanes refilled as their rays finish: veach-mis extend 31.2 -> 29.0 ms
// per 128 spp, guided-large 327 -> 342 Msamples/s with the occlusion streamed as well); the binary walk of small scenes
// does not (most rays finish inside one round of the while-while loop: nothing to refill, Cornell extend +13 %).
// PTGPU_STREAM_WALK=0: one ray per lane and trip everywhere.
bool stream_walk_enabled() {
    static const char* env = getenv("PTGPU_STREAM_WALK");
    static const bool on = env ? atoi(env) != 0 : true;
    return on;
}
void launch_extend(ptgpu_ctx* c, int grid, const PathCols& A, const HitCols& H, const RoundCounters* cnt, const uint32_t* top = nullptr) {
    if (stream_walk_enabled() && c->wide) {
        if (c->census_on) PT_LAUNCH_WALK(c, k_extend_stream<true>, grid, c->scene, A, H, cnt, c->d_census.p, top, c->capacity);
        else PT_LAUNCH_WALK(c, k_extend_stream<false>, grid, c->scene, A, H, cnt, c->d_census.p, top, c->capacity);
        return;
    }
    if (c->census_on) PT_LAUNCH_WALK(c, PT_WALKER(c, k_extend, true), grid, c->scene, A, H, cnt, c->d_census.p, top, c->capacity);
    else PT_LAUNCH_WALK(c, PT_WALKER(c, k_extend, false), grid, c->scene, A, H, cnt, c->d_census.p, top, c->capacity);
}

int upload_lambda_table(ptgpu_ctx* c, const std::vector<float>& u) {
    // grow-only buffers kept by the context: a render allocates nothing here once the first one of its size has run
    HIP_TRY(c->d_lambda_u.reserve(u.size()));
    HIP_TRY(hipMemcpyAsync(c->d_lambda_u.p, u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // `u` is the caller's temporary
    HIP_TRY(c->d_lambda_tab.reserve(4 * u.size()));
    const uint32_t n = (uint32_t)u.size();
    hipLaunchKernelGGL(k_build_lambda_table, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->d_lambda_u.p, c->d_lambda_tab.p, n);
    HIP_TRY(hipGetLastError());
    // every spectrum of the scene at every row's wavelengths (SpecTab, pt_types.h)
    const uint32_t n_spectra = 2 * c->n_materials + c->n_emissions;
    const uint64_t cells = (uint64_t)n_spectra * n;
    if (cells > 0x7fffffffull) return fail(PTGPU_E_LIMIT, "render: materials x wavelength samples too large for the spectrum table");
    HIP_TRY(c->d_spec_tab.reserve(2 * cells));
    hipLaunchKernelGGL(k_build_spectrum_table, dim3((uint32_t)((cells + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->scene, c->d_lambda_tab.p, n, c->n_materials,
                       n_spectra, c->d_spec_tab.p);
    HIP_TRY(hipGetLastError());
    c->spec = SpecTab{c->d_spec_tab.p, n, 2 * c->n_materials};
    return PTGPU_OK;
}

uint32_t pick_capacity(const ptgpu_ctx* c, uint64_t total_paths) {
    // 64 Mi paths in flight by default (~24 GB of state, hit and shadow-ray columns): with 4 Mi a launch of the walkers is
    // seven trips of the resident grid and its first and last trips run on a half-empty GPU (veach-mis, 256 spp: 891
    // Msamples/s at 4 Mi, 986 at 8 Mi, 1 044 at 16 Mi, 1 082 at 32 Mi, 1 103 at 64 Mi; lightpath 507 -> 565).
    uint64_t cap = c->batch_paths ? c->batch_paths : (1ull << 26);
    const uint64_t npix = (uint64_t)c->camera.resx * c->camera.resy;
    if (cap < npix) cap = npix;                     // a batch always covers whole sample passes
    cap = (cap / npix) * npix;
    if (cap > total_paths) cap = ((total_paths + npix - 1) / npix) * npix;
    return (uint32_t)std::min<uint64_t>(cap, 0x7fffffffull);
}

// One batch of camera or light paths through all bounces. Returns the number of rounds run.
// `deferred_at` >= 0: the scene has no pass-through surface, so the walk ends after exactly max_depth rounds and the host
// need not look at the queue: the batch uses the counters from index deferred_at on (cleared by the caller), nothing is
// read back here and the GPU goes from this batch straight into the next (flush_deferred_counters reads them all).
int run_batch_path_like(ptgpu_ctx* c, bool light, const RenderParams& rp, uint32_t n_paths, uint64_t first_sample, uint64_t sample_step,
                        ptgpu_stats* stats, int deferred_at = -1) {
    const uint32_t npix = c->camera.resx * c->camera.resy;
    if (deferred_at < 0) HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, c->d_counters.n * sizeof(RoundCounters), c->stream));
    PathCols A = c->set_a.view(c->capacity), B = c->set_b.view(c->capacity);
    HitCols H{c->d_hit_tb.p, c->d_hit_prim.p};
    ShadowCols S = shadow_view(c);
    RoundCounters* cnt = c->d_counters.p + (deferred_at < 0 ? 0 : deferred_at);
    const int grid = grid_for(c, n_paths);
    if (light) {
        {
            StageTimer t(c, PTGPU_STAGE_RAYGEN);
            hipLaunchKernelGGL(k_raygen_light, dim3(grid), dim3(kBlock), 0, c->stream, c->scene, c->camera, rp, A, S, cnt, n_paths, npix, first_sample, sample_step);
        }
        StageTimer t(c, PTGPU_STAGE_OCCLUDE);
        if (c->census_on) PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_splat, true), grid, c->scene, rp, S, cnt, c->d_census.p);
        else PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_splat, false), grid, c->scene, rp, S, cnt, c->d_census.p);
    } else {
        StageTimer t(c, PTGPU_STAGE_RAYGEN);
        hipLaunchKernelGGL(k_raygen_camera, dim3(grid), dim3(kBlock), 0, c->stream, c->scene, c->camera, rp, A, cnt, n_paths, npix, first_sample, sample_step);
    }
    HIP_TRY(hipGetLastError());
    int round = 0;
    // Rounds needed: max_depth for paths that never meet a pass-through surface; more only if they do.
    const int sure_rounds = (int)std::min<uint32_t>(rp.max_depth, kMaxRounds);
    for (;;) {
        if (round >= kMaxRounds) return fail(PTGPU_E_LIMIT, "render: more than 65536 wavefront rounds in one batch");
        {
            StageTimer t(c, PTGPU_STAGE_EXTEND);
            launch_extend(c, grid, A, H, cnt + round);
        }
        {
            StageTimer t(c, PTGPU_STAGE_SHADE);
            if (light)
                PT_LAUNCH_SHADE(c, k_shade_light, grid, c->scene, c->camera, rp, A, H, B, S, cnt + round, cnt + round + 1);
            else
                PT_LAUNCH_SHADE(c, k_shade_path, grid, c->scene, rp, A, H, B, S, cnt + round, cnt + round + 1);
        }
        {
            StageTimer t(c, PTGPU_STAGE_OCCLUDE);
            if (light && stream_walk_enabled() && c->wide) {
                if (c->census_on) PT_LAUNCH_WALK(c, k_occlude_splat_stream<true>, grid, c->scene, rp, S, cnt + round + 1, c->d_census.p);
                else PT_LAUNCH_WALK(c, k_occlude_splat_stream<false>, grid, c->scene, rp, S, cnt + round + 1, c->d_census.p);
            } else if (light) {
                if (c->census_on) PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_splat, true), grid, c->scene, rp, S, cnt + round + 1, c->d_census.p);
                else PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_splat, false), grid, c->scene, rp, S, cnt + round + 1, c->d_census.p);
            } else if (stream_walk_enabled() && c->wide) {
                if (c->census_on) PT_LAUNCH_WALK(c, k_occlude_path_stream<true>, grid, c->scene, rp, B, S, cnt + round + 1, c->d_census.p);
                else PT_LAUNCH_WALK(c, k_occlude_path_stream<false>, grid, c->scene, rp, B, S, cnt + round + 1, c->d_census.p);
            } else {
                if (c->census_on) PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_path, true), grid, c->scene, rp, B, S, cnt + round + 1, c->d_census.p);
                else PT_LAUNCH_WALK(c, PT_WALKER(c, k_occlude_path, false), grid, c->scene, rp, B, S, cnt + round + 1, c->d_census.p);
            }
        }
        HIP_TRY(hipGetLastError());
        std::swap(A, B);
        round++;
        if (deferred_at >= 0) {
            if (round >= sure_rounds) return PTGPU_OK;
            continue;
        }
        if (round >= sure_rounds || (round & 15) == 0) {  // the host looks at the queue length rarely: every look drains the stream
            // all counters so far in one copy: the look that finds the queue empty is also the batch's statistics
            HIP_TRY(hipMemcpyAsync(c->h_counters.data(), cnt, (size_t)(round + 1) * sizeof(RoundCounters), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (c->h_counters[round].n_active == 0) break;
        }
    }
    if (stats) {
        for (int r = 0; r <= round; r++) {
            stats->extend_rays += c->h_counters[r].n_active;
            stats->shadow_rays += c->h_counters[r].n_shadow;
            stats->segments += c->h_counters[r].n_segments;
        }
        stats->samples += n_paths;
    }
    return PTGPU_OK;
}

// Counters of `n_batches` deferred batches of `per_batch` entries each: one copy, one wait, the statistics of all of them.
int flush_deferred_counters(ptgpu_ctx* c, uint32_t n_batches, uint32_t per_batch, const std::vector<uint32_t>& batch_paths, ptgpu_stats* stats) {
    if (n_batches == 0) return PTGPU_OK;
    HIP_TRY(hipMemcpyAsync(c->h_counters.data(), c->d_counters.p, (size_t)n_batches * per_batch * sizeof(RoundCounters), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (uint32_t b = 0; b < n_batches; b++) {
        const RoundCounters* h = c->h_counters.data() + (size_t)b * per_batch;
        if (h[per_batch - 1].n_active != 0) return fail(PTGPU_E_STATE, "render: paths left after max_depth rounds in a scene without pass-through surfaces");
        if (!stats) continue;
        for (uint32_t r = 0; r < per_batch; r++) {
            stats->extend_rays += h[r].n_active;
            stats->shadow_rays += h[r].n_shadow;
            stats->segments += h[r].n_segments;
        }
        stats->samples += batch_paths[b];
    }
    return PTGPU_OK;
}

int render_path_like(ptgpu_ctx* c, const ptgpu_integrator_desc* integ, uint32_t spp, uint32_t shard_index, uint32_t shard_count, ptgpu_stats* stats) {
    const bool light = integ->kind == PTGPU_INTEGRATOR_LIGHTPATH;
    const uint64_t npix = (uint64_t)c->camera.resx * c->camera.resy;
    // this shard's samples: i = shard_index, shard_index + shard_count, ...
    const uint64_t my_samples = spp > shard_index ? ((uint64_t)spp - shard_index + shard_count - 1) / shard_count : 0;
    HIP_TRY(hipMemsetAsync(c->d_film.p, 0, c->d_film.n * sizeof(double), c->stream));
    if (my_samples == 0 || integ->max_depth == 0) return PTGPU_OK;
    // (the device film exists by now; what comes after the workspace is the packed film, 24 bytes per pixel, and the wavelength table)
    int rc = ensure_workspace_or_smaller(c, pick_capacity(c, my_samples * npix), npix, (size_t)npix * 24u + ((size_t)256 << 20));
    if (rc != PTGPU_OK) return rc;
    rc = upload_lambda_table(c, golden_iterates(spp));
    if (rc != PTGPU_OK) return rc;
    RenderParams rp;
    rp.max_depth = integ->max_depth > 0xffffu ? 0xffffu : integ->max_depth;
    rp.mode = integ->mode;
    rp.capacity = c->capacity;
    rp.resx = c->camera.resx;
    rp.lambda_tab = c->d_lambda_tab.p;
    rp.spec = c->spec;
    rp.film = c->d_film.p;
    rp.film_copies = c->film_copies;
    rp.n_pixels = (uint32_t)npix;
    const uint64_t passes_per_batch = c->capacity / npix;
    // Without pass-through surfaces every batch takes exactly max_depth rounds: batches follow one another on the stream
    // without a look from the host in between (each look drained the GPU: ~0.2 ms per batch, 4 % of veach-mis), each with
    // counters of its own; the host reads a group of them in one copy.
    static const bool sync_batches = getenv("PTGPU_SYNC_BATCHES") != nullptr;  // A/B: the host looks after every batch
    const bool deferred = !c->has_passthrough && rp.max_depth + 1u <= 1024u && !sync_batches;
    const uint32_t per_batch = rp.max_depth + 1u, group = deferred ? (uint32_t)((c->d_counters.n) / per_batch) : 0u;
    std::vector<uint32_t> batch_paths;
    for (uint64_t done = 0; done < my_samples; done += passes_per_batch) {
        const uint64_t passes = std::min(passes_per_batch, my_samples - done);
        const uint64_t first = shard_index + done * shard_count;
        if (deferred) {
            if (batch_paths.empty()) HIP_TRY(hipMemsetAsync(c->d_counters.p, 0, c->d_counters.n * sizeof(RoundCounters), c->stream));
            rc = run_batch_path_like(c, light, rp, (uint32_t)(passes * npix), first, shard_count, stats, (int)(batch_paths.size() * per_batch));
            if (rc != PTGPU_OK) return rc;
            batch_paths.push_back((uint32_t)(passes * npix));
            if (batch_paths.size() == group || done + passes_per_batch >= my_samples) {
                rc = flush_deferred_counters(c, (uint32_t)batch_paths.size(), per_batch, batch_paths, stats);
                if (rc != PTGPU_OK) return rc;
                batch_paths.clear();
            }
        } else {
            rc = run_batch_path_like(c, light, rp, (uint32_t)(passes * npix), first, shard_count, stats);
            if (rc != PTGPU_OK) return rc;
        }
    }
    return PTGPU_OK;
}

int check_render_args(ptgpu_ctx* c, const ptgpu_integrator_desc* integ, uint32_t shard_index, uint32_t shard_count) {
    if (!c || !integ) return fail(PTGPU_E_INVALID, "render: NULL argument");
    if (!c->have_scene || !c->have_camera) return fail(PTGPU_E_STATE, "render: scene_create and camera_set must come first");
    if (shard_count == 0 || shard_index >= shard_count) return fail(PTGPU_E_INVALID, "render: bad shard");
    if (integ->kind < 0 || integ->kind > 2) return fail(PTGPU_E_INVALID, "render: unknown integrator");
    if (integ->kind != PTGPU_INTEGRATOR_LIGHTPATH && (integ->mode < 0 || integ->mode > 2)) return fail(PTGPU_E_INVALID, "render: unknown mode");
    return PTGPU_OK;
}

// Packs the device film into RGB sums (d_film_packed: what ptgpu_film_device exposes) and, when the caller wants a host
// copy, downloads it.
int download_film(ptgpu_ctx* c, double* film_out) {
    const uint32_t npix = c->camera.resx * c->camera.resy;
    HIP_TRY(c->d_film_packed.reserve(3 * (size_t)npix));
    hipLaunchKernelGGL(k_pack_film, dim3((3 * npix + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->d_film.p, c->d_film_packed.p, npix, c->film_copies);
    HIP_TRY(hipGetLastError());
    if (film_out) HIP_TRY(hipMemcpyAsync(film_out, c->d_film_packed.p, 3 * (size_t)npix * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    uint32_t nonfinite_now = 0;
    HIP_TRY(hipMemcpyAsync(&nonfinite_now, c->d_guards.p + kGuardNonFinite, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->nonfinite_last = nonfinite_now - c->nonfinite_seen;  // (u32 wrap-around safe)
    c->nonfinite_seen = nonfinite_now;
    c->film_ready = true;
    return PTGPU_OK;
}

}  // namespace

#include "ptgpu_guided_api.inc"

extern "C" {

int ptgpu_render_shard(ptgpu_ctx* c, const ptgpu_integrator_desc* integ, uint32_t spp, uint32_t shard_index, uint32_t shard_count,
                       double* film_rgb_out, double* film_scale_out, ptgpu_stats* stats) {
    int rc = check_render_args(c, integ, shard_index, shard_count);
    if (rc != PTGPU_OK) return rc;
    if (spp == 0) return fail(PTGPU_E_INVALID, "render: spp must be > 0");
    HIP_TRY(hipSetDevice(c->device));
    if (stats) memset(stats, 0, sizeof(*stats));
    reset_stage_times(c);
    c->film_ready = false;
    const auto t0 = std::chrono::steady_clock::now();
    c->film_copies = integ->kind == PTGPU_INTEGRATOR_LIGHTPATH ? kLightFilms : 1;
    const size_t film_n = (size_t)c->camera.resx * c->camera.resy * kFilmStride * c->film_copies;
    if (c->d_film.n != film_n) HIP_TRY(c->d_film.alloc(film_n));
    if (integ->kind == PTGPU_INTEGRATOR_GUIDED) {
        uint32_t n_waves = 0;
        rc = ptgpu_guided_begin(c, integ, spp, shard_index, shard_count, &n_waves);
        for (uint32_t w = 0; rc == PTGPU_OK && w < n_waves; w++) rc = ptgpu_guided_wave(c, w);
        if (rc != PTGPU_OK) return rc;
        return ptgpu_guided_end(c, film_rgb_out, film_scale_out, stats);
    }
    rc = render_path_like(c, integ, spp, shard_index, shard_count, stats);
    if (rc != PTGPU_OK) return rc;
    rc = download_film(c, film_rgb_out);
    if (rc != PTGPU_OK) return rc;
    resolve_stage_times(c);
    if (film_scale_out) *film_scale_out = 1.0 / (double)spp;  // Integrator::film_scale (integrator/mod.rs:47-50)
    if (stats) {
        stats->render_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->extend_ms = c->stage_ms[PTGPU_STAGE_EXTEND] + c->stage_ms[PTGPU_STAGE_OCCLUDE];
        stats->shade_ms = c->stage_ms[PTGPU_STAGE_SHADE];
        stats->kernel_launches = c->launches;
        stats->nonfinite_samples = c->nonfinite_last;
    }
    // PathIntegrator::render asserts that every sample is finite (path_tracer.rs:46) and LightPathIntegrator does not: a
    // `path` render that met one says so (the render is complete: film and stats are delivered, without those samples).
    if (c->nonfinite_last != 0 && integ->kind == PTGPU_INTEGRATOR_PATH)
        return fail(PTGPU_W_NONFINITE, "render: " + std::to_string(c->nonfinite_last) + " sample(s) with non-finite radiance (the reference asserts, path_tracer.rs:46); they are left out of the film");
    return PTGPU_OK;
}

int ptgpu_render(ptgpu_ctx* c, const ptgpu_integrator_desc* integ, uint32_t spp, double* film_rgb_out, double* film_scale_out, ptgpu_stats* stats) {
    return ptgpu_render_shard(c, integ, spp, 0, 1, film_rgb_out, film_scale_out, stats);
}

}  // extern "C"
