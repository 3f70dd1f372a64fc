/* ptgpu.h — C ABI of the MI355X wavefront path tracer (libptgpu.so).
 *
 * Drop-in boundary for the hot path of eDev03/Path-tracer's `pt` crate: everything below
 * `Integrator::render(&mut self, scene, spp, camera, film)` (pt/src/integrator/mod.rs:34-46), i.e. the
 * per-sample loops of the `path`, `lightpath` and `guided` integrators and all they call.
 * The reference has no FFI; these entry points are what an `extern "C"` block in a thin Rust FFI crate
 * would bind (see INTEGRATION.md for the Rust side). Plain pointers and sizes only; the caller owns every
 * host buffer; the library copies what it needs during the call. All functions return 0 on success or a
 * negative PTGPU_E_* code (or the warning PTGPU_W_NONFINITE, 1) and never throw/abort across the ABI; ptgpu_last_error() gives the message.
 * One context = one GPU; a context is not re-entrant (one call at a time), calls block until done.
 */
#ifndef PTGPU_H
#define PTGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTGPU_OK 0
#define PTGPU_E_INVALID (-1)  /* bad argument / inconsistent description */
#define PTGPU_E_DEVICE (-2)   /* HIP runtime error (no GPU, out of memory, launch failure) */
#define PTGPU_E_STATE (-3)    /* call order (e.g. render before scene_create / camera_set) */
#define PTGPU_E_LIMIT (-4)    /* a documented limit was exceeded (e.g. DTree > 65535 nodes) */
#define PTGPU_W_NONFINITE 1    /* the one positive code: the render is COMPLETE (film, scale and stats delivered), but `path` met
                                * samples whose radiance was not finite and left them out — where the reference asserts
                                * (path_tracer.rs:46; the MIS weight of a grazing light sample is inf / inf about once in 10^6
                                * samples of the furnace fixtures). ptgpu_stats::nonfinite_samples says how many. */

typedef struct ptgpu_ctx ptgpu_ctx;

/* Sigmoid-polynomial spectrum = the reference's TristimulusSpectrum after construction
 * (pt/src/spectrum/mod.rs:262-335): s(l) = scale * (0.5 + x / (2 sqrt(1 + x^2))), x = fma(l, fma(l,c0,c1), c2).
 * The host keeps `Spectrum::new` / rgb2spec::fetch_coeffs (scene-build time) and passes the result. */
typedef struct ptgpu_spectrum {
    float c0, c1, c2, scale;
} ptgpu_spectrum;

/* io_bridge::Material (pt/src/io_bridge.rs) / scene::Material (pt/src/scene.rs:21-33). */
enum { PTGPU_MAT_LAMBERTIAN = 0, PTGPU_MAT_DIELECTRIC = 1, PTGPU_MAT_CONDUCTOR = 2 };
typedef struct ptgpu_material {
    int32_t kind;      /* PTGPU_MAT_* */
    float roughness;   /* conductor alpha */
    ptgpu_spectrum s0; /* albedo | eta */
    ptgpu_spectrum s1; /* conductor k */
} ptgpu_material;

/* io_bridge::Mesh (pt/src/io_bridge.rs), world-space, as handed to Scene::new (pt/src/scene.rs:89-93). */
typedef struct ptgpu_mesh {
    const float* points;       /* n_points * 3 */
    const float* normals;      /* n_points * 3, or NULL = flat shaded (empty `normals`) */
    const uint32_t* triangles; /* n_triangles * 3 vertex indices */
    uint32_t n_points;
    uint32_t n_triangles;
    int32_t material_index;  /* index into materials, -1 = None (pass-through surface) */
    int32_t emissive;        /* != 0 <=> io_bridge emission != RGB 0 */
    ptgpu_spectrum emission; /* illuminant spectrum (multiplied by D65 on evaluation) */
} ptgpu_mesh;

typedef struct ptgpu_scene_desc {
    const ptgpu_mesh* meshes;
    const ptgpu_material* materials;
    uint32_t n_meshes;
    uint32_t n_materials;
} ptgpu_scene_desc;

/* io_bridge::CameraOptions (pt/src/io_bridge.rs) -> Camera::new (pt/src/camera.rs:33-41). */
typedef struct ptgpu_camera_desc {
    float world_to_camera[16]; /* row-major 4x4 */
    float fov_deg;             /* vertical field of view */
    uint32_t resx, resy;
} ptgpu_camera_desc;

/* io_bridge::Integrator (pt/src/io_bridge.rs) -> Integrator::new (pt/src/integrator/mod.rs:21-33). */
enum { PTGPU_INTEGRATOR_LIGHTPATH = 0, PTGPU_INTEGRATOR_PATH = 1, PTGPU_INTEGRATOR_GUIDED = 2 };
enum { PTGPU_MODE_PT = 0, PTGPU_MODE_NEE = 1, PTGPU_MODE_MIS = 2 };
enum { PTGPU_FILTER_NEAREST = 0, PTGPU_FILTER_BOX = 1, PTGPU_FILTER_STOCHASTIC = 2 };
typedef struct ptgpu_integrator_desc {
    int32_t kind;                /* PTGPU_INTEGRATOR_* */
    uint32_t max_depth;
    int32_t mode;                /* PTGPU_MODE_* (path, guided) */
    float spatial_threshold;     /* guided: `spatial_threashold` */
    float directional_threshold; /* guided: carried for fidelity; render uses 0.01 like path_guiding/mod.rs:73 */
    int32_t spatial_filter;      /* guided: STOCHASTIC, the only value the reference's parser selects (parser/mod.rs:144;
                                  * stree.rs:186-201), NEAREST (stree.rs:117-126) or BOX (stree.rs:127-185: a record goes, in
                                  * shares of its footprint's volume, to every leaf the footprint overlaps; not with the
                                  * directional STOCHASTIC filter, whose draws would be per leaf: PTGPU_E_INVALID) */
    int32_t directional_filter;  /* guided: BOX (parser/mod.rs:145; dtree.rs:235-289), NEAREST (dtree.rs:208-222) or
                                  * STOCHASTIC (dtree.rs:223-234: two more draws per record).
                                  * Deviation: a directional quadtree is at most 24 levels deep here (cells of 2^-24; the
                                  * reference has no cap, its u16 node links bound the node count only) */
} ptgpu_integrator_desc;

/* Counters filled by ptgpu_render*. Times are host wall-clock around the device work of the call. */
typedef struct ptgpu_stats {
    uint64_t samples;        /* camera / light paths started */
    uint64_t extend_rays;    /* closest-hit rays traced */
    uint64_t shadow_rays;    /* occlusion rays traced */
    uint64_t segments;       /* shaded surface interactions */
    uint64_t kernel_launches;
    double render_ms;        /* whole call */
    double extend_ms;        /* sum of HIP-event times of extend launches (0 unless profiling enabled) */
    double shade_ms;
    uint64_t stree_nodes;    /* guided: final spatial tree size */
    uint64_t dtree_nodes;    /* guided: total directional nodes over leaves */
    uint64_t nonfinite_samples; /* samples left out of the film because their radiance was not finite: a `path` render
                                 * returns PTGPU_W_NONFINITE when this is not 0 (the reference asserts); lightpath / guided,
                                 * whose reference loops do not check, return PTGPU_OK and report it here */
} ptgpu_stats;

typedef struct ptgpu_hit {
    uint32_t mesh_index;     /* CompactTriangle (pt/src/bvh.rs:18-22) after Scene::new's emissive-first reordering */
    uint32_t triangle_index;
    uint32_t leaf_index;     /* position in Bvh::compact_triangles (leaf order) */
    uint32_t hit;            /* 1 = hit, 0 = miss (other fields zero) */
    float t, b0, b1, b2;     /* RayTriangleIntersection (pt/src/geom.rs:68-73) */
} ptgpu_hit;

const char* ptgpu_last_error(void);

/* Create a context on HIP device `device_id`. */
int ptgpu_init(ptgpu_ctx** out, int device_id);
void ptgpu_destroy(ptgpu_ctx* ctx);

/* Replaces Scene::new (pt/src/scene.rs:89-183) + Bvh::new (pt/src/bvh.rs:48-85): emissive-first mesh
 * ordering, 12-bucket SAH build on the host (bit-identical leaf order), flattening to HBM. */
int ptgpu_scene_create(ptgpu_ctx* ctx, const ptgpu_scene_desc* desc);
/* Replaces Camera::new + Film::new (pt/src/camera.rs:33-41,150-160). */
int ptgpu_camera_set(ptgpu_ctx* ctx, const ptgpu_camera_desc* cam);

/* Replaces Integrator::render + Film::create_pixels(1.0) + Integrator::film_scale
 * (pt/src/integrator/mod.rs:34-55, pt/src/camera.rs:184-196).
 * film_rgb_out: resx*resy*3 doubles, row-major (x + y*resx), UNSCALED sums of tile pixels + splats;
 * film_scale_out: the factor save_image would apply. stats may be NULL. */
int ptgpu_render(ptgpu_ctx* ctx, const ptgpu_integrator_desc* integrator, uint32_t spp, double* film_rgb_out,
                 double* film_scale_out, ptgpu_stats* stats);

/* Same, for one shard of a multi-GPU job: this context renders only the sample indices i with
 * i % shard_count == shard_index (path, lightpath), or those sample passes of every guided wave. The caller
 * sums the shards' films (and, for guided, all-reduces the statistics buffer between waves, below). */
int ptgpu_render_shard(ptgpu_ctx* ctx, const ptgpu_integrator_desc* integrator, uint32_t spp, uint32_t shard_index,
                       uint32_t shard_count, double* film_rgb_out, double* film_scale_out, ptgpu_stats* stats);

/* Guided integrator, wave by wave (PathGuidingIntegrator::render, pt/src/integrator/path_guiding/mod.rs:70-122)
 * so that a multi-GPU driver can all-reduce the SD-tree statistics between waves:
 *   begin -> for w in 0..n_waves { wave(w); [all-reduce stats in place]; } -> end.
 * ptgpu_guided_stats returns the DEVICE pointer and element count (f32) of the flat collector buffer of the
 * wave just traced: the DTree node sums of every spatial leaf (4 per node, leaves in order), then one statistical
 * weight per leaf. Topology is identical on every shard, so an element-wise sum across GPUs is exactly the
 * single-GPU state up to f32 reassociation. The buffer is complete when ptgpu_guided_wave returns; the library's
 * streams are non-blocking (they wait for no other stream, the null stream included), so whoever writes the buffer
 * completes the write — synchronises the stream the collective ran on — before the next ptgpu call. */
int ptgpu_guided_begin(ptgpu_ctx* ctx, const ptgpu_integrator_desc* integrator, uint32_t spp, uint32_t shard_index,
                       uint32_t shard_count, uint32_t* n_waves_out);
int ptgpu_guided_wave(ptgpu_ctx* ctx, uint32_t wave);
int ptgpu_guided_stats(ptgpu_ctx* ctx, float** device_ptr_out, size_t* count_out);
int ptgpu_guided_end(ptgpu_ctx* ctx, double* film_rgb_out, double* film_scale_out, ptgpu_stats* stats);
/* Multi-GPU film reduce without a host round trip (no reference counterpart): after ptgpu_render[_shard] or
 * ptgpu_guided_end returned (film_rgb_out may be NULL there to skip the download), the DEVICE pointer and element count
 * (resx * resy * 3 doubles, same layout as film_rgb_out) of this context's film sums, e.g. for an in-place ncclReduce.
 * Valid until the next render or camera_set on this context. */
int ptgpu_film_device(ptgpu_ctx* ctx, double** device_ptr_out, size_t* count_out);
/* Parity hook for the guided sampler (no reference counterpart as an API). Between two waves: draws n_probes random
 * points in the SD-tree of the wave just refined and, for each, (a) runs DTree::sample then DTree::pdf as two separate
 * descents (pt/src/integrator/path_guiding/dtree.rs:98-179) and (b) the single fused descent the shade kernel uses;
 * also DTree::pdf of a random direction both ways. counts_out = {probes that sampled a non-uniform tree, probes whose
 * sampled direction / PCG state differ between (a) and (b), probes whose pdf bits differ, pdf-only probes that differ,
 * probes whose spatial leaf, leaf box or sampler word differ between the jump-table descent and plain node chasing
 * (STree::leaf, stree.rs:202-229)}: the last four must be 0. */
int ptgpu_guided_selfcheck(ptgpu_ctx* ctx, uint32_t n_probes, uint64_t seed, uint64_t counts_out[5]);

/* SD-tree parity hooks (no reference counterpart as an API), usable between two waves (after ptgpu_guided_wave returned).
 * They expose the resident tree and run the production kernels' lookups, record pass and refinement on caller-supplied
 * inputs, so that tests can hold the SAME tree on the GPU and in the CPU oracle and compare element-wise:
 *   DTree::sample / DTree::pdf (path_guiding/dtree.rs:98-179), STree::leaf (stree.rs:202-229)       -> ptgpu_guided_probe
 *   STree::record, stochastic filter (stree.rs:102-116,186-201) + DTree::record, box (dtree.rs:200-289) -> ptgpu_guided_record_batch
 *   STree::new_iteration (stree.rs:42-101,255-260; dtree.rs:185-198,291-328)                       -> ptgpu_guided_refine
 * "Resident tree" = the sampling tree the last traced wave drew from (iteration = that wave's index) and the collector
 * it trained; after ptgpu_guided_refine, the tree the next wave will use and its empty collector. */
typedef struct ptgpu_sdtree_info {
    uint32_t n_spatial_nodes, n_leaves, n_sampling_nodes, n_collector_nodes;
    uint32_t iteration;      /* Guider::iteration of the resident sampling tree */
    uint32_t refined_ahead;  /* 1 after ptgpu_guided_refine until the next ptgpu_guided_wave */
    float bounds[6];         /* STree::bounds (stree.rs:28-31): min.xyz, max.xyz */
    uint32_t refine_repeats; /* refinements of this render that were built twice: the new generation outgrew a buffer sized
                              * before its counts were known (the spatial sweep runs on the device) */
} ptgpu_sdtree_info;
int ptgpu_guided_tree_info(ptgpu_ctx* ctx, ptgpu_sdtree_info* out);
/* Any pointer may be NULL. spatial: 4 words per STree node {left, right, split_axis, leaf id | 0xffffffff for interior};
 * leaf: 5 words per spatial leaf {first node in s_*, node count, first node in c_*, node count, sampler mean == 0};
 * s_rad / c_rad: 4 floats per quadtree node (child sums), s_child / c_child: 4 u16 links (0 = leaf cell), node indices
 * relative to the leaf's first node; s_weight / c_weight: statistical_weight of each leaf's sampler / collector DTree. */
int ptgpu_guided_tree_download(ptgpu_ctx* ctx, uint32_t* spatial, uint32_t* leaf, float* s_rad, uint16_t* s_child, float* s_weight,
                               float* c_rad, uint16_t* c_child, float* c_weight);
/* Probe i: point = points[3i..], PCG stream = Sampler::init_pixel(i & 0xffff, i >> 16, seed); out_words[16i..] =
 * {leaf id, sampled square x, y, tree pdf of it, tree pdf of query_sq[2i..], PCG state lo, hi after the draws,
 *  1 if the shade kernel's split descent disagrees with the plain one, leaf box min.xyz max.xyz,
 *  dir_to_square (mod.rs:412-421) of the point read as a direction: x, y} (floats as bits). */
int ptgpu_guided_probe(ptgpu_ctx* ctx, const float* points, const float* query_sq, uint32_t n, uint64_t seed, uint32_t* out_words);
/* Records n entries {p.xyz, wi.xyz, radiance, u0, u1, u2} (u*: the stochastic filter's three jitter draws) into the
 * resident collector through the production record pass (k_guide_locate/scatter/splat_bins, k_push_lazy, k_fold_weights). */
int ptgpu_guided_record_batch(ptgpu_ctx* ctx, const float* records, uint32_t n);
/* Builds the tree of the next wave now (what the next ptgpu_guided_wave would do first); that call then skips it. */
int ptgpu_guided_refine(ptgpu_ctx* ctx);
/* Parity hook for the back-to-front loop of PathGuidingIntegrator::render (path_guiding/mod.rs:106-115): with capture on,
 * every ptgpu_guided_wave keeps what its production kernels (k_guide_scan, then k_guide_locate) made of each
 * `guider.add(l.average() / v.wi_pdf, v.p, v.wi, sampler)` call — 14 words per record, floats as bits:
 *   {p.xyz, wi.xyz, radiance, u0, u1, u2 (the stochastic filter's draws, stree.rs:195-199), spatial leaf id the record
 *    went to (0xffffffff with PTGPU_DIRECT_SPLAT), pixel (x + y * resx), sample pass (0xffffffff unless the image is a
 *    whole number of 16 x 16 tiles and the render is unsharded), index of the vertex in its path}.
 * ptgpu_guided_captured_records returns the records of the last wave traced (all of them counted in *n_records_out,
 * at most capacity_records copied; out_words may be NULL to ask for the count). Waves traced with capture on wait for
 * the device and copy the queue through pageable memory: a test path. */
int ptgpu_guided_capture_records(ptgpu_ctx* ctx, int enabled);
int ptgpu_guided_captured_records(ptgpu_ctx* ctx, uint32_t* out_words, uint64_t capacity_records, uint64_t* n_records_out);

/* Parity hooks (no reference counterpart as an API; they expose Scene::intersect_t -> Bvh::intersect,
 * pt/src/bvh.rs:86-142, and Scene::occluded, pt/src/scene.rs:229-235, on caller-supplied rays).
 * rays: n * 7 floats (ox,oy,oz, dx,dy,dz, t_max).
 * ptgpu_trace_batch doubles as the extend micro-benchmark: it resets the stage timers, and with profiling on reports the
 * walk kernel's time as PTGPU_STAGE_EXTEND; with the census on it adds nodes popped / triangles tested to census[0..1]. */
int ptgpu_trace_batch(ptgpu_ctx* ctx, const float* rays, uint32_t n, ptgpu_hit* hits_out);
/* No counterpart in the reference. ptgpu_trace_batch for callers that cannot allocate device memory (tight serving loops,
 * harnesses that forbid hipMalloc, code that must not disturb a captured graph's memory): identical results, but the
 * device staging of the rays and hit records is a static 15 MiB array inside the library (one copy per GPU, held until
 * the process ends) instead of a per-call hipMalloc/hipFree, and no timer or census is taken. It allocates nothing; it
 * is NOT itself capturable: it copies from and to the caller's pageable host buffers and waits for the stream
 * (hipStreamSynchronize), neither of which is legal inside a HIP graph capture. n <= 262144 (PTGPU_NOALLOC_MAX_RAYS), more
 * fails with PTGPU_E_LIMIT. All contexts on one GPU share that array: calls that overlap on one GPU race. */
#define PTGPU_NOALLOC_MAX_RAYS 262144u
int ptgpu_trace_batch_noalloc(ptgpu_ctx* ctx, const float* rays, uint32_t n, ptgpu_hit* hits_out);
/* from/to: n * 3 floats each; out[i] = 1 if Scene::occluded(from_i, to_i). */
int ptgpu_occluded_batch(ptgpu_ctx* ctx, const float* from, const float* to, uint32_t n, uint8_t* out);

/* Scene introspection for parity tests: the flattened BVH exactly as Bvh::new would lay it out.
 * nodes_out: n_nodes * 8 uint32 words = {min.xyz, max.xyz as f32 bits, offset, n_tris | split_axis << 16};
 * leaf_tris_out: n_tris * 2 uint32 = {mesh_index, triangle_index} in leaf order. Either may be NULL. */
int ptgpu_scene_info(ptgpu_ctx* ctx, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* n_emissive_tris, float bounds_out[6]);
/* Where Bvh::new (bvh.rs:48-85,148-304) runs for the scenes created after the call: 0 = on the host, 1 = on the device,
 * 2 = on the device for scenes of 16 384 triangles and more (the default). Both builders produce the same tree word for
 * word (SAH decisions, itertools::partition order, depth-first numbering); the device builder hands inputs with a negative
 * zero or a non-finite coordinate to the host one. ptgpu_bvh_built_on_device: 1 if the resident scene's tree was. */
int ptgpu_set_bvh_builder(ptgpu_ctx* ctx, int mode);
int ptgpu_bvh_built_on_device(ptgpu_ctx* ctx);
int ptgpu_scene_bvh(ptgpu_ctx* ctx, uint32_t* nodes_out, uint32_t* leaf_tris_out);

/* Timing hook for bench.py: when enabled, every kernel launch is bracketed by a pair of HIP events on the library's
 * stream (recorded without waiting, resolved at the end of the call); per-stage sums via ptgpu_kernel_times. */
int ptgpu_set_profiling(ptgpu_ctx* ctx, int enabled);
/* Per-stage totals of the last render call (valid when profiling was enabled): ms_out[i] = summed HIP-event time of
 * the launches of stage i, launches_out[i] = their number; i = PTGPU_STAGE_*; n = entries available in the arrays. */
enum { PTGPU_STAGE_RAYGEN = 0, PTGPU_STAGE_EXTEND = 1, PTGPU_STAGE_SHADE = 2, PTGPU_STAGE_OCCLUDE = 3, PTGPU_STAGE_RECORD = 4,
       PTGPU_STAGE_OTHER = 5, PTGPU_N_STAGES = 6 };
int ptgpu_kernel_times(ptgpu_ctx* ctx, double* ms_out, uint64_t* launches_out, uint32_t n);
/* guided: by default the record pass of one batch (second HIP stream) runs beside the walk of the next; with 0 every
 * kernel has the GPU to itself (same results; about 1 % slower; per-kernel times that are the kernel's own). Initial
 * value: on, unless the environment variable PTGPU_NO_OVERLAP is set. No counterpart in the reference (measurement hook). */
int ptgpu_set_record_overlap(ptgpu_ctx* ctx, int enabled);
/* guided: the reference records every path of the LAST wave into the SD-tree as it does in the others
 * (path_guiding/mod.rs:106-116), although nothing reads that tree again: the image and `stree.len()` (mod.rs:121) do not
 * depend on it. 1 (the default) does the same; 0 leaves the last wave's statistics empty (ptgpu_guided_stats after the
 * last wave then returns zeros) and renders the same image about 10 % sooner. bench.py's headline is measured with 1. */
int ptgpu_set_final_training(ptgpu_ctx* ctx, int enabled);
/* Traversal census for roofline accounting (bench only; selects instrumented kernel variants, off by default):
 * out[0..3] = BVH nodes visited / triangles tested by extend rays, then by occlusion rays, since it was enabled. */
int ptgpu_set_census(ptgpu_ctx* ctx, int enabled);
int ptgpu_census(ptgpu_ctx* ctx, uint64_t out[4]);
/* Deviations that are safety limits, counted since ptgpu_init whenever one actually cut something short (tests assert 0):
 * out[0] = paths ended after 256 consecutive crossings of material-less surfaces (the reference would never return from
 * such a sample, path_tracer.rs:76-80), out[1] = guided path vertices dropped for lack of a vertex-store row (rows =
 * max_depth + 1, one per bounce plus the miss vertex: cannot happen), out[2] = quadtree cells the refinement declined to
 * split because they sit at level 24 (the reference has no depth cap, dtree.rs:291-324), out[3] = samples dropped
 * because their radiance was not finite (the reference asserts, path_tracer.rs:46, or poisons the pixel). */
int ptgpu_guard_counts(ptgpu_ctx* ctx, uint64_t out[4]);
/* Traversal of the scene's BVH (reference: Bvh::intersect, bvh.rs:86-142). 0 = the reference's binary tree,
 * node by node; 1 = a four-wide image of the SAME tree (every wide node holds up to four descendants of one binary
 * node; leaves, triangle order and the per-octant visit order are the binary tree's), which reports the same hits bit
 * for bit at a quarter of the loop trips. Fails with PTGPU_E_LIMIT for a scene whose tree has no wide image (a leaf
 * of more than 128 triangles). Takes effect from the next call on; ptgpu_scene_create resets it to the default:
 * wide for trees of more than 256 KB of binary nodes, where it is faster (PTGPU_BVH_WIDE=0|1 in the environment overrides). ptgpu_get_traversal returns the mode in use. */
int ptgpu_set_traversal(ptgpu_ctx* ctx, int wide);
int ptgpu_get_traversal(ptgpu_ctx* ctx);
/* Paths kept in flight per batch (state SoA size); 0 restores the default. */
int ptgpu_set_batch_paths(ptgpu_ctx* ctx, uint32_t n_paths);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
