// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header). C entry points over the CPU restatement so
// that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can drive it through ctypes. It takes the
// same plain-data descriptions as the product's C ABI (include/ptgpu.h) so both sides see identical inputs.
// Parity status: the reference ships no tests or golden vectors for this path and cannot be
// built here (Rust nightly, missing srgb.coeffs), so this restatement is pinned only by (i) the furnace
// scenes' analytic answer RGB [1,1,1] (README.md:6), (ii) published PCG32 reference vectors, (iii) analytic
// invariants (pdf normalisation, white furnace). Everything else is "parity unpinned".
#include <cstring>
#include <string>

#include "../include/ptgpu.h"
#include "integrators.hpp"

using namespace orc;

struct orc_scene {
    Scene scene;
    TraversalCounters counters;
};

static Spectrum to_spec(const ptgpu_spectrum& s, bool illuminant) {
    Spectrum r;
    r.c0 = s.c0;
    r.c1 = s.c1;
    r.c2 = s.c2;
    r.scale = s.scale;
    r.illuminant = illuminant;
    return r;
}

extern "C" {

orc_scene* orc_scene_create(const ptgpu_scene_desc* d) {
    if (!d || !d->meshes || d->n_meshes == 0) return nullptr;
    std::vector<InMesh> meshes;
    for (uint32_t i = 0; i < d->n_meshes; i++) {
        const ptgpu_mesh& m = d->meshes[i];
        InMesh im;
        for (uint32_t p = 0; p < m.n_points; p++) im.points.push_back(v3(m.points[3 * p], m.points[3 * p + 1], m.points[3 * p + 2]));
        if (m.normals)
            for (uint32_t p = 0; p < m.n_points; p++) im.normals.push_back(v3(m.normals[3 * p], m.normals[3 * p + 1], m.normals[3 * p + 2]));
        im.tris.assign(m.triangles, m.triangles + 3 * (size_t)m.n_triangles);
        im.material_index = m.material_index;
        im.emissive = m.emissive != 0;
        im.emission = to_spec(m.emission, true);
        meshes.push_back(std::move(im));
    }
    std::vector<Material> mats;
    for (uint32_t i = 0; i < d->n_materials; i++) {
        const ptgpu_material& m = d->materials[i];
        mats.push_back(Material{(MatKind)m.kind, m.roughness, to_spec(m.s0, false), to_spec(m.s1, false)});
    }
    orc_scene* s = new orc_scene();
    s->scene.build(std::move(meshes), std::move(mats));
    return s;
}
void orc_scene_destroy(orc_scene* s) { delete s; }

int orc_scene_info(orc_scene* s, uint32_t* n_nodes, uint32_t* n_tris, uint32_t* n_emissive, float bounds[6]) {
    if (!s) return -1;
    if (n_nodes) *n_nodes = (uint32_t)s->scene.bvh.nodes.size();
    if (n_tris) *n_tris = (uint32_t)s->scene.bvh.compact_triangles.size();
    if (n_emissive) *n_emissive = (uint32_t)s->scene.emissive_triangles.size();
    if (bounds) {
        B3 b = s->scene.bounds();
        bounds[0] = b.min.x; bounds[1] = b.min.y; bounds[2] = b.min.z;
        bounds[3] = b.max.x; bounds[4] = b.max.y; bounds[5] = b.max.z;
    }
    return 0;
}
// Same word layout as ptgpu_scene_bvh.
int orc_scene_bvh(orc_scene* s, uint32_t* nodes_out, uint32_t* leaf_tris_out) {
    if (!s) return -1;
    const auto& nodes = s->scene.bvh.nodes;
    if (nodes_out)
        for (size_t i = 0; i < nodes.size(); i++) {
            const BvhNode& n = nodes[i];
            float f[6] = {n.bounds.min.x, n.bounds.min.y, n.bounds.min.z, n.bounds.max.x, n.bounds.max.y, n.bounds.max.z};
            std::memcpy(nodes_out + 8 * i, f, 24);
            nodes_out[8 * i + 6] = n.offset;
            nodes_out[8 * i + 7] = (uint32_t)n.n_tris | ((uint32_t)n.split_axis << 16);
        }
    if (leaf_tris_out)
        for (size_t k = 0; k < s->scene.bvh.compact_triangles.size(); k++) {
            leaf_tris_out[2 * k] = s->scene.bvh.compact_triangles[k].mesh_index;
            leaf_tris_out[2 * k + 1] = s->scene.bvh.compact_triangles[k].triangle_index;
        }
    return 0;
}

// rays: n*7 (o, d, tmax). counters_out[2] = nodes visited, triangles tested (may be NULL).
int orc_trace_batch(orc_scene* s, const float* rays, uint32_t n, ptgpu_hit* out, uint64_t* counters_out) {
    if (!s) return -1;
    TraversalCounters cnt;
    for (uint32_t i = 0; i < n; i++) {
        const float* r = rays + 7 * (size_t)i;
        BvhIsect bi;
        ptgpu_hit h;
        std::memset(&h, 0, sizeof(h));
        if (s->scene.bvh.intersect(Ray{v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5])}, r[6], &bi, &cnt)) {
            h.mesh_index = bi.tri.mesh_index;
            h.triangle_index = bi.tri.triangle_index;
            h.leaf_index = bi.compact_index;
            h.hit = 1;
            h.t = bi.isect.t;
            h.b0 = bi.isect.b[0];
            h.b1 = bi.isect.b[1];
            h.b2 = bi.isect.b[2];
        }
        out[i] = h;
    }
    if (counters_out) {
        counters_out[0] = cnt.nodes;
        counters_out[1] = cnt.tris;
    }
    return 0;
}
int orc_occluded_batch(orc_scene* s, const float* from, const float* to, uint32_t n, uint8_t* out) {
    if (!s) return -1;
    for (uint32_t i = 0; i < n; i++)
        out[i] = s->scene.occluded(v3(from[3 * i], from[3 * i + 1], from[3 * i + 2]), v3(to[3 * i], to[3 * i + 1], to[3 * i + 2])) ? 1 : 0;
    return 0;
}

// Integrator::render + Film::create_pixels(1.0) + film_scale. n_threads <= 0 -> hardware_concurrency.
// stats_out[8]: samples, extend_rays, shadow_rays, segments, nodes, tris, stree_nodes, dtree_nodes.
// max_waves: guided only, stop after that many waves (0 = all).
// Returns 2 (film and stats filled, the offending samples left out) when a `path` render met a sample whose radiance is
// not finite: PathIntegrator::render asserts there (path_tracer.rs:46); the other two integrators' loops do not check.
int orc_render(orc_scene* s, const ptgpu_camera_desc* cam, const ptgpu_integrator_desc* integ, uint32_t spp, uint32_t shard_index,
               uint32_t shard_count, int n_threads, uint32_t max_waves, double* film_out, double* film_scale_out, uint64_t* stats_out) {
    if (!s || !cam || !integ || shard_count == 0) return -1;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    Camera camera;
    camera.init(cam->world_to_camera, cam->fov_deg, cam->resx, cam->resy);
    IntegratorDesc it;
    it.kind = (IntegratorKind)integ->kind;
    it.max_depth = integ->max_depth;
    it.mode = (Mode)integ->mode;
    it.spatial_threshold = integ->spatial_threshold;
    it.directional_threshold = integ->directional_threshold;
    it.spatial_filter = (Filter)integ->spatial_filter;
    it.directional_filter = (Filter)integ->directional_filter;
    Film film;
    film.init(cam->resx, cam->resy);
    RayStats st;
    GuidedReport rep;
    TraversalCounters tc;
    const uint64_t nonfinite_before = nonfinite_samples().load();
    // traversal counters are not thread safe: only collect them single-threaded
    s->scene.counters = (n_threads == 1) ? &tc : nullptr;
    switch (it.kind) {
        case IntegratorKind::Path: path_render(s->scene, camera, it, spp, shard_index, shard_count, film, n_threads, &st); break;
        case IntegratorKind::LightPath: lightpath_render(s->scene, camera, it, spp, shard_index, shard_count, film, n_threads, &st); break;
        case IntegratorKind::Guided: guided_render(s->scene, camera, it, spp, film, n_threads, &st, &rep, max_waves ? max_waves : ~size_t(0)); break;
    }
    s->scene.counters = nullptr;
    if (film_out)
        for (size_t i = 0; i < film.pixels.size(); i++) film_out[i] = film.pixels[i] + film.splats[i];
    if (film_scale_out) *film_scale_out = film_scale(it, spp);
    if (stats_out) {
        uint64_t my = 0;
        for (uint32_t i = shard_index; i < spp; i += shard_count) my++;
        stats_out[0] = (it.kind == IntegratorKind::Guided ? (uint64_t)spp : my) * cam->resx * cam->resy;
        stats_out[1] = st.extend_rays.load();
        stats_out[2] = st.shadow_rays.load();
        stats_out[3] = st.segments.load();
        stats_out[4] = tc.nodes;
        stats_out[5] = tc.tris;
        stats_out[6] = rep.stree_nodes;
        stats_out[7] = rep.dtree_nodes_total;
    }
    if (it.kind == IntegratorKind::Path && nonfinite_samples().load() != nonfinite_before) return 2;
    return 0;
}

uint64_t orc_nonfinite_samples(void) { return nonfinite_samples().load(); }  // dropped since the library was loaded

// ---- unit hooks (tests/test_oracle_units.py)
void orc_pcg_stream(uint64_t x, uint64_t y, uint64_t i, uint32_t n, uint32_t* out_u32) {
    Sampler s;
    s.init_pixel(x, y, i);
    for (uint32_t k = 0; k < n; k++) out_u32[k] = s.pcg.u32();
}
void orc_pcg_raw(uint64_t seq, uint64_t seed, uint32_t n, uint32_t* out_u32) {  // pcg32_srandom_r(seed, seq) semantics
    PCG p;
    p.set_sequence_seeded(seq, seed);
    for (uint32_t k = 0; k < n; k++) out_u32[k] = p.u32();
}
void orc_pcg_floats(uint64_t x, uint64_t y, uint64_t i, uint32_t n, float* out) {
    Sampler s;
    s.init_pixel(x, y, i);
    for (uint32_t k = 0; k < n; k++) out[k] = s.sample1d();
}
void orc_wavelength_iterates(uint32_t n, float* out) {
    Sampler s;
    s.init_pixel(0, 0, 0);
    for (uint32_t k = 0; k < n; k++) out[k] = s.wavelength_sample();
}
void orc_sample_visible(float u, float* lambdas8, float* pdfs8) {
    SWL w = SWL::sample_visible(u);
    std::memcpy(lambdas8, w.lambdas, 32);
    std::memcpy(pdfs8, w.pdfs, 32);
}
float orc_fresnel(float cos_i, float eta) { return fresnel(cos_i, eta); }
float orc_complex_fresnel(float cos_i, float eta, float k) { return complex_fresnel(cos_i, Cpx{eta, k}); }
void orc_csqrt(float re, float im, float* out2) {
    Cpx r = csqrt(Cpx{re, im});
    out2[0] = r.re;
    out2[1] = r.im;
}
void orc_spectrum_eval(const ptgpu_spectrum* s, int illuminant, const float* lambdas, uint32_t n, float* out) {
    Spectrum sp = to_spec(*s, illuminant != 0);
    for (uint32_t i = 0; i < n; i++) out[i] = sp.eval1(lambdas[i]);
}
// spectrum (8 values at the wavelengths of sample_visible(u)) -> linear RGB, as SamplePixel::add does
void orc_spectrum_to_rgb(const float* l8, float u, double* rgb3) {
    SWL w = SWL::sample_visible(u);
    SSpec s;
    std::memcpy(s.v, l8, 32);
    spectrum_to_rgb(s, w, rgb3);
}
// BSDF hooks in shading space. kind: 0 lambert, 1 dielectric, 2 conductor; a8/b8: albedo|eta , k.
int orc_bsdf_sample(int kind, float roughness, const float* a8, const float* b8, const float* wo3, float u1, float u2a, float u2b, int radiance,
                    float* f8, float* wi3, float* pdf, int* flags) {
    BSDF b;
    b.kind = (MatKind)kind;
    b.roughness = roughness;
    std::memcpy(b.a.v, a8, 32);
    std::memcpy(b.b.v, b8, 32);
    BSDFSample s;
    const float u2[2] = {u2a, u2b};
    if (!b.sample_local(v3(wo3[0], wo3[1], wo3[2]), u1, u2, radiance ? TransportMode::Radiance : TransportMode::Importance, &s)) return 0;
    std::memcpy(f8, s.f.v, 32);
    wi3[0] = s.wi.x; wi3[1] = s.wi.y; wi3[2] = s.wi.z;
    *pdf = s.pdf;
    *flags = (s.specular ? 1 : 0) | (s.terminate_secondary ? 2 : 0);
    return 1;
}
void orc_bsdf_eval(int kind, float roughness, const float* a8, const float* b8, const float* wo3, const float* wi3, float* f8, float* pdf) {
    BSDF b;
    b.kind = (MatKind)kind;
    b.roughness = roughness;
    std::memcpy(b.a.v, a8, 32);
    std::memcpy(b.b.v, b8, 32);
    SSpec f = b.eval_local(v3(wo3[0], wo3[1], wo3[2]), v3(wi3[0], wi3[1], wi3[2]));
    std::memcpy(f8, f.v, 32);
    *pdf = b.pdf_local(v3(wo3[0], wo3[1], wo3[2]), v3(wi3[0], wi3[1], wi3[2]));
}
int orc_ray_triangle(const float* o3, const float* d3, const float* p9, float max_t, float* tb4) {
    TriIsect ti;
    if (!ray_triangle_intersection(v3(o3[0], o3[1], o3[2]), v3(d3[0], d3[1], d3[2]), v3(p9[0], p9[1], p9[2]), v3(p9[3], p9[4], p9[5]),
                                   v3(p9[6], p9[7], p9[8]), max_t, &ti))
        return 0;
    tb4[0] = ti.t; tb4[1] = ti.b[0]; tb4[2] = ti.b[1]; tb4[3] = ti.b[2];
    return 1;
}
void orc_wave_sample_counts(uint32_t spp, uint32_t* n_out, uint32_t* counts_out, uint32_t max_counts) {
    auto w = wave_sample_counts(spp);
    *n_out = (uint32_t)w.size();
    for (size_t i = 0; i < w.size() && i < max_counts; i++) counts_out[i] = (uint32_t)w[i];
}
void orc_dir_square_roundtrip(const float* dir3, float* sq2, float* dir_back3) {
    V2 s = dir_to_square(v3(dir3[0], dir3[1], dir3[2]));
    sq2[0] = s.x; sq2[1] = s.y;
    V3 d = square_to_dir(s);
    dir_back3[0] = d.x; dir_back3[1] = d.y; dir_back3[2] = d.z;
}
// DTree exercise: record n (p, radiance) pairs with the box filter, build, refine `rounds` times, then report
// node count, pdf at the query points and samples drawn with a PCG seeded at (seed_x, seed_y, 0).
int orc_dtree_exercise(const float* pts2, const float* rad, uint32_t n, int filter, uint32_t rounds, float threshold, const float* query2,
                       uint32_t n_query, float* pdf_out, uint32_t n_samples, float* samples2_out, uint32_t* n_nodes_out) {
    Sampler smp;
    smp.init_pixel(7, 11, 0);
    DTree collector = DTree::new_root((Filter)filter), sampler = DTree::new_root((Filter)filter);
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t i = 0; i < n; i++) collector.record(V2{pts2[2 * i], pts2[2 * i + 1]}, rad[i], 1.0f, smp);
        collector.build();
        collector.emit_into(sampler);
        collector.subdivide_and_clear(threshold);
    }
    for (uint32_t i = 0; i < n_query; i++) pdf_out[i] = sampler.pdf(V2{query2[2 * i], query2[2 * i + 1]});
    for (uint32_t i = 0; i < n_samples; i++) {
        V2 s = sampler.sample(smp);
        samples2_out[2 * i] = s.x;
        samples2_out[2 * i + 1] = s.y;
    }
    *n_nodes_out = (uint32_t)sampler.nodes.size();
    return 0;
}

}  // extern "C"

// ---- SD-tree handle: the same tree on both sides (tests/test_gpu_sdtree.py). Layout of the flat arrays = ptgpu_guided_tree_download.
struct orc_sdtree {
    Guider guider;
    RecordLog log;
    std::vector<uint32_t> leaf_id;  // per spatial node (0xffffffff for interior), numbered in node order like the product's host tree
    void renumber() {
        leaf_id.assign(guider.stree.nodes.size(), 0xffffffffu);
        uint32_t next = 0;
        for (size_t i = 0; i < guider.stree.nodes.size(); i++)
            if (guider.stree.nodes[i].is_leaf()) leaf_id[i] = next++;
    }
};

// A DTree from `n` flat nodes (4 sums + 4 u16 links each, links relative to the tree's first node).
static DTree dtree_from_flat(const float* rad, const uint16_t* child, uint32_t n, float weight, Filter f) {
    DTree d;
    d.filter = f;
    d.statistical_weight = weight;
    d.nodes.resize(n);
    for (uint32_t k = 0; k < n; k++)
        for (int c = 0; c < 4; c++) {
            d.nodes[k].radiance[c] = rad[4 * (size_t)k + c];
            d.nodes[k].children[c] = child[4 * (size_t)k + c];
        }
    return d;
}
// Reachable nodes in a canonical order (preorder by an explicit stack, children numbered as discovered), so that two
// trees of equal shape compare equal whatever their internal numbering / free slots. Returns the deepest level.
static uint32_t dtree_canonical(const DTree& d, std::vector<float>& rad, std::vector<uint16_t>& child) {
    struct It { uint16_t old_i; uint32_t new_i, depth; };
    const size_t base = rad.size() / 4;
    std::vector<It> stack{{0, 0, 1}};
    uint32_t counter = 1, deepest = 1;
    rad.resize(4 * (base + 1));
    child.resize(4 * (base + 1), 0);
    while (!stack.empty()) {
        const It it = stack.back();
        stack.pop_back();
        deepest = std::max(deepest, it.depth);
        const DTreeNode& n = d.nodes[it.old_i];
        for (int c = 0; c < 4; c++) {
            rad[4 * (base + it.new_i) + c] = n.radiance[c];
            if (n.is_leaf(c)) continue;
            const uint32_t id = counter++;
            rad.resize(4 * (base + counter));
            child.resize(4 * (base + counter), 0);
            child[4 * (base + it.new_i) + c] = (uint16_t)id;
            stack.push_back({n.children[c], id, it.depth + 1});
        }
    }
    return deepest;
}

extern "C" {

orc_sdtree* orc_sdtree_fresh(orc_scene* s, float spatial_threshold) {  // Guider::new (mod.rs:71-77)
    if (!s) return nullptr;
    orc_sdtree* t = new orc_sdtree();
    t->guider.stree.init(s->scene.bounds(), Filter::Stochastic, Filter::Box);
    t->guider.directional_threshold = 0.01f;
    t->guider.spatial_threshold = spatial_threshold;
    t->renumber();
    return t;
}
orc_sdtree* orc_sdtree_create(const float* bounds6, const uint32_t* spatial, uint32_t n_spatial, const uint32_t* leaf, uint32_t n_leaves, const float* s_rad,
                              const uint16_t* s_child, const float* s_weight, const float* c_rad, const uint16_t* c_child, const float* c_weight,
                              float spatial_threshold) {
    orc_sdtree* t = new orc_sdtree();
    STree& st = t->guider.stree;
    st.bounds = B3{v3(bounds6[0], bounds6[1], bounds6[2]), v3(bounds6[3], bounds6[4], bounds6[5])};
    st.filter = Filter::Stochastic;
    st.nodes.resize(n_spatial);
    for (uint32_t i = 0; i < n_spatial; i++) {
        STreeNode& n = st.nodes[i];
        n.left = spatial[4 * i];
        n.right = spatial[4 * i + 1];
        n.split_axis = spatial[4 * i + 2];
        if (n.left != 0) {
            if (n.left >= n_spatial || n.right >= n_spatial || n.left <= i || n.right <= i) { delete t; return nullptr; }
            st.nodes[n.left].depth = st.nodes[n.right].depth = n.depth + 1;  // children come after their parent
            n.sampler = DTree::no_alloc();
            n.collector = DTree::no_alloc();
            continue;
        }
        const uint32_t l = spatial[4 * i + 3];
        if (l >= n_leaves) { delete t; return nullptr; }
        const uint32_t* li = leaf + 5 * (size_t)l;
        n.sampler = dtree_from_flat(s_rad + 4 * (size_t)li[0], s_child + 4 * (size_t)li[0], li[1], s_weight[l], Filter::Box);
        n.collector = dtree_from_flat(c_rad + 4 * (size_t)li[2], c_child + 4 * (size_t)li[2], li[3], c_weight[l], Filter::Box);
    }
    t->guider.directional_threshold = 0.01f;
    t->guider.spatial_threshold = spatial_threshold;
    t->guider.has_started = true;
    t->renumber();
    return t;
}
void orc_sdtree_destroy(orc_sdtree* t) { delete t; }
// The filters of STree::new (stree.rs:21-34) on an existing tree: the spatial one, and the directional one of every DTree.
void orc_sdtree_set_filters(orc_sdtree* t, int32_t spatial, int32_t directional) {
    t->guider.stree.filter = (Filter)spatial;
    for (STreeNode& n : t->guider.stree.nodes) {
        n.sampler.filter = (Filter)directional;
        n.collector.filter = (Filter)directional;
    }
}

// out[5] = spatial nodes, leaves, reachable sampling nodes, reachable collector nodes, deepest quadtree level (root = 1)
void orc_sdtree_sizes(orc_sdtree* t, uint32_t* out) {
    std::vector<float> r;
    std::vector<uint16_t> c;
    uint32_t leaves = 0, ns = 0, nc = 0, deepest = 0;
    for (const STreeNode& n : t->guider.stree.nodes) {
        if (!n.is_leaf()) continue;
        leaves++;
        r.clear(); c.clear();
        deepest = std::max(deepest, dtree_canonical(n.sampler, r, c));
        ns += (uint32_t)(r.size() / 4);
        r.clear(); c.clear();
        deepest = std::max(deepest, dtree_canonical(n.collector, r, c));
        nc += (uint32_t)(r.size() / 4);
    }
    out[0] = (uint32_t)t->guider.stree.nodes.size();
    out[1] = leaves; out[2] = ns; out[3] = nc; out[4] = deepest;
}
// The tree in the flat layout, quadtrees in canonical numbering (arrays sized from orc_sdtree_sizes).
void orc_sdtree_export(orc_sdtree* t, float* bounds6, uint32_t* spatial, uint32_t* leaf, float* s_rad, uint16_t* s_child, float* s_weight, float* c_rad,
                       uint16_t* c_child, float* c_weight) {
    const STree& st = t->guider.stree;
    bounds6[0] = st.bounds.min.x; bounds6[1] = st.bounds.min.y; bounds6[2] = st.bounds.min.z;
    bounds6[3] = st.bounds.max.x; bounds6[4] = st.bounds.max.y; bounds6[5] = st.bounds.max.z;
    std::vector<float> sr, cr;
    std::vector<uint16_t> sc, cc;
    for (size_t i = 0; i < st.nodes.size(); i++) {
        const STreeNode& n = st.nodes[i];
        spatial[4 * i] = (uint32_t)n.left; spatial[4 * i + 1] = (uint32_t)n.right; spatial[4 * i + 2] = (uint32_t)n.split_axis;
        spatial[4 * i + 3] = t->leaf_id[i];
        if (!n.is_leaf()) continue;
        const uint32_t l = t->leaf_id[i];
        const uint32_t s0 = (uint32_t)(sr.size() / 4), c0 = (uint32_t)(cr.size() / 4);
        dtree_canonical(n.sampler, sr, sc);
        dtree_canonical(n.collector, cr, cc);
        leaf[5 * (size_t)l] = s0; leaf[5 * (size_t)l + 1] = (uint32_t)(sr.size() / 4) - s0;
        leaf[5 * (size_t)l + 2] = c0; leaf[5 * (size_t)l + 3] = (uint32_t)(cr.size() / 4) - c0;
        leaf[5 * (size_t)l + 4] = n.sampler.mean() == 0.0f ? 1u : 0u;
        s_weight[l] = n.sampler.statistical_weight;
        c_weight[l] = n.collector.statistical_weight;
    }
    std::memcpy(s_rad, sr.data(), sr.size() * 4); std::memcpy(s_child, sc.data(), sc.size() * 2);
    std::memcpy(c_rad, cr.data(), cr.size() * 4); std::memcpy(c_child, cc.data(), cc.size() * 2);
}
// Same 16 words per probe as ptgpu_guided_probe: STree::leaf, DTree::sample, DTree::pdf (twice).
void orc_sdtree_probe(orc_sdtree* t, const float* points, const float* query_sq, uint32_t n, uint64_t seed, uint32_t* out) {
    const STree& st = t->guider.stree;
    for (uint32_t i = 0; i < n; i++) {
        const V3 p = v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
        B3 lb;
        const size_t ni = st.leaf(p, &lb);
        const DTree& d = st.nodes[ni].sampler;
        Sampler smp;
        smp.init_pixel(i & 0xffffu, i >> 16, (size_t)seed);
        const V2 sq = d.sample(smp);
        const float pdf_s = d.pdf(sq), pdf_q = d.pdf(V2{query_sq[2 * (size_t)i], query_sq[2 * (size_t)i + 1]});
        uint32_t* o = out + 16 * (size_t)i;
        const float f[5] = {sq.x, sq.y, pdf_s, pdf_q, 0.0f};
        o[0] = t->leaf_id[ni];
        std::memcpy(o + 1, f, 16);
        o[5] = (uint32_t)smp.pcg.state; o[6] = (uint32_t)(smp.pcg.state >> 32); o[7] = 0;
        const float b[6] = {lb.min.x, lb.min.y, lb.min.z, lb.max.x, lb.max.y, lb.max.z};
        std::memcpy(o + 8, b, 24);
        const V2 dsq = dir_to_square(p);
        std::memcpy(o + 14, &dsq.x, 4);
        std::memcpy(o + 15, &dsq.y, 4);
    }
}
// records: n * 10 floats {p, wi, radiance, u0, u1, u2}: STree::record with the stochastic filter's draws replayed.
void orc_sdtree_record(orc_sdtree* t, const float* records, uint32_t n) {
    Sampler unused;
    for (uint32_t i = 0; i < n; i++) {
        const float* r = records + 10 * (size_t)i;
        t->guider.stree.record_drawn(v3(r[0], r[1], r[2]), v3(r[3], r[4], r[5]), r[6], r[7], r[8], r[9], unused);
    }
}
// Guider::add's arguments (mod.rs:106-115) of the waves traced while the log is on: 14 words per record in the layout of
// ptgpu_guided_captured_records (p, wi, radiance, u0, u1, u2, leaf id, pixel, pass, vertex). Returns the record count.
void orc_sdtree_log_records(orc_sdtree* t, int enable) {
    t->log.words.clear();
    t->guider.stree.log = enable ? &t->log : nullptr;
}
uint64_t orc_sdtree_logged_records(orc_sdtree* t, uint32_t* out, uint64_t capacity_records) {
    const uint64_t n = t->log.words.size() / 14;
    for (uint64_t i = 0; i < n && i < capacity_records; i++) {
        std::memcpy(out + 14 * i, t->log.words.data() + 14 * i, 14 * sizeof(uint32_t));
        out[14 * i + 10] = t->leaf_id[t->log.words[14 * i + 10]];  // leaf node -> leaf id (valid until the next refinement)
    }
    return n;
}
void orc_sdtree_new_iteration(orc_sdtree* t, uint32_t iteration) {  // Guider::new_iteration's refinement (mod.rs:372-383)
    t->guider.stree.new_iteration(t->guider.directional_threshold, t->guider.spatial_threshold, iteration);
    t->renumber();
}
// One wave of PathGuidingIntegrator::render on this tree: passes [first_pass, first_pass + n_passes) with Guider::iteration
// = `iteration`; the film holds this wave only (film.clear(), mod.rs:80). stats_out[4]: samples, extend, shadow, segments.
// shard_index / shard_count: this rank's (pass, tile) pairs of a multi-GPU job (guided_wave_passes); 0 / 1 = all of them.
int orc_guided_wave(orc_scene* s, orc_sdtree* t, const ptgpu_camera_desc* cam, const ptgpu_integrator_desc* integ, uint32_t iteration, uint64_t first_pass,
                    uint32_t n_passes, int n_threads, uint32_t shard_index, uint32_t shard_count, double* film_out, uint64_t* stats_out) {
    if (!s || !t || !cam || !integ || shard_count == 0 || shard_index >= shard_count) return -1;
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads <= 0) n_threads = 1;
    Camera camera;
    camera.init(cam->world_to_camera, cam->fov_deg, cam->resx, cam->resy);
    IntegratorDesc it;
    it.kind = IntegratorKind::Guided;
    it.max_depth = integ->max_depth;
    it.mode = (Mode)integ->mode;
    it.spatial_threshold = integ->spatial_threshold;
    it.directional_threshold = integ->directional_threshold;
    it.spatial_filter = Filter::Stochastic;
    it.directional_filter = Filter::Box;
    Film film;
    film.init(cam->resx, cam->resy);
    RayStats st;
    t->guider.iteration = iteration;
    t->guider.has_started = true;
    guided_wave_passes(s->scene, camera, it, t->guider, film, (size_t)first_pass, n_passes, n_threads, &st, shard_index, shard_count);
    if (film_out)
        for (size_t i = 0; i < film.pixels.size(); i++) film_out[i] = film.pixels[i] + film.splats[i];
    if (stats_out) {
        stats_out[0] = (uint64_t)n_passes * cam->resx * cam->resy;  // (of all shards together)
        stats_out[1] = st.extend_rays.load();
        stats_out[2] = st.shadow_rays.load();
        stats_out[3] = st.segments.load();
    }
    return 0;
}

}  // extern "C"
