// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header).
// CPU restatement of pt/src/geom.rs:19-66,180-287, pt/src/bvh.rs (whole file), pt/src/scene.rs:36-341,
// pt/src/lighting.rs:12-169, pt/src/bsdfs.rs (whole file), pt/src/sg.rs, pt/src/camera.rs:13-114.
#pragma once
#include <algorithm>
#include <memory>
#include <vector>

#include "sampling.hpp"
#include "spectrum.hpp"

namespace orc {

struct Ray {
    V3 o, d;
};

// ---------------------------------------------------------------- geom.rs
inline V3 triangle_cross(V3 p0, V3 p1, V3 p2) { return cross(p0 - p1, p2 - p1); }  // geom.rs:19-25
inline float triangle_area(V3 p0, V3 p1, V3 p2) { return mag(triangle_cross(p0, p1, p2)) * 0.5f; }
inline V3 triangle_normal(V3 p0, V3 p1, V3 p2) { return normalized(triangle_cross(p0, p1, p2)); }
inline V3 triangle_point(V3 p0, V3 p1, V3 p2, const float b[3]) { return p0 * b[0] + p1 * b[1] + p2 * b[2]; }

inline V3 offset_ray_origin(B3 bounds, V3 n, V3 w) {  // geom.rs:48-66
    V3 extent = vmax(bounds.extent(), splat3(F32_EPSILON));
    float d = std::fabs(dot(n, extent));
    V3 offset = d * n;
    if (dot(w, n) < 0.0f) offset = -offset;
    V3 po = bounds.centroid() + offset;
    for (int i = 0; i < 3; i++) {
        if (offset[i] > 0.0f) po.at(i) = next_up(po[i]);
        else po.at(i) = next_down(po[i]);
    }
    return po;
}

struct TriIsect {
    float t = 0.0f;
    float b[3] = {0, 0, 0};
};

inline V3 permute(V3 v, int kz) {
    switch (kz) {
        case 0: return v3(v.y, v.z, v.x);
        case 1: return v3(v.z, v.x, v.y);
        default: return v3(v.x, v.y, v.z);
    }
}
inline float difference_of_products(float a, float b, float c, float d) {  // geom.rs:195-200
    float cd = c * d;
    float dop = std::fma(a, b, -cd);
    float e = std::fma(-c, d, cd);
    return dop + e;
}
// geom.rs:180-287
inline bool ray_triangle_intersection(V3 ro, V3 rd, V3 p0, V3 p1, V3 p2, float max_t, TriIsect* out) {
    V3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    int kz = max_axis(vabs(rd));
    V3 d = permute(rd, kz);
    p0t = permute(p0t, kz);
    p1t = permute(p1t, kz);
    p2t = permute(p2t, kz);
    float sx = -d.x / d.z, sy = -d.y / d.z, sz = 1.0f / d.z;
    p0t.x += sx * p0t.z;
    p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z;
    p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z;
    p2t.y += sy * p2t.z;
    float e0 = difference_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
    float e1 = difference_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
    float e2 = difference_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double p2txp1ty = (double)p2t.x * (double)p1t.y, p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y, p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y, p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0.0f || e1 < 0.0f || e2 < 0.0f) && (e0 > 0.0f || e1 > 0.0f || e2 > 0.0f)) return false;
    float det = e0 + e1 + e2;
    if (det == 0.0f) return false;
    p0t.z *= sz;
    p1t.z *= sz;
    p2t.z *= sz;
    float t_scaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0.0f && (t_scaled >= 0.0f || t_scaled < max_t * det)) return false;
    if (det > 0.0f && (t_scaled <= 0.0f || t_scaled > max_t * det)) return false;
    float inv_det = 1.0f / det;
    float t = t_scaled * inv_det;
    float max_zt = rmax(rmax(std::fabs(p0t.z), std::fabs(p1t.z)), std::fabs(p2t.z));
    float delta_z = numeric_gamma(3.0f) * max_zt;
    float max_xt = rmax(rmax(std::fabs(p0t.x), std::fabs(p1t.x)), std::fabs(p2t.x));
    float max_yt = rmax(rmax(std::fabs(p0t.y), std::fabs(p1t.y)), std::fabs(p2t.y));
    float delta_x = numeric_gamma(5.0f) * (max_xt + max_zt);
    float delta_y = numeric_gamma(5.0f) * (max_yt + max_zt);
    float delta_e = 2.0f * (numeric_gamma(2.0f) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = rmax(rmax(std::fabs(e0), std::fabs(e1)), std::fabs(e2));
    float delta_t = 3.0f * (numeric_gamma(3.0f) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * std::fabs(inv_det);
    if (t <= delta_t) return false;
    out->t = t;
    out->b[0] = e0 * inv_det;
    out->b[1] = e1 * inv_det;
    out->b[2] = e2 * inv_det;
    return true;
}

// ---------------------------------------------------------------- bvh.rs
struct BvhMesh {
    std::vector<V3> points;
    std::vector<uint32_t> tris;  // 3 per triangle
};
struct CompactTriangle {
    uint32_t mesh_index = 0, triangle_index = 0;
};
struct BvhNode {  // bvh.rs:30-36
    B3 bounds;
    uint32_t offset;
    uint16_t n_tris;
    uint8_t split_axis;
};
struct BvhIsect {
    CompactTriangle tri;
    TriIsect isect;
    uint32_t compact_index = 0;  // oracle-only bookkeeping: index into compact_triangles (leaf order)
};
struct TraversalCounters {
    uint64_t nodes = 0, tris = 0;
};

static constexpr int N_SAH_BUCKETS = 12;

struct BuildTriangle {
    B3 bounds;
    size_t index;
};
struct BuildNode {
    bool leaf = false;
    B3 bounds;
    size_t start = 0, end = 0;
    std::unique_ptr<BuildNode> left, right;
    int split_axis = 0;
};

inline size_t bucket_index(B3 centroid_bounds, V3 p, int axis) {  // bvh.rs:154-156
    size_t b = f32_as_usize((float)N_SAH_BUCKETS * centroid_bounds.offset(p)[axis]);
    return std::min<size_t>(N_SAH_BUCKETS - 1, b);
}

// itertools::partition (itertools 0.1x): swaps from both ends; returns the split index.
template <class Pred>
inline size_t itertools_partition(BuildTriangle* a, size_t n, Pred pred) {
    size_t split_index = 0;
    size_t front = 0, back = n;  // [front, back) remaining
    while (front < back) {
        BuildTriangle& f = a[front];
        front++;
        if (!pred(f)) {
            bool found = false;
            while (back > front) {
                back--;
                if (pred(a[back])) {
                    std::swap(a[front - 1], a[back]);
                    found = true;
                    break;
                }
            }
            if (!found) return split_index;
        }
        split_index++;
    }
    return split_index;
}

inline std::unique_ptr<BuildNode> build_recursive(const std::vector<CompactTriangle>& original,
                                                  std::vector<CompactTriangle>& compact, BuildTriangle* bt, size_t n,
                                                  size_t* n_nodes) {  // bvh.rs:177-275
    *n_nodes += 1;
    B3 bounds = B3::empty();
    for (size_t i = 0; i < n; i++) bounds = bounds | bt[i].bounds;
    auto create_leaf = [&]() {
        auto node = std::make_unique<BuildNode>();
        node->leaf = true;
        node->bounds = bounds;
        node->start = compact.size();
        for (size_t i = 0; i < n; i++) compact.push_back(original[bt[i].index]);
        node->end = compact.size();
        return node;
    };
    if (n == 1 || bounds.surface_area() == 0.0f) return create_leaf();
    B3 centroid_bounds = B3::empty();
    for (size_t i = 0; i < n; i++) centroid_bounds = centroid_bounds | bt[i].bounds.centroid();
    int split_axis = max_axis(centroid_bounds.extent());
    struct Bucket {
        B3 bounds = B3::empty();
        float count = 0.0f;
    } buckets[N_SAH_BUCKETS];
    for (size_t i = 0; i < n; i++) {
        size_t idx = bucket_index(centroid_bounds, bt[i].bounds.centroid(), split_axis);
        buckets[idx].bounds = buckets[idx].bounds | bt[i].bounds;
        buckets[idx].count += 1.0f;
    }
    float costs[N_SAH_BUCKETS - 1];
    for (float& c : costs) c = 0.0f;
    {
        B3 below = B3::empty();
        float cb = 0.0f;
        for (int i = 0; i < N_SAH_BUCKETS - 1; i++) {
            below = below | buckets[i].bounds;
            cb += buckets[i].count * below.surface_area();
            costs[i] += cb;
        }
    }
    {
        B3 above = B3::empty();
        float ca = 0.0f;
        for (int i = N_SAH_BUCKETS - 1; i >= 1; i--) {
            above = above | buckets[i].bounds;
            ca += buckets[i].count * above.surface_area();
            costs[i - 1] += ca;
        }
    }
    size_t min_index = 0;
    float min_cost = F32_INF;
    for (size_t i = 0; i < N_SAH_BUCKETS - 1; i++)
        if (costs[i] < min_cost) {
            min_index = i;
            min_cost = costs[i];
        }
    float leaf_cost = (float)n;
    float split_cost = 0.5f + min_cost / bounds.surface_area();
    if (leaf_cost < split_cost) return create_leaf();
    size_t mid = itertools_partition(bt, n, [&](const BuildTriangle& t) {
        return bucket_index(centroid_bounds, t.bounds.centroid(), split_axis) <= min_index;
    });
    if (mid == 0 || mid == n - 1) return create_leaf();  // sic (bvh.rs:257)
    auto node = std::make_unique<BuildNode>();
    node->bounds = bounds;
    node->split_axis = split_axis;
    node->left = build_recursive(original, compact, bt, mid, n_nodes);
    node->right = build_recursive(original, compact, bt + mid, n - mid, n_nodes);
    return node;
}

inline size_t flatten(std::vector<BvhNode>& nodes, const BuildNode* root) {  // bvh.rs:277-304
    size_t index = nodes.size();
    nodes.push_back(BvhNode{B3::empty(), 0, 0, 0});
    if (root->leaf) {
        nodes[index].bounds = root->bounds;
        nodes[index].offset = (uint32_t)root->start;
        nodes[index].n_tris = (uint16_t)(root->end - root->start);
    } else {
        nodes[index].bounds = root->bounds;
        nodes[index].split_axis = (uint8_t)root->split_axis;
        flatten(nodes, root->left.get());
        nodes[index].offset = (uint32_t)flatten(nodes, root->right.get());
    }
    return index;
}

inline bool ray_bounds_hit(const B3& b, V3 o, V3 inv_dir, const bool dir_is_neg[3], float ray_end_t) {  // bvh.rs:306-327
    float tmin = 0.0f, tmax = ray_end_t;
    for (int i = 0; i < 3; i++) {
        float lo = dir_is_neg[i] ? b.max[i] : b.min[i];
        float hi = dir_is_neg[i] ? b.min[i] : b.max[i];
        float ti_min = inv_dir[i] * (lo - o[i]);
        float ti_max = (1.0f + 2.0f * numeric_gamma(3.0f)) * inv_dir[i] * (hi - o[i]);
        tmin = rmax(tmin, ti_min);
        tmax = rmin(tmax, ti_max);
    }
    return (0.0f <= tmin) && (tmin <= tmax);
}

struct Bvh {
    std::vector<BvhMesh> meshes;
    std::vector<CompactTriangle> compact_triangles;
    std::vector<BvhNode> nodes;

    void build(std::vector<BvhMesh>&& ms) {  // bvh.rs:48-85
        meshes = std::move(ms);
        std::vector<CompactTriangle> original;
        std::vector<BuildTriangle> bts;
        for (size_t mi = 0; mi < meshes.size(); mi++) {
            const BvhMesh& m = meshes[mi];
            for (size_t ti = 0; ti < m.tris.size() / 3; ti++) {
                uint32_t i = m.tris[3 * ti], j = m.tris[3 * ti + 1], k = m.tris[3 * ti + 2];
                V3 p0 = m.points[i], p1 = m.points[k], p2 = m.points[j];
                bts.push_back(BuildTriangle{B3::from_points(p0, p1) | p2, original.size()});
                original.push_back(CompactTriangle{(uint32_t)mi, (uint32_t)ti});
            }
        }
        compact_triangles.clear();
        compact_triangles.reserve(original.size());
        size_t n_nodes = 0;
        auto root = build_recursive(original, compact_triangles, bts.data(), bts.size(), &n_nodes);
        nodes.clear();
        nodes.reserve(n_nodes);
        flatten(nodes, root.get());
    }
    B3 bounds() const { return nodes[0].bounds; }

    bool intersect(Ray ray, float ray_end_t, BvhIsect* out, TraversalCounters* cnt = nullptr) const {  // bvh.rs:86-142
        TriIsect isect;
        isect.t = ray_end_t;
        CompactTriangle tri;
        uint32_t tri_k = 0;
        size_t stack[64];
        int sp = 0;
        stack[sp++] = 0;
        V3 inv_dir = splat3(1.0f) / ray.d;
        bool dir_is_neg[3] = {inv_dir.x < 0.0f, inv_dir.y < 0.0f, inv_dir.z < 0.0f};
        while (sp > 0) {
            size_t ni = stack[--sp];
            const BvhNode& node = nodes[ni];
            if (cnt) cnt->nodes++;
            if (ray_bounds_hit(node.bounds, ray.o, inv_dir, dir_is_neg, ray_end_t)) {
                if (node.n_tris > 0) {
                    for (uint32_t k = node.offset; k < node.offset + node.n_tris; k++) {
                        CompactTriangle ct = compact_triangles[k];
                        const BvhMesh& mesh = meshes[ct.mesh_index];
                        uint32_t i = mesh.tris[3 * ct.triangle_index], j = mesh.tris[3 * ct.triangle_index + 1],
                                 kk = mesh.tris[3 * ct.triangle_index + 2];
                        if (cnt) cnt->tris++;
                        TriIsect ni_;
                        if (ray_triangle_intersection(ray.o, ray.d, mesh.points[i], mesh.points[j], mesh.points[kk], isect.t, &ni_)) {
                            isect = ni_;
                            tri = ct;
                            tri_k = k;
                        }
                    }
                } else {
                    if (dir_is_neg[node.split_axis]) {
                        stack[sp++] = ni + 1;
                        stack[sp++] = node.offset;
                    } else {
                        stack[sp++] = node.offset;
                        stack[sp++] = ni + 1;
                    }
                }
            }
        }
        if (isect.t != ray_end_t) {
            out->isect = isect;
            out->tri = tri;
            out->compact_index = tri_k;
            return true;
        }
        return false;
    }
    bool hit(Ray ray, float max_t, TraversalCounters* cnt = nullptr) const {  // bvh.rs:143-145
        BvhIsect tmp;
        return intersect(ray, max_t, &tmp, cnt);
    }
};

// ---------------------------------------------------------------- sg.rs
namespace sg {
inline bool same_hemisphere(V3 u, V3 v) { return u.z * v.z > 0.0f; }
inline float cos_theta(V3 v) { return v.z; }
inline float abs_cos_theta(V3 v) { return std::fabs(v.z); }
inline float cos2theta(V3 v) { return v.z * v.z; }
inline float sin2theta(V3 v) { return rmax(1.0f - cos2theta(v), 0.0f); }
inline float sin_theta(V3 v) { return std::sqrt(sin2theta(v)); }
inline float tan2theta(V3 v) { return sin2theta(v) / cos2theta(v); }
inline float cos_phi(V3 v) {
    float st = sin_theta(v);
    return st == 0.0f ? 1.0f : rclamp(v.x / st, -1.0f, 1.0f);
}
inline float sin_phi(V3 v) {
    float st = sin_theta(v);
    return st == 0.0f ? 0.0f : rclamp(v.y / st, -1.0f, 1.0f);
}
}  // namespace sg

// ---------------------------------------------------------------- bsdfs.rs
enum class TransportMode { Radiance, Importance };
enum class MatKind : int32_t { Lambertian = 0, Dielectric = 1, Conductor = 2 };

struct BSDFEvalCtx {
    V3 ng;
    ONB frame;
    V3 ns() const { return frame.z; }
};
struct BSDFSample {
    SSpec f;
    V3 wi;
    float pdf = 0.0f;
    bool specular = false;
    bool terminate_secondary = false;
};

// num-complex 0.4.6 Complex<f32> (pinned by pt/Cargo.lock:342-367; source not vendored: PARITY UNPINNED,
// checked against Python cmath in tests/test_oracle_units.py).
struct Cpx {
    float re, im;
};
inline Cpx cmul(Cpx a, Cpx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
inline Cpx cadd(Cpx a, Cpx b) { return {a.re + b.re, a.im + b.im}; }
inline Cpx csub(Cpx a, Cpx b) { return {a.re - b.re, a.im - b.im}; }
inline float cnorm_sqr(Cpx a) { return a.re * a.re + a.im * a.im; }
inline Cpx cdiv(Cpx a, Cpx b) {
    float ns = cnorm_sqr(b);
    float re = a.re * b.re + a.im * b.im;
    float im = a.im * b.re - a.re * b.im;
    return {re / ns, im / ns};
}
inline Cpx csqrt(Cpx a) {
    if (a.im == 0.0f) {
        if (!std::signbit(a.re)) return {std::sqrt(a.re), a.im};
        float im = std::sqrt(-a.re);
        return std::signbit(a.im) ? Cpx{0.0f, -im} : Cpx{0.0f, im};
    } else if (a.re == 0.0f) {
        float x = std::sqrt(std::fabs(a.im) / 2.0f);
        return std::signbit(a.im) ? Cpx{x, -x} : Cpx{x, x};
    } else {
        float r = std::hypot(a.re, a.im);
        float theta = std::atan2(a.im, a.re);
        float sr = std::sqrt(r), ht = theta / 2.0f;
        return {sr * std::cos(ht), sr * std::sin(ht)};
    }
}

inline float fresnel(float cos_theta_i, float eta) {  // bsdfs.rs:316-331
    cos_theta_i = rclamp(cos_theta_i, -1.0f, 1.0f);
    if (cos_theta_i < 0.0f) {
        eta = 1.0f / eta;
        cos_theta_i = -cos_theta_i;
    }
    float sin2theta_i = 1.0f - cos_theta_i * cos_theta_i;
    float sin2theta_t = sin2theta_i / (eta * eta);
    if (sin2theta_t >= 1.0f) return 1.0f;
    float cos_theta_t = safe_sqrt(1.0f - sin2theta_t);
    float r_parl = (eta * cos_theta_i - cos_theta_t) / (eta * cos_theta_i + cos_theta_t);
    float r_perp = (cos_theta_i - eta * cos_theta_t) / (cos_theta_i + eta * cos_theta_t);
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
inline float complex_fresnel(float cos_theta_i, Cpx eta) {  // bsdfs.rs:333-343
    cos_theta_i = rclamp(cos_theta_i, 0.0f, 1.0f);
    float sin2theta_i = 1.0f - cos_theta_i * cos_theta_i;
    Cpx sin2theta_t = cdiv(Cpx{sin2theta_i, 0.0f}, cmul(eta, eta));
    Cpx cos_theta_t = csqrt(csub(Cpx{1.0f, 0.0f}, sin2theta_t));
    Cpx ci{cos_theta_i, 0.0f};
    Cpx r_parl = cdiv(csub(cmul(eta, ci), cos_theta_t), cadd(cmul(eta, ci), cos_theta_t));
    Cpx r_perp = cdiv(csub(ci, cmul(eta, cos_theta_t)), cadd(ci, cmul(eta, cos_theta_t)));
    return (cnorm_sqr(r_parl) + cnorm_sqr(r_perp)) / 2.0f;
}
inline SSpec complex_fresnel_spectrum(float cos_theta_i, const SSpec& eta, const SSpec& k) {
    SSpec r;
    for (int i = 0; i < NSW; i++) r.v[i] = complex_fresnel(cos_theta_i, Cpx{eta.v[i], k.v[i]});
    return r;
}
inline bool ggx_effectively_smooth(float a) { return a < 0.0001f; }
inline V3 ggx_sample_wm(float a, V3 wo, const float u[2]) {  // bsdfs.rs:359-369
    V3 wi = wo.z < 0.0f ? -wo : wo;
    V3 wi_std = v3(wi.x * a, wi.y * a, wi.z);
    float phi = (2.0f * u[0] - 1.0f) * PI;
    float z = std::fma(1.0f - u[1], 1.0f + wi_std.z, -wi_std.z);
    float sin_theta = safe_sqrt(1.0f - z * z);
    float x = sin_theta * std::cos(phi);
    float y = sin_theta * std::sin(phi);
    V3 wm_std = wi_std + v3(x, y, z);
    return normalized(v3(wm_std.x * a, wm_std.y * a, wm_std.z));
}
inline float ggx_lambda(float a, V3 w) {  // bsdfs.rs:391-397
    float t2 = sg::tan2theta(w);
    if (std::isinf(t2)) return 0.0f;
    return (std::sqrt(1.0f + a * a * t2) - 1.0f) / 2.0f;
}
inline float ggx_g1(float a, V3 w) { return 1.0f / (1.0f + ggx_lambda(a, w)); }
inline float ggx_g(float a, V3 wo, V3 wi) { return 1.0f / (1.0f + ggx_lambda(a, wo) + ggx_lambda(a, wi)); }
inline float ggx_dwm(float a, V3 wm) {  // bsdfs.rs:370-378
    float t2 = sg::tan2theta(wm);
    if (std::isinf(t2)) return 0.0f;
    float c2 = sg::cos2theta(wm);
    float cos4theta = c2 * c2;
    float cpa = sg::cos_phi(wm) / a, spa = sg::sin_phi(wm) / a;
    float e = t2 * (cpa * cpa + spa * spa);
    float ope = 1.0f + e;
    return 1.0f / (PI * a * a * cos4theta * (ope * ope));
}
inline float ggx_pdf(float a, V3 w, V3 wm) {  // ggx_dwwm, bsdfs.rs:379-384
    return ggx_g1(a, w) / sg::abs_cos_theta(w) * ggx_dwm(a, wm) * std::fabs(dot(w, wm));
}
inline V3 reflect(V3 w, V3 n) { return -w + 2.0f * dot(w, n) * n; }  // bsdfs.rs:289-291
inline bool refract(V3 w, V3 n, float eta, float* etap, V3* wt) {     // bsdfs.rs:293-314
    float cos_theta_i = dot(w, n);
    if (cos_theta_i < 0.0f) {
        eta = 1.0f / eta;
        cos_theta_i = -cos_theta_i;
        n = -n;
    }
    float sin2theta_i = rmax(0.0f, 1.0f - cos_theta_i * cos_theta_i);
    float sin2theta_t = sin2theta_i / (eta * eta);
    if (sin2theta_t < 1.0f) {
        float cos_theta_t = safe_sqrt(1.0f - sin2theta_t);
        *wt = -w / eta + (cos_theta_i / eta - cos_theta_t) * n;
        *etap = eta;
        return true;
    }
    return false;  // reference asserts (unreachable: fresnel()==1 at TIR and u<1)
}
inline float correct_shading_normals(V3 wo, V3 wi, V3 ng, V3 ns) {  // bsdfs.rs:130-134
    float a = dot(wo, ns) * dot(wi, ng);
    float b = dot(wo, ng) * dot(wi, ns);
    return b == 0.0f ? 0.0f : std::fabs(a / b);
}

struct BSDF {
    MatKind kind = MatKind::Lambertian;
    float roughness = 0.0f;
    SSpec a;  // albedo | eta
    SSpec b;  // k (conductor)

    bool is_specular() const {
        switch (kind) {
            case MatKind::Lambertian: return false;
            case MatKind::Dielectric: return true;
            default: return ggx_effectively_smooth(roughness);
        }
    }
    // local-space samplers; return false for None
    bool sample_local(V3 wo, float u1d, const float u2d[2], TransportMode mode, BSDFSample* s) const {
        switch (kind) {
            case MatKind::Lambertian: {  // bsdfs.rs:136-156
                V3 wi = cosine_hemisphere(u2d);
                if (!sg::same_hemisphere(wo, wi)) wi = -wi;
                float pdf = cosine_hemisphere_pdf(sg::abs_cos_theta(wi));
                if (pdf == 0.0f) return false;
                s->f = a * sg::abs_cos_theta(wi) * INV_PI;
                s->wi = wi;
                s->pdf = pdf;
                s->specular = false;
                s->terminate_secondary = false;
                return true;
            }
            case MatKind::Dielectric: {  // bsdfs.rs:172-205
                float reflectance = fresnel(sg::cos_theta(wo), a.v[0]);
                if (u1d < reflectance) {
                    s->wi = v3(-wo.x, -wo.y, wo.z);
                    s->f = SSpec::splat(reflectance);
                    s->pdf = reflectance;
                    s->specular = true;
                    s->terminate_secondary = false;
                    return true;
                }
                float etap;
                V3 wi;
                if (!refract(wo, v3(0.0f, 0.0f, 1.0f), a.v[0], &etap, &wi)) return false;
                float f = 1.0f - reflectance;
                if (mode == TransportMode::Radiance) f /= etap * etap;
                s->f = SSpec::splat(f);
                s->wi = wi;
                s->pdf = 1.0f - reflectance;
                s->specular = true;
                s->terminate_secondary = !a.is_uniform();
                return true;
            }
            default: {  // bsdfs.rs:220-259
                float al = roughness;
                if (ggx_effectively_smooth(al)) {
                    V3 wi = v3(-wo.x, -wo.y, wo.z);
                    s->f = complex_fresnel_spectrum(sg::cos_theta(wi), a, b);
                    s->wi = wi;
                    s->pdf = 1.0f;
                    s->specular = true;
                    s->terminate_secondary = false;
                    return true;
                }
                V3 wm = ggx_sample_wm(al, wo, u2d);
                V3 wi = reflect(wo, wm);
                if (!sg::same_hemisphere(wo, wi)) return false;
                float pdf = ggx_pdf(al, wo, wm) / (4.0f * std::fabs(dot(wo, wm)));
                float cos_o = sg::abs_cos_theta(wo), cos_i = sg::abs_cos_theta(wi);
                s->f = sg::abs_cos_theta(wi) * complex_fresnel_spectrum(std::fabs(dot(wo, wm)), a, b) * ggx_dwm(al, wm) *
                       ggx_g(al, wo, wi) / (4.0f * cos_o * cos_i);
                s->wi = wi;
                s->pdf = pdf;
                s->specular = false;
                s->terminate_secondary = false;
                return true;
            }
        }
    }
    SSpec eval_local(V3 wo, V3 wi) const {
        switch (kind) {
            case MatKind::Lambertian:  // bsdfs.rs:157-162
                if (!sg::same_hemisphere(wo, wi)) return SSpec::zero();
                return a * sg::abs_cos_theta(wi) * INV_PI;
            case MatKind::Dielectric: return SSpec::zero();
            default: {  // bsdfs.rs:261-279
                float al = roughness;
                if (ggx_effectively_smooth(al) || !sg::same_hemisphere(wo, wi) || mag_sq(wo + wi) == 0.0f) return SSpec::zero();
                V3 wm = normalized(wo + wi);
                float cos_o = sg::abs_cos_theta(wo), cos_i = sg::abs_cos_theta(wi);
                return sg::abs_cos_theta(wi) * complex_fresnel_spectrum(std::fabs(dot(wo, wm)), a, b) * ggx_dwm(al, wm) *
                       ggx_g(al, wo, wi) / (4.0f * cos_o * cos_i);
            }
        }
    }
    float pdf_local(V3 wo, V3 wi) const {
        switch (kind) {
            case MatKind::Lambertian:  // bsdfs.rs:164-170
                return sg::same_hemisphere(wo, wi) ? cosine_hemisphere_pdf(sg::abs_cos_theta(wi)) : 0.0f;
            case MatKind::Dielectric: return 0.0f;
            default: {  // bsdfs.rs:281-287
                float al = roughness;
                if (ggx_effectively_smooth(al) || !sg::same_hemisphere(wo, wi) || mag_sq(wo + wi) == 0.0f) return 0.0f;
                V3 wm = normalized(wo + wi);
                return ggx_pdf(al, wo, wm) / (4.0f * std::fabs(dot(wo, wm)));
            }
        }
    }
    // bsdfs.rs:66-127 (world-space wrappers)
    bool sample(const BSDFEvalCtx& ctx, V3 wo, float u1d, const float u2d[2], TransportMode mode, BSDFSample* out) const {
        V3 wol = ctx.frame.apply(wo);
        BSDFSample s;
        if (!sample_local(wol, u1d, u2d, mode, &s)) return false;
        V3 wi = ctx.frame.apply_inverse(s.wi);
        if (mode == TransportMode::Importance) s.f = s.f * SSpec::splat(correct_shading_normals(wo, wi, ctx.ng, ctx.ns()));
        s.wi = wi;
        *out = s;
        return true;
    }
    SSpec eval(const BSDFEvalCtx& ctx, V3 wo, V3 wi, TransportMode mode) const {
        SSpec spec = eval_local(ctx.frame.apply(wo), ctx.frame.apply(wi));
        // NB argument order (ns, ng) into a (.., ng, ns) function: bsdfs.rs:113 vs :130 (reciprocal factor; kept).
        if (mode == TransportMode::Importance) spec = spec * SSpec::splat(correct_shading_normals(wo, wi, ctx.ns(), ctx.ng));
        return spec;
    }
    float pdf(const BSDFEvalCtx& ctx, V3 wo, V3 wi) const { return pdf_local(ctx.frame.apply(wo), ctx.frame.apply(wi)); }
};

// ---------------------------------------------------------------- lighting.rs:12-169
struct LiSample {
    SSpec l;
    B3 bounds;
    V3 n, wl;
    float pdf;
    V3 p() const { return bounds.centroid(); }
};
struct LeSample {
    SSpec l;
    B3 bounds;
    V3 n, dir;
    float pdf_dir, pdf_pos;
    V3 p() const { return bounds.centroid(); }
    Ray spawn_ray() const { return Ray{offset_ray_origin(bounds, n, dir), dir}; }
};
struct AreaLight {
    V3 p0, p1, p2;
    ONB onb;
    const Spectrum* emission;
    AreaLight(V3 a, V3 b, V3 c, const Spectrum* e) : p0(a), p1(b), p2(c), onb(ONB::init_z(triangle_normal(a, b, c))), emission(e) {}
    V3 n() const { return onb.z; }
    SSpec emitted(V3 from, V3 to, const SWL& swl) const {
        V3 incoming = to - from;
        if (dot(incoming, n()) <= 0.0f) return SSpec::zero();
        return emission->eval(swl.lambdas);
    }
    bool sample_le(const SWL& swl, const float u0[2], const float u1[2], LeSample* out) const {
        float b[3];
        uniform_barycentric(u0, b);
        V3 p = triangle_point(p0, p1, p2, b);
        float pdf_pos = 1.0f / triangle_area(p0, p1, p2);
        V3 dir = cosine_hemisphere(u1);
        float pdf_dir = cosine_hemisphere_pdf(sg::cos_theta(dir));
        if (pdf_dir == 0.0f) return false;
        V3 abs_sum = vabs(p0) * b[0] + vabs(p1) * b[1] + vabs(p2) * b[2];
        V3 err = numeric_gamma(6.0f) * abs_sum;
        dir = onb.apply_inverse(dir);
        out->l = emission->eval(swl.lambdas);
        out->bounds = B3{p - err, p + err};
        out->n = n();
        out->dir = dir;
        out->pdf_dir = pdf_dir;
        out->pdf_pos = pdf_pos;
        return true;
    }
    bool sample_li_solid_angle(V3 from_p, const SWL& swl, const float u[2], LiSample* out) const {
        float b[3];
        uniform_barycentric(u, b);
        V3 p = triangle_point(p0, p1, p2, b);
        V3 wl = p - from_p;
        if (dot(n(), wl) >= 0.0f) return false;
        float dist2 = mag_sq(wl);
        wl = normalized(wl);
        float area_pdf = 1.0f / triangle_area(p0, p1, p2);
        float cos_theta = std::fabs(dot(wl, n()));
        float solid_angle_pdf = area_pdf * dist2 / cos_theta;
        if (solid_angle_pdf == 0.0f) return false;
        V3 abs_sum = vabs(p0) * b[0] + vabs(p1) * b[1] + vabs(p2) * b[2];
        V3 err = numeric_gamma(6.0f) * abs_sum;
        out->l = emission->eval(swl.lambdas);
        out->bounds = B3{p - err, p + err};
        out->n = n();
        out->wl = wl;
        out->pdf = solid_angle_pdf;
        return true;
    }
    float pdf_li_solid_angle(V3 from_p, V3 to_p) const {
        float area_pdf = 1.0f / triangle_area(p0, p1, p2);
        V3 wl = to_p - from_p;
        float dist2 = mag_sq(wl);
        wl = normalized(wl);
        float cos_theta = std::fabs(dot(wl, n()));
        if (cos_theta == 0.0f) return 0.0f;
        return area_pdf * dist2 / cos_theta;
    }
};

// ---------------------------------------------------------------- scene.rs
struct Material {
    MatKind kind;
    float roughness;
    Spectrum s0, s1;
};
struct SceneInteraction {
    BvhIsect bvh_isect;
    ONB shading_frame{};
    V3 geometric_normal{};
    B3 bounds{};
    V3 p() const { return bounds.centroid(); }
    V3 ng() const { return geometric_normal; }
    V3 ns() const { return shading_frame.z; }
    Ray spawn_ray(V3 d) const { return Ray{offset_ray_origin(bounds, ng(), d), d}; }
    BSDFEvalCtx bsdf_eval_ctx() const { return BSDFEvalCtx{ng(), shading_frame}; }
};

struct InMesh {  // io_bridge::Mesh after spectrum upsampling
    std::vector<V3> points, normals;
    std::vector<uint32_t> tris;
    int32_t material_index = -1;
    bool emissive = false;
    Spectrum emission;
};

struct Scene {
    Bvh bvh;
    std::vector<std::vector<V3>> normals;
    std::vector<int32_t> material_indices;
    std::vector<Material> materials;
    std::vector<Spectrum> emissions;
    std::vector<CompactTriangle> emissive_triangles;
    mutable TraversalCounters* counters = nullptr;

    void build(std::vector<InMesh>&& meshes, std::vector<Material>&& mats) {  // scene.rs:89-183
        size_t n = meshes.size();
        material_indices.assign(n, -1);
        normals.assign(n, {});
        std::vector<BvhMesh> em, nonem;
        size_t n_emissive = 0;
        for (auto& m : meshes) n_emissive += m.emissive ? 1 : 0;
        for (auto& m : meshes) {
            if (!m.emissive) {
                size_t mesh_index = n_emissive + nonem.size();
                material_indices[mesh_index] = m.material_index;
                normals[mesh_index] = std::move(m.normals);
                nonem.push_back(BvhMesh{std::move(m.points), std::move(m.tris)});
            } else {
                size_t mesh_index = em.size();
                for (size_t t = 0; t < m.tris.size() / 3; t++) emissive_triangles.push_back(CompactTriangle{(uint32_t)mesh_index, (uint32_t)t});
                emissions.push_back(m.emission);
                material_indices[mesh_index] = m.material_index;
                normals[mesh_index] = std::move(m.normals);
                em.push_back(BvhMesh{std::move(m.points), std::move(m.tris)});
            }
        }
        materials = std::move(mats);
        for (auto& b : nonem) em.push_back(std::move(b));
        bvh.build(std::move(em));
    }
    void tri_points(CompactTriangle t, V3 p[3]) const {
        const BvhMesh& m = bvh.meshes[t.mesh_index];
        p[0] = m.points[m.tris[3 * t.triangle_index]];
        p[1] = m.points[m.tris[3 * t.triangle_index + 1]];
        p[2] = m.points[m.tris[3 * t.triangle_index + 2]];
    }
    bool tri_normals(CompactTriangle t, V3 nn[3]) const {
        const auto& ns = normals[t.mesh_index];
        if (ns.empty()) return false;
        const BvhMesh& m = bvh.meshes[t.mesh_index];
        nn[0] = ns[m.tris[3 * t.triangle_index]];
        nn[1] = ns[m.tris[3 * t.triangle_index + 1]];
        nn[2] = ns[m.tris[3 * t.triangle_index + 2]];
        return true;
    }
    bool intersect_t(Ray ray, float t, SceneInteraction* out) const {  // scene.rs:192-217
        BvhIsect bi;
        if (!bvh.intersect(ray, t, &bi, counters)) return false;
        V3 p[3];
        tri_points(bi.tri, p);
        const float* b = bi.isect.b;
        V3 pp = triangle_point(p[0], p[1], p[2], b);
        V3 abs_sum = vabs(p[0]) * b[0] + vabs(p[1]) * b[1] + vabs(p[2]) * b[2];
        V3 err = numeric_gamma(7.0f) * abs_sum;
        V3 ng = triangle_normal(p[0], p[1], p[2]);
        V3 ns = ng;
        V3 nn[3];
        if (tri_normals(bi.tri, nn)) ns = normalized((nn[0] * b[0]) + (nn[1] * b[1]) + (nn[2] * b[2]));
        out->bvh_isect = bi;
        out->shading_frame = ONB::init_z(ns);
        out->bounds = B3{pp - err, pp + err};
        out->geometric_normal = ng;
        return true;
    }
    bool intersect(Ray ray, SceneInteraction* out) const { return intersect_t(ray, F32_INF, out); }
    bool occluded(V3 from_p, V3 check) const {  // scene.rs:229-235
        Ray ray{from_p, check - from_p};
        return bvh.hit(ray, 0.9999f, counters);
    }
    bool occluded_interaction(const SceneInteraction& from, V3 check) const {
        V3 w = normalized(check - from.p());
        return occluded(offset_ray_origin(from.bounds, from.ng(), w), check);
    }
    bool occluded_le_sample(const LeSample& from, V3 check) const {
        V3 w = normalized(check - from.p());
        return occluded(offset_ray_origin(from.bounds, from.n, w), check);
    }
    bool compute_bsdf(const SceneInteraction& it, const SWL& swl, BSDF* out) const {  // scene.rs:236-256
        int32_t mi = material_indices[it.bvh_isect.tri.mesh_index];
        if (mi < 0) return false;
        const Material& m = materials[mi];
        out->kind = m.kind;
        out->roughness = m.roughness;
        out->a = m.s0.eval(swl.lambdas);
        out->b = m.kind == MatKind::Conductor ? m.s1.eval(swl.lambdas) : SSpec::zero();
        return true;
    }
    bool choose_light(float u1d, V3 p[3], const Spectrum** emission, float* pmf) const {  // scene.rs:257-268
        if (emissive_triangles.empty()) return false;
        size_t index = f32_as_usize((float)emissive_triangles.size() * u1d);
        CompactTriangle tri = emissive_triangles.at(index);
        tri_points(tri, p);
        *emission = &emissions[tri.mesh_index];
        *pmf = 1.0f / (float)emissive_triangles.size();
        return true;
    }
    bool is_light(const SceneInteraction& it) const { return it.bvh_isect.tri.mesh_index < emissions.size(); }
    AreaLight get_light(const SceneInteraction& it) const {  // scene.rs:269-277 (caller checks is_light)
        V3 p[3];
        tri_points(it.bvh_isect.tri, p);
        return AreaLight(p[0], p[1], p[2], &emissions[it.bvh_isect.tri.mesh_index]);
    }
    bool sample_li(V3 from_p, const SWL& swl, float u1d, const float u2d[2], LiSample* s, float* pmf) const {
        V3 p[3];
        const Spectrum* e;
        if (!choose_light(u1d, p, &e, pmf)) return false;
        return AreaLight(p[0], p[1], p[2], e).sample_li_solid_angle(from_p, swl, u2d, s);
    }
    bool sample_le(const SWL& swl, float u1d, const float u0[2], const float u1[2], LeSample* s, float* pmf) const {
        V3 p[3];
        const Spectrum* e;
        if (!choose_light(u1d, p, &e, pmf)) return false;
        return AreaLight(p[0], p[1], p[2], e).sample_le(swl, u0, u1, s);
    }
    B3 bounds() const { return bvh.bounds(); }
};

// ---------------------------------------------------------------- camera.rs:13-114
struct CameraWiSample {
    SSpec importance;
    V3 wc;
    float pdf;
    V2 p_raster;
};
struct Camera {
    float pixel_size;
    float image_size_f32[2];
    M4 world_to_camera, camera_to_world;
    size_t image_size[2];
    void init(const float w2c_rows[16], float fov, size_t resx, size_t resy) {
        image_size[0] = resx;
        image_size[1] = resy;
        // f32::to_radians: x * (PI/180)
        float half = (fov / 2.0f) * (PI / 180.0f);
        pixel_size = 2.0f * std::tan(half) / (float)resy;
        image_size_f32[0] = (float)resx;
        image_size_f32[1] = (float)resy;
        world_to_camera = M4::from_rows(w2c_rows);
        camera_to_world = world_to_camera.inverse();
    }
    V3 p() const { return camera_to_world.mul_point(v3(0, 0, 0)); }
    V3 n() const { return normalized(camera_to_world.mul_vector(v3(0.0f, 0.0f, 1.0f))); }
    Ray sample_we(size_t ix, size_t iy, const float u[2]) const {  // camera.rs:51-63
        float x = (image_size_f32[0] * 0.5f - (float)ix - u[0]) * pixel_size;
        float y = (image_size_f32[1] * 0.5f - (float)iy - u[1]) * pixel_size;
        return Ray{camera_to_world.mul_point(v3(0, 0, 0)), normalized(camera_to_world.mul_vector(v3(x, y, 1.0f)))};
    }
    bool sample_wi(V3 from_p, CameraWiSample* out) const {  // camera.rs:64-114
        V3 to_cam = p() - from_p;
        V3 wc = normalized(to_cam);
        V3 dir = world_to_camera.mul_vector(-wc);
        if (dir.z <= 0.0f) return false;
        V3 dn = dir / dir.z;
        float x = image_size_f32[0] / 2.0f - dn.x / pixel_size;
        float y = image_size_f32[1] / 2.0f - dn.y / pixel_size;
        if (!((0.0f <= x && x <= image_size_f32[0]) && (0.0f <= y && y <= image_size_f32[1]))) return false;
        float dist2 = mag_sq(to_cam);
        float cos_theta = std::fabs(dot(n(), wc));
        float c = sg::cos_theta(dir);
        float c4 = (c * c) * (c * c);  // powi(4)
        out->importance = SSpec::splat(1.0f / (pixel_size * pixel_size * image_size_f32[0] * image_size_f32[1] * c4));
        out->wc = wc;
        out->pdf = dist2 / cos_theta;
        out->p_raster = V2{x, y};
        return true;
    }
};

// camera.rs:116-196 (accumulators only; tiles flattened to one row-major f64 RGB image + a splat image)
struct Film {
    size_t w = 0, h = 0;
    std::vector<double> pixels, splats;  // 3 per pixel
    void init(size_t w_, size_t h_) {
        w = w_;
        h = h_;
        pixels.assign(w * h * 3, 0.0);
        splats.assign(w * h * 3, 0.0);
    }
    void clear() {
        std::fill(pixels.begin(), pixels.end(), 0.0);
        std::fill(splats.begin(), splats.end(), 0.0);
    }
};
inline void spectrum_to_rgb(const SSpec& l, const SWL& swl, double rgb[3]) {
    double xyz[3];
    to_xyz(l, swl, xyz);
    xyz_to_rgb(xyz, rgb);
}

}  // namespace orc
