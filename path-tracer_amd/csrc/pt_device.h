// pt_device.h — per-sample device functions of the wavefront path tracer (gfx950).
// Each block cites the reference code it replaces; the arithmetic (order, f32/f64 choice, fma sites)
// follows it so that integer/traversal results are bit-identical and shading differs only by libm ulps.
#pragma once
#include "pt_math.h"
#include "pt_types.h"

namespace pt {

// ------------------------------------------------------------------ sampling (pt/src/sampling.rs)
PT_HD uint64_t mix_bits(uint64_t v) {  // sampling.rs:149-156
    v ^= v >> 31;
    v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27;
    v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}
constexpr uint64_t kPcgMult = 0x5851f42d4c957f2dull;
struct Rng {  // PCG32 XSH-RR (sampling.rs:31-70); inc is a function of the pixel, state travels with the path
    uint64_t state, inc;
};
PT_HD uint32_t rng_u32(Rng& r) {
    uint64_t old = r.state;
    r.state = old * kPcgMult + r.inc;
    uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
    uint32_t rot = (uint32_t)(old >> 59);
    return (xs >> rot) | (xs << ((32u - rot) & 31u));
}
PT_HD float rng_f32(Rng& r) { return fminf(kOneMinusEps, (float)rng_u32(r) * 0x1p-32f); }
PT_HD uint64_t rng_inc_for_pixel(uint32_t x, uint32_t y) { return (mix_bits(((uint64_t)x << 32) | (uint64_t)y) << 1) | 1ull; }
// Sampler::init_pixel (sampling.rs:183-190): set_sequence(seed) then advance(0xffff * i).
PT_HD Rng rng_init_pixel(uint32_t x, uint32_t y, uint64_t i) {
    uint64_t seed = mix_bits(((uint64_t)x << 32) | (uint64_t)y);
    Rng r;
    r.state = 0;
    r.inc = (seed << 1) | 1ull;
    (void)rng_u32(r);
    r.state += mix_bits(seed);
    (void)rng_u32(r);
    uint64_t delta = 0xffffull * i;
    uint64_t cur_mult = kPcgMult, cur_plus = r.inc, acc_mult = 1, acc_plus = 0;
    while (delta > 0) {
        if (delta & 1) {
            acc_mult *= cur_mult;
            acc_plus = acc_plus * cur_mult + cur_plus;
        }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta >>= 1;
    }
    r.state = acc_mult * r.state + acc_plus;
    return r;
}

PT_HD void uniform_barycentric(float u0, float u1, float b[3]) {  // sampling.rs:202-213
    float b0 = 0.0f, b1 = 0.0f;
    if (u0 < u1) {
        b0 = u0 / 2.0f;
        b1 = u1 - b0;
    } else {
        b1 = u1 / 2.0f;
        b0 = u0 - b1;
    }
    b[0] = b0;
    b[1] = b1;
    b[2] = 1.0f - b0 - b1;
}
PT_D V3 cosine_hemisphere(float u0, float u1) {  // sampling.rs:215-219,251-255 (polar disk map)
    float r = sqrtf(u0);
    float theta = kTwoPi * u1;
    float s, c;
    sincosf(theta, &s, &c);
    float dx = r * c, dy = r * s;
    return mk3(dx, dy, safe_sqrtf(1.0f - dx * dx - dy * dy));
}
PT_HD float cosine_hemisphere_pdf(float cos_theta) { return cos_theta * kInvPi; }
// sampling.rs:273-283. With an all-zero input 0*inf = NaN, every compare is false and the last index wins.
PT_HD int sample_discrete4(float u, float v0, float v1, float v2, float v3) {
    float total = ((v0 + v1) + v2) + v3;
    float inv = 1.0f / total;
    float sum = 0.0f;
    sum += v0 * inv;
    if (sum >= u) return 0;
    sum += v1 * inv;
    if (sum >= u) return 1;
    sum += v2 * inv;
    if (sum >= u) return 2;
    return 3;
}

// ------------------------------------------------------------------ spectrum (pt/src/spectrum/mod.rs)
struct Spec {  // SampledSpectrum = [f32; 8]
    float v[NSW];
};
PT_HD Spec spec_splat(float x) {
    Spec s;
#pragma unroll
    for (int i = 0; i < NSW; i++) s.v[i] = x;
    return s;
}
PT_HD bool spec_is_black(const Spec& s) {
    bool b = true;
#pragma unroll
    for (int i = 0; i < NSW; i++) b = b && (s.v[i] == 0.0f);
    return b;
}
PT_HD bool spec_is_uniform(const Spec& s) {
    bool b = true;
#pragma unroll
    for (int i = 1; i < NSW; i++) b = b && (s.v[i] == s.v[0]);
    return b;
}
PT_HD float spec_average(const Spec& s) {  // mod.rs:94-96
    float a = 0.0f;
#pragma unroll
    for (int i = 0; i < NSW; i++) a += s.v[i];
    return a / (float)NSW;
}
PT_HD Spec operator+(Spec a, const Spec& b) {
#pragma unroll
    for (int i = 0; i < NSW; i++) a.v[i] += b.v[i];
    return a;
}
PT_HD Spec operator*(Spec a, const Spec& b) {
#pragma unroll
    for (int i = 0; i < NSW; i++) a.v[i] *= b.v[i];
    return a;
}
PT_HD Spec operator*(Spec a, float s) {
#pragma unroll
    for (int i = 0; i < NSW; i++) a.v[i] *= s;
    return a;
}
PT_HD Spec operator*(float s, Spec a) { return a * s; }
PT_HD Spec operator/(Spec a, float s) {
#pragma unroll
    for (int i = 0; i < NSW; i++) a.v[i] /= s;
    return a;
}

// Rust std f32::atanh: 0.5 * ln_1p(2|x| / (1 - |x|)) with the sign of x.
PT_D float atanh_rs(float x) {
    float ax = fabsf(x);
    return copysignf(0.5f * log1pf((ax + ax) / (1.0f - ax)), x);
}
PT_D float sample_visible_lambda(float u) { return 538.0f - 138.888889f * atanh_rs(0.85691062f - 1.82750197f * u); }  // mod.rs:154-156
PT_D float visible_pdf(float lambda) {  // mod.rs:157-159
    float c = coshf(0.0072f * (lambda - 538.0f));
    return 0.0039398042f / (c * c);
}
struct Lambdas {  // SampledWavelengths (mod.rs:15-18) before any terminate_secondary
    float l[NSW];
    float pdf[NSW];
};
PT_D Lambdas sample_visible(float u1d) {  // mod.rs:103-117
    Lambdas r;
#pragma unroll
    for (int i = 0; i < NSW; i++) {
        float u = u1d + (float)i / (float)NSW;
        if (u >= 1.0f) u -= 1.0f;
        r.l[i] = sample_visible_lambda(u);
        r.pdf[i] = visible_pdf(r.l[i]);
    }
    return r;
}
// The wavelength sample of a path is one of a few hundred table values (see ptgpu.hip, wavelength tables), so the
// 8 wavelengths and pdfs are evaluated once per table entry (k_build_lambda_table) and every stage reads the 64-byte
// row instead of redoing 8 atanh + 8 cosh per segment.
PT_D Lambdas load_lambdas(const float4* __restrict__ tab, uint32_t idx, bool with_pdf) {
    Lambdas r;
    const float4 a = tab[4 * (size_t)idx], b = tab[4 * (size_t)idx + 1];
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    if (with_pdf) {
        const float4 c = tab[4 * (size_t)idx + 2], d = tab[4 * (size_t)idx + 3];
        r.pdf[0] = c.x; r.pdf[1] = c.y; r.pdf[2] = c.z; r.pdf[3] = c.w;
        r.pdf[4] = d.x; r.pdf[5] = d.y; r.pdf[6] = d.z; r.pdf[7] = d.w;
    } else {
#pragma unroll
        for (int i = 0; i < NSW; i++) r.pdf[i] = 0.0f;
    }
    return r;
}
PT_HD uint32_t lambda_index(float l) {  // (l - 360) as usize, saturating
    float d = l - 360.0f;
    if (!(d > 0.0f)) return 0;
    uint32_t i = (uint32_t)d;
    return i > (uint32_t)(CIE_N - 1) ? (uint32_t)(CIE_N - 1) : i;
}
// Spectrum::eval (mod.rs:303-335): sigmoid polynomial * scale (* D65 for illuminants).
PT_D float spectrum_eval1(const SpectrumDev& s, float lambda, const float* __restrict__ d65) {
    float x = fmaf(lambda, fmaf(lambda, s.c0, s.c1), s.c2);
    float sig;
    if (isinf(x)) sig = x > 0.0f ? 1.0f : 0.0f;
    else sig = 0.5f + x / (2.0f * sqrtf(1.0f + x * x));
    float v = sig * s.scale;
    if (s.illuminant) v = v * d65[lambda_index(lambda)];
    return v;
}
PT_D Spec spectrum_eval(const SpectrumDev& s, const Lambdas& wl, const float* __restrict__ d65) {
    Spec r;
#pragma unroll
    for (int i = 0; i < NSW; i++) r.v[i] = spectrum_eval1(s, wl.l[i], d65);
    return r;
}
// SampledSpectrum::to_xyz + xyz_to_rgb (mod.rs:81-90,130-136), f64. `secondary_terminated` reproduces
// SampledWavelengths::terminate_secondary (mod.rs:118-127): pdf[0] /= 8, the others 0.
PT_D void spec_to_rgb(const Spec& l, const Lambdas& wl, bool secondary_terminated, const double* __restrict__ cie_xyz, double rgb[3]) {
    double X = 0.0, Y = 0.0, Z = 0.0;
#pragma unroll
    for (int i = 0; i < NSW; i++) {
        float pdf = wl.pdf[i];
        if (secondary_terminated) pdf = (i == 0) ? pdf / (float)NSW : 0.0f;
        if (pdf != 0.0f) {
            const double* c = cie_xyz + 3 * lambda_index(wl.l[i]);
            double w = (double)(l.v[i] / pdf);
            X += c[0] * w;
            Y += c[1] * w;
            Z += c[2] * w;
        }
    }
    X /= (double)NSW;
    Y /= (double)NSW;
    Z /= (double)NSW;
    rgb[0] = X * 3.240479 + Y * -1.537150 + Z * -0.498535;
    rgb[1] = X * -0.969256 + Y * 1.875991 + Z * 0.041556;
    rgb[2] = X * 0.055648 + Y * -0.204043 + Z * 1.057311;
}

// ------------------------------------------------------------------ geometry (pt/src/geom.rs)
PT_HD V3 tri_cross(V3 p0, V3 p1, V3 p2) { return cross(p0 - p1, p2 - p1); }  // geom.rs:19-25
PT_HD float tri_area(V3 p0, V3 p1, V3 p2) { return len(tri_cross(p0, p1, p2)) * 0.5f; }
PT_HD V3 tri_normal(V3 p0, V3 p1, V3 p2) { return normalize(tri_cross(p0, p1, p2)); }
PT_HD V3 tri_point(V3 p0, V3 p1, V3 p2, const float b[3]) { return p0 * b[0] + p1 * b[1] + p2 * b[2]; }

// offset_ray_origin (geom.rs:48-66): leave the error box along +-n, then one ulp outward per component.
PT_HD V3 offset_ray_origin(V3 blo, V3 bhi, V3 n, V3 w) {
    V3 extent = max3(bhi - blo, splat(kEps));
    float d = fabsf(dot(n, extent));
    V3 off = d * n;
    if (dot(w, n) < 0.0f) off = -off;
    V3 po = (blo + bhi) / 2.0f + off;
    po.x = off.x > 0.0f ? next_up(po.x) : next_down(po.x);
    po.y = off.y > 0.0f ? next_up(po.y) : next_down(po.y);
    po.z = off.z > 0.0f ? next_up(po.z) : next_down(po.z);
    return po;
}

// offset_ray_origin with its direction-independent part computed ahead: `mag` = offset_magnitude(blo, bhi, n), `p` =
// the box's centre (the interaction point). Same operations, same bits; a caller that must keep the surface over long
// code keeps one float instead of the box.
PT_HD float offset_magnitude(V3 blo, V3 bhi, V3 n) { return fabsf(dot(n, max3(bhi - blo, splat(kEps)))); }
PT_HD V3 offset_ray_origin_from(V3 p, V3 n, float mag, V3 w) {
    V3 off = mag * n;
    if (dot(w, n) < 0.0f) off = -off;
    V3 po = p + off;
    po.x = off.x > 0.0f ? next_up(po.x) : next_down(po.x);
    po.y = off.y > 0.0f ? next_up(po.y) : next_down(po.y);
    po.z = off.z > 0.0f ? next_up(po.z) : next_down(po.z);
    return po;
}

// (component by component: as a choice between three whole vectors the compiler turns it into two nested branches with
// register moves, ~17 instructions per vertex in the triangle loop where six selects do)
PT_HD V3 permute_kz(V3 v, int kz) {
    const bool k0 = kz == 0, k1 = kz == 1;
    return mk3(k0 ? v.y : (k1 ? v.z : v.x), k0 ? v.z : (k1 ? v.x : v.y), k0 ? v.x : (k1 ? v.y : v.z));
}
PT_HD float diff_of_products(float a, float b, float c, float d) {  // geom.rs:195-200
    float cd = c * d;
    float dop = fmaf(a, b, -cd);
    float e = fmaf(-c, d, cd);
    return dop + e;
}
struct TriHit {
    float t, b0, b1, b2;
};
// Per-ray part of ray_triangle_intersection (geom.rs:183-194): dominant axis and shear. It depends on the ray only,
// so the traversal computes it once per ray instead of once per triangle (three IEEE divisions saved per test;
// the values, hence the results, are the same bits).
struct RayShear {
    int kz;
    float sx, sy, sz;
};
PT_HD RayShear ray_shear(V3 rd) {
    RayShear r;
    r.kz = max_axis(absv(rd));
    const V3 d = permute_kz(rd, r.kz);
    r.sx = -d.x / d.z;
    r.sy = -d.y / d.z;
    r.sz = 1.0f / d.z;
    return r;
}
// ray_triangle_intersection (geom.rs:180-287): PBRT-v4 watertight test, f64 fallback on a zero edge function.
PT_HD bool ray_triangle(V3 ro, const RayShear& rs, V3 p0, V3 p1, V3 p2, float max_t, TriHit& out) {
    V3 p0t = p0 - ro, p1t = p1 - ro, p2t = p2 - ro;
    p0t = permute_kz(p0t, rs.kz);
    p1t = permute_kz(p1t, rs.kz);
    p2t = permute_kz(p2t, rs.kz);
    const float sx = rs.sx, sy = rs.sy, sz = rs.sz;
    p0t.x += sx * p0t.z;
    p0t.y += sy * p0t.z;
    p1t.x += sx * p1t.z;
    p1t.y += sy * p1t.z;
    p2t.x += sx * p2t.z;
    p2t.y += sy * p2t.z;
    float e0 = diff_of_products(p1t.x, p2t.y, p1t.y, p2t.x);
    float e1 = diff_of_products(p2t.x, p0t.y, p2t.y, p0t.x);
    float e2 = diff_of_products(p0t.x, p1t.y, p0t.y, p1t.x);
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {
        double a = (double)p2t.x * (double)p1t.y, b = (double)p2t.y * (double)p1t.x;
        e0 = (float)(b - a);
        a = (double)p0t.x * (double)p2t.y;
        b = (double)p0t.y * (double)p2t.x;
        e1 = (float)(b - a);
        a = (double)p1t.x * (double)p0t.y;
        b = (double)p1t.y * (double)p0t.x;
        e2 = (float)(b - a);
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
    float max_zt = fmaxf(fmaxf(fabsf(p0t.z), fabsf(p1t.z)), fabsf(p2t.z));
    float delta_z = gamma_n(3.0f) * max_zt;
    float max_xt = fmaxf(fmaxf(fabsf(p0t.x), fabsf(p1t.x)), fabsf(p2t.x));
    float max_yt = fmaxf(fmaxf(fabsf(p0t.y), fabsf(p1t.y)), fabsf(p2t.y));
    float delta_x = gamma_n(5.0f) * (max_xt + max_zt);
    float delta_y = gamma_n(5.0f) * (max_yt + max_zt);
    float delta_e = 2.0f * (gamma_n(2.0f) * max_xt * max_yt + delta_y * max_xt + delta_x * max_yt);
    float max_e = fmaxf(fmaxf(fabsf(e0), fabsf(e1)), fabsf(e2));
    float delta_t = 3.0f * (gamma_n(3.0f) * max_e * max_zt + delta_e * max_zt + delta_z * max_e) * fabsf(inv_det);
    if (t <= delta_t) return false;
    out.t = t;
    out.b0 = e0 * inv_det;
    out.b1 = e1 * inv_det;
    out.b2 = e2 * inv_det;
    return true;
}

#ifdef PT_CENSUS_WAVE  // experiment: the census counts loop trips of the WAVE (one lane of those in the trip counts it)
#define PT_CENSUS_LEAD ((int)__lane_id() == __ffsll((long long)__ballot(1)) - 1)
#else
#define PT_CENSUS_LEAD true
#endif
// ------------------------------------------------------------------ BVH traversal (pt/src/bvh.rs:86-146,306-327)
// Node = two float4: {lo.xyz, bits(offset)}, {hi.xyz, bits(n_tris | split_axis << 16)}; interior nodes keep
// their left child at index+1 and `offset` = right child. The visit order (far child pushed first, near on
// top), the box test against the ORIGINAL ray_end_t and the "later triangle wins ties" rule are the
// reference's, so the reported primitive is the same one bit for bit.
//
// Culling. The reference tests every box against the ORIGINAL ray_end_t (bvh.rs:100-106), never against the hit found
// so far. One cull is provably invisible and is applied here: with kz the ray's dominant axis, every triangle inside a
// box has all three sheared depths z_i = fl(fl(p_i[kz] - o[kz]) * (1/d[kz])) >= t_dom, the box's entry distance on
// that axis computed by the very same two operations on a coordinate that bounds p_i[kz] (rounding is monotone). An
// accepted hit has edge functions of one sign, so t_scaled/det is a convex combination of the z_i up to 7 roundings
// without cancellation; hence t_dom > 0 and t_dom > best.t * (1 + 32 eps) imply t_scaled beyond max_t * det for every
// triangle below that node: the reference would reject them all (geom.rs:261-268). Skipping the node changes nothing.
// (The full slab entry max(t_x, t_y, t_z) would cull a little more, but the other two axes have no such bound in
// floating point: barycentrics are ill-conditioned at grazing incidence.)
// The test is written !(t_dom > best.t * (1 + 32 eps)): a NaN best.t culls nothing. The reference's triangle test
// accepts a NaN hit (its rejections are all comparisons, false on NaN; it happens when |p - o| exceeds ~1e19 and the
// edge functions overflow), after which it still visits every box: so does this walk. What the cull cannot follow is
// the reverse order: a finite hit first, then a triangle in a farther box whose test comes out NaN and REPLACES it in
// the reference. Scenes that reach across 1e19 units lose bit parity there (parity).
//
// Shape of the loop (CDNA4): the order in which nodes are visited does not depend on what the ray hits (only whether a node is skipped
// does). That allows a "while-while" walk — every lane first descends boxes until it stands on
// a leaf, then all lanes test their leaf's triangles together — instead of mixing box and triangle work in one
// divergent loop body. The near child stays in a register; the far-child stack lives in LDS (column `threadIdx.x` of
// a [depth][256] array, conflict-free, sized from the tree's depth at launch) instead of scratch memory. Up to
// kLeafQueue leaves wait (in LDS as well) before the triangle phase starts, so a lane with many small leaves does not
// force the whole wave through one box phase per leaf; leaves and triangles are still tested in the reference's order.
struct Hit {
    uint32_t prim;  // leaf-order triangle index, kNoHit = miss
    uint32_t tag;   // tri_v[3 * prim + 2].w of the hit triangle
    TriHit th;
};
// ANY_HIT (occlusion queries only): Scene::occluded keeps no more than "was anything hit" (scene.rs:229-235). Once
// a triangle is accepted clearly inside the segment, later triangles can only be accepted at t <= best (up to a few
// ulps), so the answer `best.t != ray_end_t` is already decided and the walk stops. Hits within 0.1 % of the end
// keep walking so that the reference's exact-equality rule (bvh.rs:137) sees the same final t.
// Small node arrays are read from LDS. What bounds the walk of a small scene is the vector-memory front end: a
// 16-byte-per-lane load occupies the CU's address/tag pipeline for >= 16 cycles whatever it hits (k_extend on the
// Cornell box: TCP_TOTAL_CACHE_ACCESSES ~0.86 per cycle per CU, TA busy 82 %), two per visited node plus three per
// tested triangle. LDS is a second, independent pipe: nodes from LDS, triangles from L1. Every kernel that walks the
// BVH calls this first (all threads).
template <bool WIDE = false>
PT_D void bvh_stage_nodes(const SceneDev& sc) {
    extern __shared__ uint32_t sm_bvh_stack[];
    if (WIDE) {
        if (sc.lds_wnodes) {
            float4* dst = reinterpret_cast<float4*>(sm_bvh_stack + (sc.wide_stack + kLeafQueue) * kBlock);
            for (uint32_t i = threadIdx.x; i < 8 * sc.n_wnodes; i += kBlock) dst[i] = sc.wnodes[i];
            __syncthreads();
        }
    } else if (sc.lds_nodes) {
        float4* dst = reinterpret_cast<float4*>(sm_bvh_stack + (sc.bvh_stack + 2 * kLeafQueue) * kBlock);
        for (uint32_t i = threadIdx.x; i < 2 * sc.n_nodes; i += kBlock) dst[i] = sc.nodes[i];
        __syncthreads();
    }
}
// The triangles of one leaf against the ray, in leaf order (shared by the binary and the wide walk). Returns true when
// an occlusion query is decided.
template <bool COUNT, bool ANY_HIT>
PT_D bool bvh_test_leaf(const SceneDev& sc, V3 o, const RayShear& shear, float ray_end_t, uint32_t leaf_first, uint32_t leaf_n, Hit& best, uint32_t* n_tris_tested) {
    for (uint32_t k = leaf_first; k < leaf_first + leaf_n; k++) {
        const float4* tv = sc.tri_v + 3 * (size_t)k;  // one address, two immediate offsets
        float4 a = tv[0], b = tv[1], c = tv[2];
        // Left alone, the compiler re-loads overlapping halves of each vertex (x,y and y,z, for its packed-math
        // pairings): six vector-memory instructions per triangle instead of three, and vector-memory issue is
        // what bounds this loop. Passing the loaded values through an empty asm pins one 16-byte load each.
        // (Closest-hit walks only: the occlusion kernels carry more state and would drop to 7 waves per SIMD.)
        if (!ANY_HIT) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(c.x), "+v"(c.y), "+v"(c.z));
        if (COUNT && PT_CENSUS_LEAD) (*n_tris_tested)++;
        TriHit th;
        if (ray_triangle(o, shear, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), best.th.t, th)) {
            best.th = th;
            best.prim = k;
            if (ANY_HIT && th.t < 0.999f * ray_end_t) return true;
        }
    }
    return false;
}
// The wide walk (bvh_build.h: WideNode): the same leaves in the same order as the binary walk below, a quarter of the
// loop trips. Per visited node: four slab tests (against the ORIGINAL ray_end_t, as the reference tests every box,
// plus the dominant-axis cull, which only ever skips what the reference would reject), then the children that passed
// go onto the LDS stack in reverse visit order — the visit order of a node's children depends on the ray's direction
// signs only and comes out of the node's table. Why skipping the collapsed inner nodes' box tests changes nothing: the
// reference reaches a node only through its ancestors' boxes, an ancestor's box contains the node's box exactly (a join
// of the same triangle bounds), and the slab test is monotone in the box (the products inv * (near - o) and
// slack * inv * (far - o) are monotone in near and far for a fixed ray, a NaN from 0 * inf is ignored by fmaxf/fminf
// exactly where the larger box gives -inf/+inf or NaN as well): a ray that passes the node's test passes every ancestor's.
// STAGED (nodes in LDS) is a template parameter, not a run-time choice of pointer: with `staged ? lds : global` the
// compiler merges both reads into one FLAT load of a selected generic pointer, which sends the LDS reads through the
// vector-memory front end (aperture check in the texture addresser, both wait counters) instead of ds_read_b128 and
// the global ones through flat_load instead of global_load. bvh_intersect branches once per ray (uniform) instead.
template <bool COUNT, bool ANY_HIT, bool STAGED>
PT_D Hit bvh_intersect_wide(const SceneDev& sc, V3 o, V3 d, float ray_end_t, uint32_t* n_nodes_visited, uint32_t* n_tris_tested) {
    extern __shared__ uint32_t sm_bvh_stack[];
    uint32_t* stack = sm_bvh_stack + threadIdx.x;
    Hit best;
    best.prim = kNoHit;
    best.tag = 0;
    best.th.t = ray_end_t;
    best.th.b0 = best.th.b1 = best.th.b2 = 0.0f;
    const V3 inv = splat(1.0f) / d;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t octant = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    const float slack = 1.0f + 2.0f * gamma_n(3.0f);
    const float kCullSlack = 1.0f + 0x1p-19f;
    const V3 sinv = mk3(slack * inv.x, slack * inv.y, slack * inv.z);
    const RayShear shear = ray_shear(d);
    uint32_t* pend = sm_bvh_stack + sc.wide_stack * kBlock + threadIdx.x;
    const float4* lds_nodes = reinterpret_cast<const float4*>(sm_bvh_stack + (sc.wide_stack + kLeafQueue) * kBlock);
    uint32_t sp = 0;
    uint32_t cur = 0;  // reference of the node or leaf to visit next (the root is wide node 0)
    bool have = true;
    for (;;) {
        uint32_t n_pend = 0;
        while (have && n_pend < kLeafQueue) {
            if (cur & 0x80000000u) {
                pend[n_pend * kBlock] = cur;
                n_pend++;
            } else {
                float4 lox, loy, loz, hix, hiy, hiz, refs_f, meta_f;
                if (STAGED) {
                    const float4* wn = lds_nodes + 8 * cur;
                    lox = wn[0], loy = wn[1], loz = wn[2], hix = wn[3], hiy = wn[4], hiz = wn[5], refs_f = wn[6], meta_f = wn[7];
                } else {
                    const float4* wn = sc.wnodes + 8 * (size_t)cur;
                    lox = wn[0], loy = wn[1], loz = wn[2], hix = wn[3], hiy = wn[4], hiz = wn[5], refs_f = wn[6], meta_f = wn[7];
                }
                const uint32_t refs[4] = {f2u(refs_f.x), f2u(refs_f.y), f2u(refs_f.z), f2u(refs_f.w)};
                const uint32_t table = octant < 4u ? f2u(meta_f.x) : f2u(meta_f.y);
                const uint32_t perm = (table >> (8u * (octant & 3u))) & 0xffu, n_children = f2u(meta_f.z);
                const float nearx[4] = {nx ? hix.x : lox.x, nx ? hix.y : lox.y, nx ? hix.z : lox.z, nx ? hix.w : lox.w};
                const float farx[4] = {nx ? lox.x : hix.x, nx ? lox.y : hix.y, nx ? lox.z : hix.z, nx ? lox.w : hix.w};
                const float neary[4] = {ny ? hiy.x : loy.x, ny ? hiy.y : loy.y, ny ? hiy.z : loy.z, ny ? hiy.w : loy.w};
                const float fary[4] = {ny ? loy.x : hiy.x, ny ? loy.y : hiy.y, ny ? loy.z : hiy.z, ny ? loy.w : hiy.w};
                const float nearz[4] = {nz ? hiz.x : loz.x, nz ? hiz.y : loz.y, nz ? hiz.z : loz.z, nz ? hiz.w : loz.w};
                const float farz[4] = {nz ? loz.x : hiz.x, nz ? loz.y : hiz.y, nz ? loz.z : hiz.z, nz ? loz.w : hiz.w};
                uint32_t hit_mask = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float tmin = 0.0f, tmax = ray_end_t, t_dom;
                    float t0 = inv.x * (nearx[c] - o.x);
                    float t1 = sinv.x * (farx[c] - o.x);
                    t_dom = t0;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    t0 = inv.y * (neary[c] - o.y);
                    t1 = sinv.y * (fary[c] - o.y);
                    t_dom = shear.kz == 1 ? t0 : t_dom;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    t0 = inv.z * (nearz[c] - o.z);
                    t1 = sinv.z * (farz[c] - o.z);
                    t_dom = shear.kz == 2 ? t0 : t_dom;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    if ((uint32_t)c < n_children && (0.0f <= tmin) && (tmin <= tmax) && !(t_dom > best.th.t * kCullSlack)) hit_mask |= 1u << c;
                }
                if (COUNT) (*n_nodes_visited) += n_children;
#pragma unroll
                for (int k = 3; k >= 0; k--) {  // last to visit first, so that the first is on top
                    const uint32_t c = (perm >> (2 * k)) & 3u;
                    if ((uint32_t)k < n_children && ((hit_mask >> c) & 1u) && sp < sc.wide_stack) {
                        stack[sp * kBlock] = c == 0u ? refs[0] : (c == 1u ? refs[1] : (c == 2u ? refs[2] : refs[3]));
                        sp++;
                    }
                }
            }
            if (sp > 0) cur = stack[(--sp) * kBlock];
            else have = false;
        }
        if (n_pend == 0) break;
        for (uint32_t j = 0; j < n_pend; j++) {
            const uint32_t ref = pend[j * kBlock];
            if (bvh_test_leaf<COUNT, ANY_HIT>(sc, o, shear, ray_end_t, ref & 0xffffffu, ((ref >> 24) & 0x7fu) + 1u, best, n_tris_tested)) return best;
        }
    }
    if (!(best.th.t != ray_end_t)) best.prim = kNoHit;  // bvh.rs:137
    if (!ANY_HIT && best.prim != kNoHit) best.tag = f2u(sc.tri_v[3 * best.prim + 2].w);
    return best;
}
template <bool COUNT, bool ANY_HIT, bool STAGED>
PT_D Hit bvh_intersect_binary(const SceneDev& sc, V3 o, V3 d, float ray_end_t, uint32_t* n_nodes_visited, uint32_t* n_tris_tested) {
    extern __shared__ uint32_t sm_bvh_stack[];
    uint32_t* stack = sm_bvh_stack + threadIdx.x;
    Hit best;
    best.prim = kNoHit;
    best.tag = 0;
    best.th.t = ray_end_t;
    best.th.b0 = best.th.b1 = best.th.b2 = 0.0f;
    const V3 inv = splat(1.0f) / d;
    const bool nx = inv.x < 0.0f, ny = inv.y < 0.0f, nz = inv.z < 0.0f;
    const uint32_t neg_mask = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
    const float slack = 1.0f + 2.0f * gamma_n(3.0f);
    const float kCullSlack = 1.0f + 0x1p-19f;  // 32 eps: see "Culling" above
    const V3 sinv = mk3(slack * inv.x, slack * inv.y, slack * inv.z);  // the reference's `slack * inv * (far - o)` associates this way
    const RayShear shear = ray_shear(d);
    uint32_t* pend = sm_bvh_stack + sc.bvh_stack * kBlock + threadIdx.x;  // leaves waiting for their triangle tests
    const float4* lds_nodes = reinterpret_cast<const float4*>(sm_bvh_stack + (sc.bvh_stack + 2 * kLeafQueue) * kBlock);
    uint32_t sp = 0;
    uint32_t ni = 0;
    bool have_node = true;
    for (;;) {
        uint32_t n_pend = 0;
        while (have_node && n_pend < kLeafQueue) {
            float4 n0, n1;
            if (STAGED) {
                const float4* np = lds_nodes + 2 * ni;
                n0 = np[0];
                n1 = np[1];
            } else {
                const float4* np = sc.nodes + 2 * (size_t)ni;
                n0 = np[0];
                n1 = np[1];
            }
            const uint32_t offset = f2u(n0.w), meta = f2u(n1.w);
            if (COUNT && PT_CENSUS_LEAD) (*n_nodes_visited)++;
            float tmin = 0.0f, tmax = ray_end_t, t_dom;
            {
                float near_x = nx ? n1.x : n0.x, far_x = nx ? n0.x : n1.x;
                float t0 = inv.x * (near_x - o.x);
                float t1 = sinv.x * (far_x - o.x);
                t_dom = t0;
                tmin = fmaxf(tmin, t0);
                tmax = fminf(tmax, t1);
                float near_y = ny ? n1.y : n0.y, far_y = ny ? n0.y : n1.y;
                t0 = inv.y * (near_y - o.y);
                t1 = sinv.y * (far_y - o.y);
                t_dom = shear.kz == 1 ? t0 : t_dom;
                tmin = fmaxf(tmin, t0);
                tmax = fminf(tmax, t1);
                float near_z = nz ? n1.z : n0.z, far_z = nz ? n0.z : n1.z;
                t0 = inv.z * (near_z - o.z);
                t1 = sinv.z * (far_z - o.z);
                t_dom = shear.kz == 2 ? t0 : t_dom;
                tmin = fmaxf(tmin, t0);
                tmax = fminf(tmax, t1);
            }
            if ((0.0f <= tmin) && (tmin <= tmax) && !(t_dom > best.th.t * kCullSlack)) {
                const uint32_t ntris = meta & 0xffffu;
                if (ntris == 0) {
                    const bool neg = ((neg_mask >> (meta >> 16)) & 1u) != 0u;
                    if (sp < sc.bvh_stack) {  // holds for every tree scene_create accepts
                        stack[sp * kBlock] = neg ? ni + 1 : offset;  // far child waits
                        sp++;
                        ni = neg ? offset : ni + 1;                  // near child is next
                        continue;
                    }
                } else {
                    pend[(2 * n_pend) * kBlock] = offset;
                    pend[(2 * n_pend + 1) * kBlock] = ntris;
                    n_pend++;
                }
            }
            if (sp > 0) ni = stack[(--sp) * kBlock];
            else have_node = false;
        }
        if (n_pend == 0) break;
        for (uint32_t j = 0; j < n_pend; j++) {
            const uint32_t leaf_first = pend[(2 * j) * kBlock], leaf_n = pend[(2 * j + 1) * kBlock];
            for (uint32_t k = leaf_first; k < leaf_first + leaf_n; k++) {
                const float4* tv = sc.tri_v + 3 * (size_t)k;  // one address, two immediate offsets
                float4 a = tv[0], b = tv[1], c = tv[2];
                // Left alone, the compiler re-loads overlapping halves of each vertex (x,y and y,z, for its packed-math
                // pairings): six vector-memory instructions per triangle instead of three, and vector-memory issue is
                // what bounds this loop. Passing the loaded values through an empty asm pins one 16-byte load each.
                // (Closest-hit walks only: the occlusion kernels carry more state and would drop to 7 waves per SIMD.)
                if (!ANY_HIT) asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(b.x), "+v"(b.y), "+v"(b.z), "+v"(c.x), "+v"(c.y), "+v"(c.z));
                if (COUNT && PT_CENSUS_LEAD) (*n_tris_tested)++;
                TriHit th;
                if (ray_triangle(o, shear, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), best.th.t, th)) {
                    best.th = th;
                    best.prim = k;
                    if (ANY_HIT && th.t < 0.999f * ray_end_t) return best;
                }
            }
        }
    }
    if (!(best.th.t != ray_end_t)) best.prim = kNoHit;  // bvh.rs:137
    // fetched once per ray here rather than carried through the walk (a register the loop does not have to spare)
    if (!ANY_HIT && best.prim != kNoHit) best.tag = f2u(sc.tri_v[3 * best.prim + 2].w);
    return best;
}
template <bool COUNT, bool ANY_HIT = false, bool WIDE = false>
PT_D Hit bvh_intersect(const SceneDev& sc, V3 o, V3 d, float ray_end_t, uint32_t* n_nodes_visited, uint32_t* n_tris_tested) {
    if (WIDE) {
        if (sc.lds_wnodes) return bvh_intersect_wide<COUNT, ANY_HIT, true>(sc, o, d, ray_end_t, n_nodes_visited, n_tris_tested);
        return bvh_intersect_wide<COUNT, ANY_HIT, false>(sc, o, d, ray_end_t, n_nodes_visited, n_tris_tested);
    }
    if (sc.lds_nodes) return bvh_intersect_binary<COUNT, ANY_HIT, true>(sc, o, d, ray_end_t, n_nodes_visited, n_tris_tested);
    return bvh_intersect_binary<COUNT, ANY_HIT, false>(sc, o, d, ray_end_t, n_nodes_visited, n_tris_tested);
}
// ------------------------------------------------------------------ the four-wide walk over a STREAM of rays
// bvh_intersect serves one ray per lane and returns when the slowest lane of the wave is done. In a large scene the
// rays of a wave (unrelated bounce rays after compaction) differ widely in length: veach-mis runs its walkers with
// 0.28 of the lanes active. Here a wave owns a sequence of rays (its 64-ray pieces of the queue, a grid stride apart),
// and a lane whose ray is finished hands in its result and takes the wave's next ray while the others walk on. Per ray
// nothing changes — the same nodes in the same order, the same triangle tests, the same culls as bvh_intersect_wide —,
// only which lane serves which ray does. Lanes are refilled between two rounds of the while-while loop, once
// kRefillAt of them are free (a refill runs the ray set-up, six IEEE divisions, for the whole wave) or none has work.
//   fetch(j, o, d) : ray j of the queue; emit(j, hit) : its result.
#ifndef PT_REFILL_AT
#define PT_REFILL_AT 16
#endif
constexpr uint32_t kRefillAt = PT_REFILL_AT;
template <bool COUNT, bool ANY_HIT, bool STAGED, class Fetch, class Emit>
PT_D void bvh_stream_wide(const SceneDev& sc, uint32_t n, const float ray_end_t, Fetch fetch, Emit emit, uint32_t* n_nodes_visited, uint32_t* n_tris_tested) {
    extern __shared__ uint32_t sm_bvh_stack[];
    uint32_t* stack = sm_bvh_stack + threadIdx.x;
    uint32_t* pend = sm_bvh_stack + sc.wide_stack * kBlock + threadIdx.x;
    const float4* lds_nodes = reinterpret_cast<const float4*>(sm_bvh_stack + (sc.wide_stack + kLeafQueue) * kBlock);
    const float slack = 1.0f + 2.0f * gamma_n(3.0f);
    const float kCullSlack = 1.0f + 0x1p-19f;
    const uint32_t lane = __lane_id();
    // this wave's rays: s = 0, 1, ... -> queue index base + (s / 64) * stride + s % 64
    const uint32_t base = blockIdx.x * kBlock + (threadIdx.x & ~63u), stride = gridDim.x * kBlock;
    const uint32_t total = base < n ? ((n - base + stride - 1u) / stride) * 64u : 0u;
    uint32_t next = 0;  // wave-uniform
    V3 o = mk3(0.0f, 0.0f, 0.0f), inv = o;
    RayShear shear{0, 0.0f, 0.0f, 0.0f};
    uint32_t octant = 0, sp = 0, cur = 0, j = 0;
    bool have_ray = false, have = false;
    Hit best;
    best.prim = kNoHit;
    best.tag = 0;
    best.th.t = ray_end_t;
    best.th.b0 = best.th.b1 = best.th.b2 = 0.0f;
    for (;;) {
        const bool done = have_ray && !have;  // this lane's ray is finished
        if (__any(done)) {  // emit is called by the whole wave (it may add to the film wave-wide), `done` marks its lanes
            if (done) {
                if (!(best.th.t != ray_end_t)) best.prim = kNoHit;  // bvh.rs:137
                if (!ANY_HIT && best.prim != kNoHit) best.tag = f2u(sc.tri_v[3 * best.prim + 2].w);
            }
            emit(done, j, best);
            if (done) have_ray = false;
        }
        const unsigned long long free_mask = __ballot(!have_ray);
        const uint32_t n_free = (uint32_t)__popcll(free_mask);
        if (next < total && (n_free >= kRefillAt || n_free == 64u)) {
            if (!have_ray) {
                const uint32_t s = next + (uint32_t)__popcll(free_mask & ((1ull << lane) - 1ull));
                const uint32_t jj = base + (s >> 6) * stride + (s & 63u);
                if (s < total && jj < n) {
                    V3 d;
                    fetch(jj, o, d);
                    j = jj;
                    inv = splat(1.0f) / d;
                    octant = (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u);
                    shear = ray_shear(d);
                    best.prim = kNoHit;
                    best.th.t = ray_end_t;
                    best.th.b0 = best.th.b1 = best.th.b2 = 0.0f;
                    sp = 0;
                    cur = 0;
                    have_ray = have = true;
                }
            }
            next += n_free;
        } else if (n_free == 64u)
            break;
        const bool nx = (octant & 1u) != 0u, ny = (octant & 2u) != 0u, nz = (octant & 4u) != 0u;
        const V3 sinv = mk3(slack * inv.x, slack * inv.y, slack * inv.z);
        uint32_t n_pend = 0;
        while (have && n_pend < kLeafQueue) {
            if (cur & 0x80000000u) {
                pend[n_pend * kBlock] = cur;
                n_pend++;
            } else {
                float4 lox, loy, loz, hix, hiy, hiz, refs_f, meta_f;
                if (STAGED) {
                    const float4* wn = lds_nodes + 8 * cur;
                    lox = wn[0], loy = wn[1], loz = wn[2], hix = wn[3], hiy = wn[4], hiz = wn[5], refs_f = wn[6], meta_f = wn[7];
                } else {
                    const float4* wn = sc.wnodes + 8 * (size_t)cur;
                    lox = wn[0], loy = wn[1], loz = wn[2], hix = wn[3], hiy = wn[4], hiz = wn[5], refs_f = wn[6], meta_f = wn[7];
                }
                const uint32_t refs[4] = {f2u(refs_f.x), f2u(refs_f.y), f2u(refs_f.z), f2u(refs_f.w)};
                const uint32_t table = octant < 4u ? f2u(meta_f.x) : f2u(meta_f.y);
                const uint32_t perm = (table >> (8u * (octant & 3u))) & 0xffu, n_children = f2u(meta_f.z);
                const float nearx[4] = {nx ? hix.x : lox.x, nx ? hix.y : lox.y, nx ? hix.z : lox.z, nx ? hix.w : lox.w};
                const float farx[4] = {nx ? lox.x : hix.x, nx ? lox.y : hix.y, nx ? lox.z : hix.z, nx ? lox.w : hix.w};
                const float neary[4] = {ny ? hiy.x : loy.x, ny ? hiy.y : loy.y, ny ? hiy.z : loy.z, ny ? hiy.w : loy.w};
                const float fary[4] = {ny ? loy.x : hiy.x, ny ? loy.y : hiy.y, ny ? loy.z : hiy.z, ny ? loy.w : hiy.w};
                const float nearz[4] = {nz ? hiz.x : loz.x, nz ? hiz.y : loz.y, nz ? hiz.z : loz.z, nz ? hiz.w : loz.w};
                const float farz[4] = {nz ? loz.x : hiz.x, nz ? loz.y : hiz.y, nz ? loz.z : hiz.z, nz ? loz.w : hiz.w};
                uint32_t hit_mask = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float tmin = 0.0f, tmax = ray_end_t, t_dom;
                    float t0 = inv.x * (nearx[c] - o.x);
                    float t1 = sinv.x * (farx[c] - o.x);
                    t_dom = t0;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    t0 = inv.y * (neary[c] - o.y);
                    t1 = sinv.y * (fary[c] - o.y);
                    t_dom = shear.kz == 1 ? t0 : t_dom;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    t0 = inv.z * (nearz[c] - o.z);
                    t1 = sinv.z * (farz[c] - o.z);
                    t_dom = shear.kz == 2 ? t0 : t_dom;
                    tmin = fmaxf(tmin, t0);
                    tmax = fminf(tmax, t1);
                    if ((uint32_t)c < n_children && (0.0f <= tmin) && (tmin <= tmax) && !(t_dom > best.th.t * kCullSlack)) hit_mask |= 1u << c;
                }
                if (COUNT) (*n_nodes_visited) += n_children;
#pragma unroll
                for (int k = 3; k >= 0; k--) {  // last to visit first, so that the first is on top
                    const uint32_t c = (perm >> (2 * k)) & 3u;
                    if ((uint32_t)k < n_children && ((hit_mask >> c) & 1u) && sp < sc.wide_stack) {
                        stack[sp * kBlock] = c == 0u ? refs[0] : (c == 1u ? refs[1] : (c == 2u ? refs[2] : refs[3]));
                        sp++;
                    }
                }
            }
            if (sp > 0) cur = stack[(--sp) * kBlock];
            else have = false;
        }
        bool decided = false;  // ANY_HIT: the occlusion query is answered (see bvh_intersect), the rest of the walk is dropped
        for (uint32_t q = 0; q < n_pend && !decided; q++) {
            const uint32_t ref = pend[q * kBlock];
            decided = bvh_test_leaf<COUNT, ANY_HIT>(sc, o, shear, ray_end_t, ref & 0xffffffu, ((ref >> 24) & 0x7fu) + 1u, best, n_tris_tested);
        }
        if (ANY_HIT && decided) have = false;
    }
}

// Scene::occluded (scene.rs:229-235): unnormalised segment, t_max 0.9999; only the boolean leaves this function.
template <bool COUNT = false, bool WIDE = false>
PT_D bool scene_occluded(const SceneDev& sc, V3 from, V3 to, uint32_t* n_nodes = nullptr, uint32_t* n_tris = nullptr) {
    Hit h = bvh_intersect<COUNT, true, WIDE>(sc, from, to - from, 0.9999f, n_nodes, n_tris);
    return h.prim != kNoHit;
}

// ------------------------------------------------------------------ shading-space helpers (pt/src/sg.rs)
PT_HD bool same_hemisphere(V3 a, V3 b) { return a.z * b.z > 0.0f; }
PT_HD float cos2theta(V3 v) { return v.z * v.z; }
PT_HD float sin2theta(V3 v) { return fmaxf(1.0f - cos2theta(v), 0.0f); }
PT_HD float tan2theta(V3 v) { return sin2theta(v) / cos2theta(v); }
PT_HD float cos_phi(V3 v) {
    float st = sqrtf(sin2theta(v));
    return st == 0.0f ? 1.0f : clampf(v.x / st, -1.0f, 1.0f);
}
PT_HD float sin_phi(V3 v) {
    float st = sqrtf(sin2theta(v));
    return st == 0.0f ? 0.0f : clampf(v.y / st, -1.0f, 1.0f);
}

// ------------------------------------------------------------------ BSDFs (pt/src/bsdfs.rs)
struct Cx {
    float re, im;
};
PT_HD Cx cx_mul(Cx a, Cx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
PT_HD Cx cx_add(Cx a, Cx b) { return {a.re + b.re, a.im + b.im}; }
PT_HD Cx cx_sub(Cx a, Cx b) { return {a.re - b.re, a.im - b.im}; }
PT_HD float cx_norm2(Cx a) { return a.re * a.re + a.im * a.im; }
PT_HD Cx cx_div(Cx a, Cx b) {
    float n = cx_norm2(b);
    return {(a.re * b.re + a.im * b.im) / n, (a.im * b.re - a.re * b.im) / n};
}
PT_D Cx cx_sqrt(Cx a) {  // num-complex 0.4 Complex::sqrt (polar form with the axis special cases)
    if (a.im == 0.0f) {
        if (!signbit(a.re)) return {sqrtf(a.re), a.im};
        float im = sqrtf(-a.re);
        return {0.0f, signbit(a.im) ? -im : im};
    } else if (a.re == 0.0f) {
        float x = sqrtf(fabsf(a.im) / 2.0f);
        return {x, signbit(a.im) ? -x : x};
    }
    float r = hypotf(a.re, a.im);
    float th = atan2f(a.im, a.re) / 2.0f;
    float sr = sqrtf(r), s, c;
    sincosf(th, &s, &c);
    return {sr * c, sr * s};
}
PT_HD float fresnel_dielectric(float cos_i, float eta) {  // bsdfs.rs:316-331
    cos_i = clampf(cos_i, -1.0f, 1.0f);
    if (cos_i < 0.0f) {
        eta = 1.0f / eta;
        cos_i = -cos_i;
    }
    float s2i = 1.0f - cos_i * cos_i;
    float s2t = s2i / (eta * eta);
    if (s2t >= 1.0f) return 1.0f;
    float cos_t = safe_sqrtf(1.0f - s2t);
    float r_parl = (eta * cos_i - cos_t) / (eta * cos_i + cos_t);
    float r_perp = (cos_i - eta * cos_t) / (cos_i + eta * cos_t);
    return (r_parl * r_parl + r_perp * r_perp) / 2.0f;
}
// Not inlined: eight wavelengths times several call sites of this (complex sqrt inside) made the all-materials shade
// kernels 100-180 KB, far beyond the 64 KB instruction cache two CUs share.
__device__ __attribute__((noinline)) float fresnel_complex(float cos_i, Cx eta) {  // bsdfs.rs:333-343
    cos_i = clampf(cos_i, 0.0f, 1.0f);
    float s2i = 1.0f - cos_i * cos_i;
    Cx s2t = cx_div(Cx{s2i, 0.0f}, cx_mul(eta, eta));
    Cx cos_t = cx_sqrt(cx_sub(Cx{1.0f, 0.0f}, s2t));
    Cx ci{cos_i, 0.0f};
    Cx r_parl = cx_div(cx_sub(cx_mul(eta, ci), cos_t), cx_add(cx_mul(eta, ci), cos_t));
    Cx r_perp = cx_div(cx_sub(ci, cx_mul(eta, cos_t)), cx_add(ci, cx_mul(eta, cos_t)));
    return (cx_norm2(r_parl) + cx_norm2(r_perp)) / 2.0f;
}
PT_D Spec fresnel_complex_spec(float cos_i, const Spec& eta, const Spec& k) {
    Spec r;
#pragma unroll
    for (int i = 0; i < NSW; i++) r.v[i] = fresnel_complex(cos_i, Cx{eta.v[i], k.v[i]});
    return r;
}
PT_HD bool ggx_smooth(float a) { return a < 0.0001f; }
PT_D V3 ggx_sample_wm(float a, V3 wo, float u0, float u1) {  // bsdfs.rs:359-369
    V3 wi = wo.z < 0.0f ? -wo : wo;
    V3 ws = mk3(wi.x * a, wi.y * a, wi.z);
    float phi = (2.0f * u0 - 1.0f) * kPi;
    float z = fmaf(1.0f - u1, 1.0f + ws.z, -ws.z);
    float st = safe_sqrtf(1.0f - z * z);
    float s, c;
    sincosf(phi, &s, &c);
    V3 wm = ws + mk3(st * c, st * s, z);
    return normalize(mk3(wm.x * a, wm.y * a, wm.z));
}
PT_HD float ggx_lambda(float a, V3 w) {  // bsdfs.rs:391-397
    float t2 = tan2theta(w);
    if (isinf(t2)) return 0.0f;
    return (sqrtf(1.0f + a * a * t2) - 1.0f) / 2.0f;
}
PT_HD float ggx_g1(float a, V3 w) { return 1.0f / (1.0f + ggx_lambda(a, w)); }
PT_HD float ggx_g(float a, V3 wo, V3 wi) { return 1.0f / (1.0f + ggx_lambda(a, wo) + ggx_lambda(a, wi)); }
PT_HD float ggx_d(float a, V3 wm) {  // bsdfs.rs:370-378
    float t2 = tan2theta(wm);
    if (isinf(t2)) return 0.0f;
    float c2 = cos2theta(wm);
    float c4 = c2 * c2;
    float cpa = cos_phi(wm) / a, spa = sin_phi(wm) / a;
    float e = t2 * (cpa * cpa + spa * spa);
    float ope = 1.0f + e;
    return 1.0f / (kPi * a * a * c4 * (ope * ope));
}
PT_HD float ggx_d_visible(float a, V3 w, V3 wm) { return ggx_g1(a, w) / fabsf(w.z) * ggx_d(a, wm) * fabsf(dot(w, wm)); }  // bsdfs.rs:379-384
PT_HD V3 reflect_about(V3 w, V3 n) { return -w + 2.0f * dot(w, n) * n; }
PT_HD float shading_normal_correction(V3 wo, V3 wi, V3 ng, V3 ns) {  // bsdfs.rs:130-134
    float a = dot(wo, ns) * dot(wi, ng);
    float b = dot(wo, ng) * dot(wi, ns);
    return b == 0.0f ? 0.0f : fabsf(a / b);
}

struct Bsdf {  // scene::compute_bsdf result (scene.rs:236-256)
    int kind;
    float roughness;
    Spec a, b;  // albedo | eta ; conductor k
};
struct BsdfSample {
    Spec f;
    V3 wi;
    float pdf;
    bool specular, terminate_secondary;
};
PT_HD bool bsdf_is_specular(const Bsdf& m) { return m.kind == MAT_DIELECTRIC || (m.kind == MAT_CONDUCTOR && ggx_smooth(m.roughness)); }

PT_D Spec conductor_f(const Bsdf& m, V3 wo, V3 wi, V3 wm) {
    float co = fabsf(wo.z), ci = fabsf(wi.z);
    return fabsf(wi.z) * fresnel_complex_spec(fabsf(dot(wo, wm)), m.a, m.b) * ggx_d(m.roughness, wm) * ggx_g(m.roughness, wo, wi) / (4.0f * co * ci);
}
// BSDF::sample in shading space (bsdfs.rs:136-156,172-205,220-259). `radiance` = TransportMode::Radiance.
PT_D bool bsdf_sample_local(const Bsdf& m, V3 wo, float u1d, float u0, float u1, bool radiance, BsdfSample& s) {
    if (m.kind == MAT_LAMBERTIAN) {
        V3 wi = cosine_hemisphere(u0, u1);
        if (!same_hemisphere(wo, wi)) wi = -wi;
        float pdf = cosine_hemisphere_pdf(fabsf(wi.z));
        if (pdf == 0.0f) return false;
        s.f = m.a * fabsf(wi.z) * kInvPi;
        s.wi = wi;
        s.pdf = pdf;
        s.specular = false;
        s.terminate_secondary = false;
        return true;
    }
    if (m.kind == MAT_DIELECTRIC) {
        float eta = m.a.v[0];
        float R = fresnel_dielectric(wo.z, eta);
        if (u1d < R) {
            s.wi = mk3(-wo.x, -wo.y, wo.z);
            s.f = spec_splat(R);
            s.pdf = R;
            s.specular = true;
            s.terminate_secondary = false;
            return true;
        }
        // refract(wo, +z, eta) (bsdfs.rs:293-314)
        float cos_i = wo.z;
        V3 n = mk3(0.0f, 0.0f, 1.0f);
        if (cos_i < 0.0f) {
            eta = 1.0f / eta;
            cos_i = -cos_i;
            n = -n;
        }
        float s2i = fmaxf(0.0f, 1.0f - cos_i * cos_i);
        float s2t = s2i / (eta * eta);
        if (!(s2t < 1.0f)) return false;  // unreachable in the reference (asserts); path ends here
        float cos_t = safe_sqrtf(1.0f - s2t);
        V3 wt = -wo / eta + (cos_i / eta - cos_t) * n;
        float f = 1.0f - R;
        if (radiance) f /= eta * eta;
        s.f = spec_splat(f);
        s.wi = wt;
        s.pdf = 1.0f - R;
        s.specular = true;
        s.terminate_secondary = !spec_is_uniform(m.a);
        return true;
    }
    float al = m.roughness;
    if (ggx_smooth(al)) {
        V3 wi = mk3(-wo.x, -wo.y, wo.z);
        s.f = fresnel_complex_spec(wi.z, m.a, m.b);
        s.wi = wi;
        s.pdf = 1.0f;
        s.specular = true;
        s.terminate_secondary = false;
        return true;
    }
    V3 wm = ggx_sample_wm(al, wo, u0, u1);
    V3 wi = reflect_about(wo, wm);
    if (!same_hemisphere(wo, wi)) return false;
    s.pdf = ggx_d_visible(al, wo, wm) / (4.0f * fabsf(dot(wo, wm)));
    s.f = conductor_f(m, wo, wi, wm);
    s.wi = wi;
    s.specular = false;
    s.terminate_secondary = false;
    return true;
}
PT_D Spec bsdf_eval_local(const Bsdf& m, V3 wo, V3 wi) {  // bsdfs.rs:157-162,207-209,261-279
    if (m.kind == MAT_LAMBERTIAN) {
        if (!same_hemisphere(wo, wi)) return spec_splat(0.0f);
        return m.a * fabsf(wi.z) * kInvPi;
    }
    if (m.kind == MAT_DIELECTRIC) return spec_splat(0.0f);
    if (ggx_smooth(m.roughness) || !same_hemisphere(wo, wi) || len2(wo + wi) == 0.0f) return spec_splat(0.0f);
    V3 wm = normalize(wo + wi);
    return conductor_f(m, wo, wi, wm);
}
PT_D float bsdf_pdf_local(const Bsdf& m, V3 wo, V3 wi) {  // bsdfs.rs:164-170,211-213,281-287
    if (m.kind == MAT_LAMBERTIAN) return same_hemisphere(wo, wi) ? cosine_hemisphere_pdf(fabsf(wi.z)) : 0.0f;
    if (m.kind == MAT_DIELECTRIC) return 0.0f;
    if (ggx_smooth(m.roughness) || !same_hemisphere(wo, wi) || len2(wo + wi) == 0.0f) return 0.0f;
    V3 wm = normalize(wo + wi);
    return ggx_d_visible(m.roughness, wo, wm) / (4.0f * fabsf(dot(wo, wm)));
}
// World-space wrappers (bsdfs.rs:66-127). NB eval passes (ns, ng) where sample passes (ng, ns): kept.
PT_D bool bsdf_sample(const Bsdf& m, const Frame& fr, V3 ng, V3 wo, float u1d, float u0, float u1, bool radiance, BsdfSample& s) {
    if (!bsdf_sample_local(m, to_local(fr, wo), u1d, u0, u1, radiance, s)) return false;
    V3 wi = to_world(fr, s.wi);
    if (!radiance) s.f = s.f * shading_normal_correction(wo, wi, ng, fr.z);
    s.wi = wi;
    return true;
}
PT_D Spec bsdf_eval(const Bsdf& m, const Frame& fr, V3 ng, V3 wo, V3 wi, bool radiance) {
    Spec f = bsdf_eval_local(m, to_local(fr, wo), to_local(fr, wi));
    if (!radiance) f = f * shading_normal_correction(wo, wi, fr.z, ng);
    return f;
}
PT_D float bsdf_pdf(const Bsdf& m, const Frame& fr, V3 wo, V3 wi) { return bsdf_pdf_local(m, to_local(fr, wo), to_local(fr, wi)); }

// ------------------------------------------------------------------ scene queries (pt/src/scene.rs)
struct Tri {
    V3 p0, p1, p2;
    uint32_t mesh;
    uint32_t tag;  // material + 1 (0 = pass-through) | kind << 28 | has_normals << 31 (pt_types.h)
};
PT_HD int tri_material(const Tri& t) { return tag_material(t.tag); }
PT_HD bool tri_has_normals(const Tri& t) { return (t.tag & kTagNormalsBit) != 0; }
PT_D Tri load_tri_from(const float4* tri_v, uint32_t k) {  // tri_v: the scene's array, or a copy of it in LDS (small scenes)
    const float4 a = tri_v[3 * k], b = tri_v[3 * k + 1], c = tri_v[3 * k + 2];
    return Tri{mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), f2u(a.w), f2u(c.w)};
}
PT_D Tri load_tri(const SceneDev& sc, uint32_t k) { return load_tri_from(sc.tri_v, k); }
struct Surface {  // SceneInteraction (scene.rs:65-86,192-217)
    V3 blo, bhi;   // error box around p
    V3 ng;
    Frame frame;   // z = ns
    Tri tri;
};
PT_HD V3 surf_p(const Surface& s) { return (s.blo + s.bhi) / 2.0f; }
PT_D Surface make_surface_from(const SceneDev& sc, const float4* tri_v, uint32_t prim, float b0, float b1, float b2) {
    Surface s;
    s.tri = load_tri_from(tri_v, prim);
    const float b[3] = {b0, b1, b2};
    V3 p = tri_point(s.tri.p0, s.tri.p1, s.tri.p2, b);
    V3 abs_sum = absv(s.tri.p0) * b0 + absv(s.tri.p1) * b1 + absv(s.tri.p2) * b2;
    V3 err = gamma_n(7.0f) * abs_sum;
    s.ng = tri_normal(s.tri.p0, s.tri.p1, s.tri.p2);
    V3 ns = s.ng;
    if (tri_has_normals(s.tri)) {
        const float4 n0 = sc.tri_n[3 * prim], n1 = sc.tri_n[3 * prim + 1], n2 = sc.tri_n[3 * prim + 2];
        ns = normalize((mk3(n0.x, n0.y, n0.z) * b0) + (mk3(n1.x, n1.y, n1.z) * b1) + (mk3(n2.x, n2.y, n2.z) * b2));
    }
    s.frame = frame_from_z(ns);
    s.blo = p - err;
    s.bhi = p + err;
    return s;
}
PT_D Surface make_surface(const SceneDev& sc, uint32_t prim, float b0, float b1, float b2) { return make_surface_from(sc, sc.tri_v, prim, b0, b1, b2); }
// KINDS = bit mask of the material kinds the scene uses (1 << MAT_*), fixed at scene creation. The shade kernels are
// compiled per mask so that a scene without, say, conductors does not carry their registers and branches.
// (spec_lookup: see SpecTab in pt_types.h.)
PT_D Spec spec_lookup(const SpecTab& st, uint32_t spectrum, uint32_t row) {
    const float4* e = st.tab + 2 * ((size_t)spectrum * st.n_rows + row);
    const float4 a = e[0], b = e[1];
    Spec s;
    s.v[0] = a.x; s.v[1] = a.y; s.v[2] = a.z; s.v[3] = a.w;
    s.v[4] = b.x; s.v[5] = b.y; s.v[6] = b.z; s.v[7] = b.w;
    return s;
}
struct MatHead {  // what shading needs of a MaterialDev beyond the spectral table
    int kind;
    float roughness;
};
template <int KINDS>
PT_D Bsdf make_bsdf_head(const SpecTab& st, MatHead m, int material, uint32_t row) {
    Bsdf b;
    b.kind = m.kind;
    if (KINDS == (1 << MAT_LAMBERTIAN)) b.kind = MAT_LAMBERTIAN;
    else if (KINDS == (1 << MAT_DIELECTRIC)) b.kind = MAT_DIELECTRIC;
    else if (KINDS == (1 << MAT_CONDUCTOR)) b.kind = MAT_CONDUCTOR;
    else if (!(KINDS & (1 << MAT_CONDUCTOR))) b.kind = m.kind == MAT_LAMBERTIAN ? MAT_LAMBERTIAN : MAT_DIELECTRIC;
    else if (!(KINDS & (1 << MAT_DIELECTRIC))) b.kind = m.kind == MAT_LAMBERTIAN ? MAT_LAMBERTIAN : MAT_CONDUCTOR;
    else if (!(KINDS & (1 << MAT_LAMBERTIAN))) b.kind = m.kind == MAT_DIELECTRIC ? MAT_DIELECTRIC : MAT_CONDUCTOR;
    b.roughness = m.roughness;
    b.a = spec_lookup(st, 2u * (uint32_t)material, row);
    b.b = (b.kind == MAT_CONDUCTOR) ? spec_lookup(st, 2u * (uint32_t)material + 1u, row) : spec_splat(0.0f);
    return b;
}
template <int KINDS>
PT_D Bsdf make_bsdf(const SceneDev& sc, const SpecTab& st, int material, uint32_t row) {
    return make_bsdf_head<KINDS>(st, MatHead{sc.materials[material].kind, sc.materials[material].roughness}, material, row);
}

// ------------------------------------------------------------------ area lights (pt/src/lighting.rs:12-169)
// AreaLight::emitted: one-sided along the geometric normal.
PT_D Spec light_emitted(const SpecTab& st, const Tri& t, V3 n, V3 from, V3 to, uint32_t row) {
    if (dot(to - from, n) <= 0.0f) return spec_splat(0.0f);
    return spec_lookup(st, st.emission_base + t.mesh, row);
}
PT_HD float light_pdf_li(const Tri& t, V3 n, V3 from, V3 to) {  // pdf_li_solid_angle
    float area_pdf = 1.0f / tri_area(t.p0, t.p1, t.p2);
    V3 wl = to - from;
    float d2 = len2(wl);
    wl = normalize(wl);
    float c = fabsf(dot(wl, n));
    if (c == 0.0f) return 0.0f;
    return area_pdf * d2 / c;
}
struct LightPoint {  // common part of LiSample / LeSample
    V3 blo, bhi, n;
    Tri tri;
    float pmf;
};
// Scene::choose_light (scene.rs:257-268): uniform over emissive triangles.
PT_D bool choose_light_from(const SceneDev& sc, const float4* light_v, float u, Tri& t, float& pmf) {
    if (sc.n_emissive_tris == 0) return false;
    uint32_t idx = (uint32_t)((float)sc.n_emissive_tris * u);
    if (idx >= sc.n_emissive_tris) idx = sc.n_emissive_tris - 1;
    t = load_tri_from(light_v, idx);
    pmf = 1.0f / (float)sc.n_emissive_tris;
    return true;
}
PT_D bool choose_light(const SceneDev& sc, float u, Tri& t, float& pmf) { return choose_light_from(sc, sc.light_v, u, t, pmf); }
struct LiSample {
    V3 blo, bhi, n, wl;
    float pdf;
    Tri tri;
};
PT_D bool light_sample_li(const Tri& t, V3 from, float u0, float u1, LiSample& s) {  // sample_li_solid_angle
    float b[3];
    uniform_barycentric(u0, u1, b);
    V3 p = tri_point(t.p0, t.p1, t.p2, b);
    V3 n = tri_normal(t.p0, t.p1, t.p2);
    V3 wl = p - from;
    if (dot(n, wl) >= 0.0f) return false;
    float d2 = len2(wl);
    wl = normalize(wl);
    float area_pdf = 1.0f / tri_area(t.p0, t.p1, t.p2);
    float c = fabsf(dot(wl, n));
    float pdf = area_pdf * d2 / c;
    if (pdf == 0.0f) return false;
    V3 abs_sum = absv(t.p0) * b[0] + absv(t.p1) * b[1] + absv(t.p2) * b[2];
    V3 err = gamma_n(6.0f) * abs_sum;
    s.blo = p - err;
    s.bhi = p + err;
    s.n = n;
    s.wl = wl;
    s.pdf = pdf;
    s.tri = t;
    return true;
}
struct LeSample {
    V3 blo, bhi, n, dir;
    float pdf_dir, pdf_pos;
};
PT_D bool light_sample_le(const Tri& t, float u00, float u01, float u10, float u11, LeSample& s) {  // sample_le
    float b[3];
    uniform_barycentric(u00, u01, b);
    V3 p = tri_point(t.p0, t.p1, t.p2, b);
    float pdf_pos = 1.0f / tri_area(t.p0, t.p1, t.p2);
    V3 dir = cosine_hemisphere(u10, u11);
    float pdf_dir = cosine_hemisphere_pdf(dir.z);
    if (pdf_dir == 0.0f) return false;
    V3 abs_sum = absv(t.p0) * b[0] + absv(t.p1) * b[1] + absv(t.p2) * b[2];
    V3 err = gamma_n(6.0f) * abs_sum;
    V3 n = tri_normal(t.p0, t.p1, t.p2);
    Frame fr = frame_from_z(n);
    s.dir = to_world(fr, dir);
    s.blo = p - err;
    s.bhi = p + err;
    s.n = n;
    s.pdf_dir = pdf_dir;
    s.pdf_pos = pdf_pos;
    return true;
}

// ------------------------------------------------------------------ camera (pt/src/camera.rs:32-114)
PT_D void camera_sample_we(const CameraDev& cam, uint32_t ix, uint32_t iy, float u0, float u1, V3& o, V3& d) {
    float x = ((float)cam.resx * 0.5f - (float)ix - u0) * cam.pixel_size;
    float y = ((float)cam.resy * 0.5f - (float)iy - u1) * cam.pixel_size;
    o = mat_point(cam.c2w, mk3(0.0f, 0.0f, 0.0f));
    d = normalize(mat_vector(cam.c2w, mk3(x, y, 1.0f)));
}
struct CameraWi {
    float importance, pdf, rx, ry;
    V3 wc;
};
PT_D bool camera_sample_wi(const CameraDev& cam, V3 from, CameraWi& s) {
    V3 to_cam = cam.p - from;
    V3 wc = normalize(to_cam);
    V3 dir = mat_vector(cam.w2c, -wc);
    if (dir.z <= 0.0f) return false;
    V3 dn = dir / dir.z;
    float W = (float)cam.resx, H = (float)cam.resy;
    float x = W / 2.0f - dn.x / cam.pixel_size;
    float y = H / 2.0f - dn.y / cam.pixel_size;
    if (!((0.0f <= x && x <= W) && (0.0f <= y && y <= H))) return false;
    float d2 = len2(to_cam);
    float c = fabsf(dot(cam.n, wc));
    float cz = dir.z;
    float c4 = (cz * cz) * (cz * cz);
    s.importance = 1.0f / (cam.pixel_size * cam.pixel_size * W * H * c4);
    s.wc = wc;
    s.pdf = d2 / c;
    s.rx = x;
    s.ry = y;
    return true;
}
PT_HD uint32_t splat_pixel_index(const CameraDev& cam, float rx, float ry) {  // Film::add_splat (camera.rs:161-166)
    uint32_t ix = rx > 0.0f ? (uint32_t)rx : 0u, iy = ry > 0.0f ? (uint32_t)ry : 0u;
    if (ix > cam.resx - 1) ix = cam.resx - 1;
    if (iy > cam.resy - 1) iy = cam.resy - 1;
    return ix + iy * cam.resx;
}

}  // namespace pt
