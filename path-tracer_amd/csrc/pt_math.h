// pt_math.h — vector / bounds / frame / matrix arithmetic shared by the host-side scene build and the
// gfx950 kernels of libptgpu. f32 everywhere, dot products left-to-right without fma, explicit fmaf only
// where the reference uses mul_add (math/src/vector.rs:126-207,1055-1061, bounds.rs:25-132, onb.rs:12-30,
// matrix.rs:23-49,109-178, numerics.rs:5-9). The whole library is compiled with -ffp-contract=off so the
// compiler never fuses behind our back: hit records must be bit-identical to the CPU restatement.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#define PT_HD __host__ __device__ __forceinline__
#define PT_D __device__ __forceinline__

namespace pt {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 2.0f * kPi;
constexpr float kFourPi = 4.0f * kPi;
constexpr float kInvPi = 1.0f / kPi;
constexpr float kInvTwoPi = 1.0f / kTwoPi;
constexpr float kInvFourPi = 1.0f / kFourPi;
constexpr float kEps = 1.1920929e-07f;      // f32::EPSILON
constexpr float kOneMinusEps = 0x1.fffffep-1f;  // 1 - EPSILON/2

PT_HD float f_inf() { return __builtin_huge_valf(); }
PT_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
PT_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

// numeric_gamma(n) = n*eps/(1 - n*eps), eps = 2^-24 (numerics.rs:5-9)
PT_HD constexpr float gamma_n(float n) { return (n * (kEps / 2.0f)) / (1.0f - n * (kEps / 2.0f)); }

PT_HD float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
PT_HD float safe_sqrtf(float x) { return sqrtf(fmaxf(0.0f, x)); }

// f32::next_up / next_down on finite non-NaN inputs (+-0 -> smallest subnormal of the right sign).
PT_HD float next_up(float x) {
    if (x != x || x == f_inf()) return x;
    uint32_t b = f2u(x), a = b & 0x7fffffffu;
    uint32_t nb = (a == 0) ? 1u : ((b == a) ? b + 1 : b - 1);
    return u2f(nb);
}
PT_HD float next_down(float x) {
    if (x != x || x == -f_inf()) return x;
    uint32_t b = f2u(x), a = b & 0x7fffffffu;
    uint32_t nb = (a == 0) ? 0x80000001u : ((b == a) ? b - 1 : b + 1);
    return u2f(nb);
}

struct V3 {
    float x, y, z;
};
PT_HD V3 mk3(float x, float y, float z) { return V3{x, y, z}; }
PT_HD V3 splat(float v) { return V3{v, v, v}; }
PT_HD V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
PT_HD V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
PT_HD V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }
PT_HD V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
PT_HD V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_HD float get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
PT_HD void set(V3& a, int i, float v) {
    if (i == 0) a.x = v;
    else if (i == 1) a.y = v;
    else a.z = v;
}
PT_HD V3 absv(V3 a) { return {fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
PT_HD float len2(V3 a) { return dot(a, a); }
PT_HD float len(V3 a) { return sqrtf(dot(a, a)); }
PT_HD V3 normalize(V3 a) { return a * (1.0f / len(a)); }
PT_HD V3 min3(V3 a, V3 b) { return {fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
PT_HD V3 max3(V3 a, V3 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
PT_HD V3 clamp3(V3 a, V3 lo, V3 hi) { return {clampf(a.x, lo.x, hi.x), clampf(a.y, lo.y, hi.y), clampf(a.z, lo.z, hi.z)}; }
// vector.rs:154-165 — the running maximum is NOT updated by the y test's winner when comparing z (sic).
PT_HD int max_axis(V3 a) {
    int axis = 0;
    float m = a.x;
    if (a.y > m) {
        m = a.y;
        axis = 1;
    }
    if (a.z > m) axis = 2;
    return axis;
}
PT_HD float max_comp(V3 a) { return fmaxf(fmaxf(a.x, a.y), a.z); }

struct V2 {
    float x, y;
};
PT_HD V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
PT_HD V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
PT_HD V2 operator*(V2 a, V2 b) { return {a.x * b.x, a.y * b.y}; }
PT_HD V2 operator/(V2 a, float s) { return {a.x / s, a.y / s}; }
PT_HD V2 min2(V2 a, V2 b) { return {fminf(a.x, b.x), fminf(a.y, b.y)}; }
PT_HD V2 max2(V2 a, V2 b) { return {fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
PT_HD V2 clamp2(V2 a, V2 lo, V2 hi) { return {clampf(a.x, lo.x, hi.x), clampf(a.y, lo.y, hi.y)}; }

struct Box3 {
    V3 lo, hi;
};
PT_HD Box3 box_empty() { return {splat(f_inf()), splat(-f_inf())}; }
PT_HD Box3 box_join(Box3 a, Box3 b) { return {min3(a.lo, b.lo), max3(a.hi, b.hi)}; }
PT_HD Box3 box_join(Box3 a, V3 p) { return {min3(a.lo, p), max3(a.hi, p)}; }
PT_HD V3 box_centroid(Box3 b) { return (b.lo + b.hi) / 2.0f; }
PT_HD V3 box_extent(Box3 b) { return b.hi - b.lo; }
PT_HD float box_area(Box3 b) {
    V3 e = box_extent(b);
    return 2.0f * (e.x * e.y + e.y * e.z + e.x * e.z);
}
PT_HD Box3 box_around(V3 c, V3 extent) {
    V3 h = extent / 2.0f;
    return {c - h, c + h};
}

struct Box2 {
    V2 lo, hi;
};
PT_HD Box2 box2_unit() { return {{0.0f, 0.0f}, {1.0f, 1.0f}}; }
PT_HD V2 box2_centroid(Box2 b) { return (b.lo + b.hi) / 2.0f; }
PT_HD V2 box2_extent(Box2 b) { return b.hi - b.lo; }
PT_HD float box2_area(Box2 b) {
    V2 e = box2_extent(b);
    return e.x * e.y;
}
PT_HD Box2 box2_around(V2 c, V2 extent) {
    V2 h = extent / 2.0f;
    return {c - h, c + h};
}
PT_HD float box2_overlap(Box2 a, Box2 b) {
    V2 lo = max2(a.lo, b.lo), hi = min2(a.hi, b.hi);
    V2 e = hi - lo;
    if (e.x <= 0.0f || e.y <= 0.0f) return 0.0f;
    return e.x * e.y;
}

// Orthonormal frame from its z axis (onb.rs:12-24; tangents swapped w.r.t. Duff et al.).
struct Frame {
    V3 x, y, z;
};
PT_HD Frame frame_from_z(V3 z) {
    float s = copysignf(1.0f, z.z);
    float a = -1.0f / (s + z.z);
    float b = z.x * z.y * a;
    Frame f;
    f.x = mk3(b, s + z.y * z.y * a, -z.y);
    f.y = mk3(1.0f + s * z.x * z.x * a, s * b, -s * z.x);
    f.z = z;
    return f;
}
PT_HD V3 to_local(const Frame& f, V3 v) { return mk3(dot(f.x, v), dot(f.y, v), dot(f.z, v)); }
PT_HD V3 to_world(const Frame& f, V3 v) { return f.x * v.x + f.y * v.y + f.z * v.z; }

// 4x4, column-major like the reference (c[col][row]).
struct Mat4 {
    float c[4][4];
};
PT_HD V3 mat_point(const Mat4& m, V3 p) {
    return mk3(m.c[0][0] * p.x + m.c[1][0] * p.y + m.c[2][0] * p.z + m.c[3][0] * 1.0f,
               m.c[0][1] * p.x + m.c[1][1] * p.y + m.c[2][1] * p.z + m.c[3][1] * 1.0f,
               m.c[0][2] * p.x + m.c[1][2] * p.y + m.c[2][2] * p.z + m.c[3][2] * 1.0f);
}
PT_HD V3 mat_vector(const Mat4& m, V3 v) {
    return mk3(m.c[0][0] * v.x + m.c[1][0] * v.y + m.c[2][0] * v.z + m.c[3][0] * 0.0f,
               m.c[0][1] * v.x + m.c[1][1] * v.y + m.c[2][1] * v.z + m.c[3][1] * 0.0f,
               m.c[0][2] * v.x + m.c[1][2] * v.y + m.c[2][2] * v.z + m.c[3][2] * 0.0f);
}

}  // namespace pt
