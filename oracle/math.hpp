// ORACLE — TEST INFRASTRUCTURE ONLY. CPU restatement of the reference's `math` crate subset used on
// the hot path. Nothing under oracle/ may be used by the product (path-tracer_amd/): only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg call it, as the checker.
//
// Follows (operation order, f32 everywhere, explicit fmaf only where the reference uses mul_add):
//   math/src/vector.rs:126-207,1055-1061  (abs/clamp/max_axis/min,max_by_component/cross/dot/mag/normalized)
//   math/src/bounds.rs:25-132             (Bounds2/3: around, centroid, clip, clamp, extent, area, volume,
//                                          surface_area, offset, lerp, overlap, | )
//   math/src/matrix.rs:23-49,109-178,294-308 (mat4 * point/vector, adjugate, inverse, transposed)
//   math/src/onb.rs:12-30                 (ONB3f32::init_z / apply_vector / apply_vector_inverse)
//   math/src/numerics.rs:5-9              (numeric_gamma)    math/src/lib.rs:15-40 (constants, safe_sqrt)
// Build with -ffp-contract=off so that a*b+c is never fused behind our back.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

namespace orc {

static constexpr float PI = 3.14159265358979323846264338327950288f;  // f32::consts::PI
static constexpr float TWO_PI = 2.0f * PI;
static constexpr float FOUR_PI = 4.0f * PI;
static constexpr float INV_PI = 1.0f / PI;
static constexpr float INV_TWO_PI = 1.0f / TWO_PI;
static constexpr float INV_FOUR_PI = 1.0f / FOUR_PI;
static constexpr float F32_EPSILON = 1.1920929e-07f;  // f32::EPSILON = 2^-23
static constexpr float F32_INF = std::numeric_limits<float>::infinity();

// Rust f32::max / f32::min: if one operand is NaN the other is returned (== fmaxf / fminf).
inline float rmax(float a, float b) { return std::fmax(a, b); }
inline float rmin(float a, float b) { return std::fmin(a, b); }
// Rust f32::clamp(min,max): NaN stays NaN; otherwise `if x < min {min} else if x > max {max} else {x}`.
inline float rclamp(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
inline float safe_sqrt(float x) { return std::sqrt(rmax(0.0f, x)); }  // lib.rs:36-40
// numerics.rs:5-9
inline float numeric_gamma(float n) {
    const float EPS = F32_EPSILON / 2.0f;
    return (n * EPS) / (1.0f - n * EPS);
}
// Rust `x as usize` for f32: saturating, NaN -> 0.
inline size_t f32_as_usize(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 18446744073709551616.0f) return ~size_t(0);
    return (size_t)x;
}
inline int32_t f32_as_i32(float x) {
    if (x != x) return 0;
    if (x <= -2147483648.0f) return INT32_MIN;
    if (x >= 2147483648.0f) return INT32_MAX;
    return (int32_t)x;
}
inline float next_up(float x) {  // f32::next_up
    if (x != x || x == F32_INF) return x;
    uint32_t b;
    std::memcpy(&b, &x, 4);
    uint32_t a = b & 0x7fffffffu;
    uint32_t nb;
    if (a == 0) nb = 1u;                 // +-0 -> smallest positive subnormal
    else if (b == a) nb = b + 1;         // positive
    else nb = b - 1;                     // negative
    float r;
    std::memcpy(&r, &nb, 4);
    return r;
}
inline float next_down(float x) {  // f32::next_down
    if (x != x || x == -F32_INF) return x;
    uint32_t b;
    std::memcpy(&b, &x, 4);
    uint32_t a = b & 0x7fffffffu;
    uint32_t nb;
    if (a == 0) nb = 0x80000001u;        // +-0 -> smallest negative subnormal
    else if (b == a) nb = b - 1;         // positive
    else nb = b + 1;                     // negative
    float r;
    std::memcpy(&r, &nb, 4);
    return r;
}

struct V2 {
    float x = 0, y = 0;
};
struct V3 {
    float x = 0, y = 0, z = 0;
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 splat3(float v) { return V3{v, v, v}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 operator*(float s, V3 a) { return {a.x * s, a.y * s, a.z * s}; }  // vector.rs: f32 * V == V * f32
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V3 operator/(V3 a, V3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 vabs(V3 a) { return {std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }  // vector.rs:190-192 (no fma)
inline V3 cross(V3 a, V3 b) {                                                  // vector.rs:183-189
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline float mag_sq(V3 a) { return dot(a, a); }
inline float mag(V3 a) { return std::sqrt(mag_sq(a)); }
inline V3 normalized(V3 a) { return a * (1.0f / mag(a)); }  // vector.rs:1059-1061: self * mag().recip()
inline int max_axis(V3 a) {                                   // vector.rs:154-165 (note: max not updated on z)
    int axis = 0;
    float mx = a.x;
    if (a.y > mx) {
        mx = a.y;
        axis = 1;
    }
    if (a.z > mx) axis = 2;
    return axis;
}
inline float max_component(V3 a) { return rmax(rmax(a.x, a.y), a.z); }
inline V3 vmin(V3 a, V3 b) { return {rmin(a.x, b.x), rmin(a.y, b.y), rmin(a.z, b.z)}; }
inline V3 vmax(V3 a, V3 b) { return {rmax(a.x, b.x), rmax(a.y, b.y), rmax(a.z, b.z)}; }
inline V3 vclamp(V3 a, V3 lo, V3 hi) { return {rclamp(a.x, lo.x, hi.x), rclamp(a.y, lo.y, hi.y), rclamp(a.z, lo.z, hi.z)}; }

inline V2 operator+(V2 a, V2 b) { return {a.x + b.x, a.y + b.y}; }
inline V2 operator-(V2 a, V2 b) { return {a.x - b.x, a.y - b.y}; }
inline V2 operator*(V2 a, V2 b) { return {a.x * b.x, a.y * b.y}; }
inline V2 operator/(V2 a, float s) { return {a.x / s, a.y / s}; }
inline V2 v2min(V2 a, V2 b) { return {rmin(a.x, b.x), rmin(a.y, b.y)}; }
inline V2 v2max(V2 a, V2 b) { return {rmax(a.x, b.x), rmax(a.y, b.y)}; }
inline V2 v2clamp(V2 a, V2 lo, V2 hi) { return {rclamp(a.x, lo.x, hi.x), rclamp(a.y, lo.y, hi.y)}; }

// bounds.rs:76-162
struct B3 {
    V3 min, max;
    static B3 empty() { return {splat3(F32_INF), splat3(-F32_INF)}; }
    static B3 from_points(V3 a, V3 b) { return {vmin(a, b), vmax(a, b)}; }
    static B3 around(V3 b, V3 extent) {
        V3 half = extent / 2.0f;
        return {b - half, b + half};
    }
    V3 centroid() const { return (min + max) / 2.0f; }
    V3 clip(V3 p) const { return vclamp(p, min, max); }
    B3 clamp(V3 lo, V3 hi) const { return {vclamp(lo, min, max), vclamp(hi, min, max)}; }  // bounds.rs:100-104 (sic)
    V3 extent() const { return max - min; }
    float volume() const {
        V3 e = extent();
        return e.x * e.y * e.z;
    }
    float surface_area() const {
        V3 e = extent();
        return 2.0f * (e.x * e.y + e.y * e.z + e.x * e.z);
    }
    V3 offset(V3 p) const { return (p - min) / extent(); }
    V3 lerp(V3 v) const { return min + v * extent(); }
    float overlap(B3 o) const {
        V3 mn = vmax(min, o.min), mx = vmin(max, o.max);
        V3 e = mx - mn;
        if (e.x <= 0.0f || e.y <= 0.0f || e.z <= 0.0f) return 0.0f;
        return B3{mn, mx}.volume();
    }
};
inline B3 operator|(B3 a, B3 b) { return {vmin(a.min, b.min), vmax(a.max, b.max)}; }
inline B3 operator|(B3 a, V3 p) { return {vmin(a.min, p), vmax(a.max, p)}; }

// bounds.rs:18-75
struct B2 {
    V2 min, max;
    static B2 unit() { return {{0.0f, 0.0f}, {1.0f, 1.0f}}; }
    static B2 around(V2 b, V2 extent) {
        V2 half = extent / 2.0f;
        return {b - half, b + half};
    }
    V2 centroid() const { return (min + max) / 2.0f; }
    V2 clip(V2 p) const { return v2clamp(p, min, max); }
    B2 clamp(V2 lo, V2 hi) const { return {v2clamp(lo, min, max), v2clamp(hi, min, max)}; }
    V2 extent() const { return max - min; }
    float area() const {
        V2 e = extent();
        return e.x * e.y;
    }
    V2 lerp(V2 v) const { return min + v * extent(); }
    float overlap(B2 o) const {
        V2 mn = v2max(min, o.min), mx = v2min(max, o.max);
        V2 e = mx - mn;
        if (e.x <= 0.0f || e.y <= 0.0f) return 0.0f;
        return B2{mn, mx}.area();
    }
};

// onb.rs:12-30. NB the tangents are swapped relative to Duff et al.
struct ONB {
    V3 x, y, z;
    static ONB init_z(V3 z) {
        float sign = std::copysign(1.0f, z.z);
        float a = -1.0f / (sign + z.z);
        float b = z.x * z.y * a;
        V3 x = v3(b, sign + z.y * z.y * a, -z.y);
        V3 y = v3(1.0f + sign * z.x * z.x * a, sign * b, -sign * z.x);
        return {x, y, z};
    }
    V3 apply(V3 v) const { return v3(dot(x, v), dot(y, v), dot(z, v)); }
    V3 apply_inverse(V3 v) const { return x * v.x + y * v.y + z * v.z; }
};

// matrix.rs. Stored column-major as the reference does (cols[c][r]).
struct M4 {
    float c[4][4];  // c[col][row]
    static M4 from_rows(const float r[16]) {
        M4 m;
        for (int row = 0; row < 4; row++)
            for (int col = 0; col < 4; col++) m.c[col][row] = r[row * 4 + col];
        return m;
    }
    // matrix.rs:23-49: rows dotted with (v,w): x*m0 + y*m1 + z*m2 + w*m3, left to right.
    V3 mul_point(V3 p) const {
        float o[3];
        for (int r = 0; r < 3; r++) o[r] = c[0][r] * p.x + c[1][r] * p.y + c[2][r] * p.z + c[3][r] * 1.0f;
        return v3(o[0], o[1], o[2]);
    }
    V3 mul_vector(V3 v) const {
        float o[3];
        for (int r = 0; r < 3; r++) o[r] = c[0][r] * v.x + c[1][r] * v.y + c[2][r] * v.z + c[3][r] * 0.0f;
        return v3(o[0], o[1], o[2]);
    }
    // matrix.rs:109-178
    M4 inverse() const {
        const float m00 = c[0][0], m01 = c[0][1], m02 = c[0][2], m03 = c[0][3];
        const float m10 = c[1][0], m11 = c[1][1], m12 = c[1][2], m13 = c[1][3];
        const float m20 = c[2][0], m21 = c[2][1], m22 = c[2][2], m23 = c[2][3];
        const float m30 = c[3][0], m31 = c[3][1], m32 = c[3][2], m33 = c[3][3];
        float coef00 = (m22 * m33) - (m32 * m23), coef02 = (m12 * m33) - (m32 * m13), coef03 = (m12 * m23) - (m22 * m13);
        float coef04 = (m21 * m33) - (m31 * m23), coef06 = (m11 * m33) - (m31 * m13), coef07 = (m11 * m23) - (m21 * m13);
        float coef08 = (m21 * m32) - (m31 * m22), coef10 = (m11 * m32) - (m31 * m12), coef11 = (m11 * m22) - (m21 * m12);
        float coef12 = (m20 * m33) - (m30 * m23), coef14 = (m10 * m33) - (m30 * m13), coef15 = (m10 * m23) - (m20 * m13);
        float coef16 = (m20 * m32) - (m30 * m22), coef18 = (m10 * m32) - (m30 * m12), coef19 = (m10 * m22) - (m20 * m12);
        float coef20 = (m20 * m31) - (m30 * m21), coef22 = (m10 * m31) - (m30 * m11), coef23 = (m10 * m21) - (m20 * m11);
        float fac0[4] = {coef00, coef00, coef02, coef03}, fac1[4] = {coef04, coef04, coef06, coef07};
        float fac2[4] = {coef08, coef08, coef10, coef11}, fac3[4] = {coef12, coef12, coef14, coef15};
        float fac4[4] = {coef16, coef16, coef18, coef19}, fac5[4] = {coef20, coef20, coef22, coef23};
        float vec0[4] = {m10, m00, m00, m00}, vec1[4] = {m11, m01, m01, m01};
        float vec2[4] = {m12, m02, m02, m02}, vec3_[4] = {m13, m03, m03, m03};
        const float sa[4] = {1.0f, -1.0f, 1.0f, -1.0f}, sb[4] = {-1.0f, 1.0f, -1.0f, 1.0f};
        M4 ad;
        for (int i = 0; i < 4; i++) {
            float inv0 = (vec1[i] * fac0[i]) - (vec2[i] * fac1[i]) + (vec3_[i] * fac2[i]);
            float inv1 = (vec0[i] * fac0[i]) - (vec2[i] * fac3[i]) + (vec3_[i] * fac4[i]);
            float inv2 = (vec0[i] * fac1[i]) - (vec1[i] * fac3[i]) + (vec3_[i] * fac5[i]);
            float inv3 = (vec0[i] * fac2[i]) - (vec1[i] * fac4[i]) + (vec2[i] * fac5[i]);
            ad.c[0][i] = inv0 * sa[i];
            ad.c[1][i] = inv1 * sb[i];
            ad.c[2][i] = inv2 * sa[i];
            ad.c[3][i] = inv3 * sb[i];
        }
        float row0[4] = {ad.c[0][0], ad.c[1][0], ad.c[2][0], ad.c[3][0]};
        float d0 = c[0][0] * row0[0], d1 = c[0][1] * row0[1], d2 = c[0][2] * row0[2], d3 = c[0][3] * row0[3];
        float dot1 = d0 + d1 + d2 + d3;
        float recip = 1.0f / dot1;
        M4 out;
        for (int col = 0; col < 4; col++)
            for (int r = 0; r < 4; r++) out.c[col][r] = ad.c[col][r] * recip;
        return out;
    }
};

}  // namespace orc
