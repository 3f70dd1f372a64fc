// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header).
// CPU restatement of pt/src/spectrum/mod.rs:6-159,262-335 and pt/src/spectrum/rgb2spec.rs:8-71.
#pragma once
#include "cie_data.inc"
#include "math.hpp"

namespace orc {

static constexpr int NSW = 8;  // N_SAMPLED_WAVELENGTHS
static constexpr float LAMBDA_MIN = 360.0f;

static const float CIE_D65_F32[PT_CIE_N] = {PT_CIE_D65_VALUES};
static const double CIE_XYZ_F64[PT_CIE_N][3] = {PT_CIE_XYZ_VALUES};

inline size_t lambda_index(float l) { return f32_as_usize(l - LAMBDA_MIN); }  // mod.rs:9-11

struct SSpec {  // SampledSpectrum
    float v[NSW];
    static SSpec splat(float x) {
        SSpec s;
        for (int i = 0; i < NSW; i++) s.v[i] = x;
        return s;
    }
    static SSpec zero() { return splat(0.0f); }
    static SSpec one() { return splat(1.0f); }
    bool is_uniform() const {
        for (int i = 1; i < NSW; i++)
            if (!(v[i] == v[0])) return false;
        return true;
    }
    bool is_black() const {
        for (int i = 0; i < NSW; i++)
            if (!(v[i] == 0.0f)) return false;
        return true;
    }
    float average() const {  // mod.rs:94-96
        float s = 0.0f;
        for (int i = 0; i < NSW; i++) s += v[i];
        return s / (float)NSW;
    }
};
inline SSpec operator+(SSpec a, SSpec b) {
    for (int i = 0; i < NSW; i++) a.v[i] += b.v[i];
    return a;
}
inline SSpec operator*(SSpec a, SSpec b) {
    for (int i = 0; i < NSW; i++) a.v[i] *= b.v[i];
    return a;
}
inline SSpec operator*(SSpec a, float s) {
    for (int i = 0; i < NSW; i++) a.v[i] *= s;
    return a;
}
inline SSpec operator*(float s, SSpec a) { return a * s; }
inline SSpec operator/(SSpec a, SSpec b) {
    for (int i = 0; i < NSW; i++) a.v[i] /= b.v[i];
    return a;
}
inline SSpec operator/(SSpec a, float s) {
    for (int i = 0; i < NSW; i++) a.v[i] /= s;
    return a;
}

// Rust std f32::atanh (library/std/src/f32.rs, nightly-2024-11): 0.5*ln_1p((|x|+|x|)/(1-|x|)) with copysign.
inline float rust_atanhf(float x) {
    float ax = std::fabs(x);
    float ix = 0.5f * std::log1p((ax + ax) / (1.0f - ax));
    return std::copysign(ix, x);
}
inline float sample_visible_lambda(float u) {  // mod.rs:154-156
    return 538.0f - 138.888889f * rust_atanhf(0.85691062f - 1.82750197f * u);
}
inline float visible_pdf(float lambda) {  // mod.rs:157-159
    float c = std::cosh(0.0072f * (lambda - 538.0f));
    return 0.0039398042f / (c * c);
}

struct SWL {  // SampledWavelengths, mod.rs:15-18,102-128
    float lambdas[NSW];
    float pdfs[NSW];
    static SWL sample_visible(float u1d) {
        SWL r;
        for (int i = 0; i < NSW; i++) {
            float u = u1d + (float)i / (float)NSW;
            if (u >= 1.0f) u -= 1.0f;
            r.lambdas[i] = sample_visible_lambda(u);
            r.pdfs[i] = visible_pdf(r.lambdas[i]);
        }
        return r;
    }
    bool secondary_terminated() const {
        for (int i = 1; i < NSW; i++)
            if (!(pdfs[i] == 0.0f)) return false;
        return true;
    }
    void terminate_secondary() {
        if (secondary_terminated()) return;
        pdfs[0] /= (float)NSW;
        for (int i = 1; i < NSW; i++) pdfs[i] = 0.0f;
    }
};

inline void to_xyz(const SSpec& s, const SWL& swl, double xyz[3]) {  // mod.rs:81-90
    double acc[3] = {0, 0, 0};
    for (int i = 0; i < NSW; i++) {
        const double* c = CIE_XYZ_F64[lambda_index(swl.lambdas[i])];
        if (swl.pdfs[i] != 0.0f) {
            double w = (double)(s.v[i] / swl.pdfs[i]);
            acc[0] += c[0] * w;
            acc[1] += c[1] * w;
            acc[2] += c[2] * w;
        }
    }
    for (int k = 0; k < 3; k++) xyz[k] = acc[k] / (double)NSW;
}
inline void xyz_to_rgb(const double xyz[3], double rgb[3]) {  // mod.rs:130-136
    rgb[0] = xyz[0] * 3.240479 + xyz[1] * -1.537150 + xyz[2] * -0.498535;
    rgb[1] = xyz[0] * -0.969256 + xyz[1] * 1.875991 + xyz[2] * 0.041556;
    rgb[2] = xyz[0] * 0.055648 + xyz[1] * -0.204043 + xyz[2] * 1.057311;
}

// TristimulusSpectrum (mod.rs:262-306) in flattened form: sigmoid polynomial + scale (+ illuminant flag).
struct Spectrum {
    float c0 = 0, c1 = 0, c2 = 0, scale = 0;
    bool illuminant = false;
    float eval1(float lambda) const {
        float x = std::fma(lambda, std::fma(lambda, c0, c1), c2);  // mod.rs:325-334
        float s;
        if (std::isinf(x) && x > 0.0f) s = 1.0f;
        else if (std::isinf(x) && x < 0.0f) s = 0.0f;
        else s = 0.5f + x / (2.0f * std::sqrt(1.0f + x * x));
        float v = s * scale;                                        // mod.rs:303-305
        if (illuminant) v = v * CIE_D65_F32[lambda_index(lambda)];  // mod.rs:312-314
        return v;
    }
    SSpec eval(const float lambdas[NSW]) const {
        SSpec r;
        for (int i = 0; i < NSW; i++) r.v[i] = eval1(lambdas[i]);
        return r;
    }
};

// rgb2spec.rs:56-71
inline int find_interval(const float* values, size_t len, float x) {
    size_t last = len - 2, left = 0, size = last;
    while (size > 0) {
        size_t half = size >> 1, middle = left + half;
        if (values[middle] <= x) {
            left = middle;
            size -= half + 1;
        } else {
            size = half;
        }
    }
    return (int)(left < last ? left : last);
}
// rgb2spec.rs:8-54. coeff_data = whole srgb.coeffs file viewed as f32 (2 header words, 64 scale, 3*64^3*3 coeffs).
inline void fetch_coeffs(const float* coeff_data, const float rgb_in[3], float out[3]) {
    const float* model_scale = coeff_data + 2;
    const float* model_data = coeff_data + 66;
    const int res = 64;
    float rgb[3];
    for (int k = 0; k < 3; k++) rgb[k] = rclamp(rgb_in[k], 0.0f, 1.0f);
    int i = 0;  // max_by returns the LAST maximum
    for (int k = 1; k < 3; k++)
        if (rgb[k] >= rgb[i]) i = k;
    float z = rgb[i];
    float scale = ((float)res - 1.0f) / z;
    float x = rgb[(i + 1) % 3] * scale;
    float y = rgb[(i + 2) % 3] * scale;
    int xi = std::min(f32_as_i32(x), res - 2);
    int yi = std::min(f32_as_i32(y), res - 2);
    int zi = find_interval(model_scale, 64, z);
    int dx = 3, dy = 3 * res, dz = 3 * res * res;
    float x1 = x - (float)xi, x0 = 1.0f - x1;
    float y1 = y - (float)yi, y0 = 1.0f - y1;
    float z1 = (z - model_scale[zi]) / (model_scale[zi + 1] - model_scale[zi]);
    float z0 = 1.0f - z1;
    int offset = (((i * res + zi) * res + yi) * res + xi) * 3;
    for (int j = 0; j < 3; j++) {
        out[j] = ((model_data[offset] * x0 + model_data[offset + dx] * x1) * y0 +
                  (model_data[offset + dy] * x0 + model_data[offset + dy + dx] * x1) * y1) *
                     z0 +
                 ((model_data[offset + dz] * x0 + model_data[offset + dz + dx] * x1) * y0 +
                  (model_data[offset + dz + dy] * x0 + model_data[offset + dz + dy + dx] * x1) * y1) *
                     z1;
        offset += 1;
    }
}
// TristimulusSpectrum::new (mod.rs:272-302). coeff_data may be null when every colour is grey/black.
inline bool tristimulus_new(const float* coeff_data, const float rgb_in[3], bool illuminant, Spectrum* out) {
    Spectrum s;
    s.illuminant = illuminant;
    float scale = rmax(rmax(rgb_in[0], rgb_in[1]), rgb_in[2]);
    if (scale == 0.0f) {
        s.scale = scale;
        *out = s;
        return true;
    }
    scale *= 2.0f;
    float rgb[3] = {rgb_in[0] / scale, rgb_in[1] / scale, rgb_in[2] / scale};
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) {
        float d = rgb[0] * (1.0f - rgb[0]);
        s.c2 = (rgb[0] - 0.5f) / (d * d);
        s.scale = scale;
        *out = s;
        return true;
    }
    if (!coeff_data) return false;
    float c[3];
    fetch_coeffs(coeff_data, rgb, c);
    s.c0 = c[0];
    s.c1 = c[1];
    s.c2 = c[2];
    s.scale = scale;
    *out = s;
    return true;
}

}  // namespace orc
