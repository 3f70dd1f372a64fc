// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/math.hpp header).
// CPU restatement of pt/src/sampling.rs:3-70,149-219,251-283 (PCG32, ARLDS, Sampler, warps).
#pragma once
#include "math.hpp"

namespace orc {

inline uint64_t mix_bits(uint64_t v) {  // sampling.rs:149-156
    v ^= v >> 31;
    v *= 0x7fb5d329728ea185ull;
    v ^= v >> 27;
    v *= 0x81dadef4bc2dd44dull;
    v ^= v >> 33;
    return v;
}

struct PCG {  // sampling.rs:3-6,13-20,31-70
    uint64_t state = 0x853c49e6748fea9bull;
    uint64_t inc = 0xda3e39cb94b95bdbull;
    static constexpr uint64_t MULT = 0x5851f42d4c957f2dull;
    void set_sequence(uint64_t seq_index) { set_sequence_seeded(seq_index, mix_bits(seq_index)); }
    void set_sequence_seeded(uint64_t seq_index, uint64_t seed) {
        state = 0;
        inc = (seq_index << 1) | 1;
        (void)u32();
        state += seed;
        (void)u32();
    }
    void advance(uint64_t delta) {
        uint64_t cur_mult = MULT, cur_plus = inc, acc_mult = 1, acc_plus = 0;
        while (delta > 0) {
            if (delta & 1) {
                acc_mult *= cur_mult;
                acc_plus = acc_plus * cur_mult + cur_plus;
            }
            cur_plus = (cur_mult + 1) * cur_plus;
            cur_mult *= cur_mult;
            delta /= 2;
        }
        state = acc_mult * state + acc_plus;
    }
    uint32_t u32() {
        uint64_t old = state;
        state = old * MULT + inc;
        uint32_t xorshifted = (uint32_t)(((old >> 18) ^ old) >> 27);
        uint32_t rot = (uint32_t)(old >> 59);
        return (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    float f32() {  // min(1 - eps/2, u32 as f32 * 2^-32)
        return rmin(1.0f - F32_EPSILON / 2.0f, (float)u32() * 2.3283064365386963e-10f);
    }
};

struct ARLDS {  // sampling.rs:8-11,22-29,162-174
    float state = 0.0f;
    float inc = std::sqrt(2.0f) - 1.0f;
    static ARLDS make(size_t n_) {
        float n = (float)n_;
        float inc = (n + std::sqrt(n * n + 4.0f)) / 2.0f;
        inc = inc - std::floor(inc);
        ARLDS a;
        a.state = 0.0f;
        a.inc = inc;
        return a;
    }
    float advance() {
        state += inc;
        state -= std::floor(state);
        return state;
    }
};

struct Sampler {  // sampling.rs:176-200
    PCG pcg;
    ARLDS arlds;
    void init_pixel(size_t x, size_t y, size_t i) {
        uint64_t seed = mix_bits(((uint64_t)x << 32) | (uint64_t)y);
        pcg.set_sequence(seed);
        pcg.advance(0xffffull * (uint64_t)i);
        if (i == 0) arlds = ARLDS::make(1);
    }
    float wavelength_sample() { return arlds.advance(); }
    float sample1d() { return pcg.f32(); }
    void sample2d(float out[2]) {
        out[0] = sample1d();
        out[1] = sample1d();
    }
};

inline void uniform_barycentric(const float u[2], float b[3]) {  // sampling.rs:202-213
    b[0] = b[1] = b[2] = 0.0f;
    if (u[0] < u[1]) {
        b[0] = u[0] / 2.0f;
        b[1] = u[1] - b[0];
    } else {
        b[1] = u[1] / 2.0f;
        b[0] = u[0] - b[1];
    }
    b[2] = 1.0f - b[0] - b[1];
}
inline void uniform_disk_polar(const float u[2], float d[2]) {  // sampling.rs:215-219
    float r = std::sqrt(u[0]);
    float theta = TWO_PI * u[1];
    d[0] = r * std::cos(theta);
    d[1] = r * std::sin(theta);
}
inline V3 cosine_hemisphere(const float u[2]) {  // sampling.rs:251-255
    float d[2];
    uniform_disk_polar(u, d);
    float z = safe_sqrt(1.0f - d[0] * d[0] - d[1] * d[1]);
    return v3(d[0], d[1], z);
}
inline float cosine_hemisphere_pdf(float cos_theta) { return cos_theta * INV_PI; }  // sampling.rs:265-267

inline size_t sample_discrete(float u, const float* vals, size_t n) {  // sampling.rs:273-283
    float total = 0.0f;                                                  // iter().sum::<f32>() starts at 0.0... (-0.0 in newer std; irrelevant)
    for (size_t i = 0; i < n; i++) total += vals[i];
    float inv_total = 1.0f / total;
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) {
        sum += vals[i] * inv_total;
        if (sum >= u) return i;
    }
    return n - 1;
}

}  // namespace orc
