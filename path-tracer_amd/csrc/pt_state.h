// pt_state.h — HBM layout of the in-flight path state (structure of arrays, 16-byte columns so every lane
// moves whole dwordx4s and consecutive lanes touch consecutive addresses).
//
// One "slot" = one path. Slots are dense: the shade stage reads slot i of the `in` set and writes the
// surviving path to a freshly allocated slot of the `out` set (wave-aggregated atomic), so every stage always
// streams a compact [0, n_active) range — compaction is free because shade rewrites the state anyway.
// Paths that end while a shadow ray is still pending park their radiance in slots handed out from the TOP of
// the `out` set (capacity-1 downwards); the occlude stage finishes them.
#pragma once
#include "pt_types.h"

namespace pt {

// flags word (d_flags.w)
constexpr uint32_t FLAG_DEPTH_MASK = 0xffffu;
constexpr uint32_t FLAG_SPECULAR_BOUNCE = 1u << 16;
constexpr uint32_t FLAG_SECONDARY_TERMINATED = 1u << 17;
// Consecutive pass-through (material-less) surfaces crossed without a real bounce. The reference does not count
// them (path_tracer.rs:76-80), so a ray that re-hits a pass-through triangle at t ~ 0 forever (observed: about one
// path per million on a material-less icosphere) never returns there. The GPU ends such a path after
// kMaxPassThrough consecutive crossings and keeps what it has gathered.
constexpr uint32_t FLAG_PASS_SHIFT = 18;
constexpr uint32_t FLAG_PASS_MASK = 0xfffu << FLAG_PASS_SHIFT;
constexpr uint32_t kMaxPassThrough = 256;
// The path is inside a dielectric body (toggled by every specular transmission). Used ONLY to decide where a surviving
// path is queued — predicted-specular survivors fill the state set from the top, the others from the bottom
// (slot_of_index) — so that the lanes of a wave run the same material code; no result depends on it.
constexpr uint32_t FLAG_INSIDE = 1u << 30;
// A path straight out of ray generation: throughput 1, no radiance yet, no previous vertex, path id = slot. Ray generation
// writes origin, direction and sampler state only; the first shade pass fills in the rest instead of loading it.
constexpr uint32_t FLAG_FRESH = 1u << 31;
PT_HD bool passthrough_step(uint32_t& flags) {
    const uint32_t n = ((flags & FLAG_PASS_MASK) >> FLAG_PASS_SHIFT) + 1;
    flags = (flags & ~FLAG_PASS_MASK) | (n << FLAG_PASS_SHIFT);
    return n < kMaxPassThrough;
}

struct PathCols {
    float4* o_pdf;    // ray origin xyz, prev_bsdf_pdf
    float4* d_flags;  // ray direction xyz, bits(flags)
    float4* beta0;    // throughput, wavelengths 0-3
    float4* beta1;    //             wavelengths 4-7
    float4* L0;       // radiance so far 0-3
    float4* L1;       //                 4-7
    uint4* rng_px;    // PCG state lo, hi, pixel index (x + y*W), wavelength-table index
    float4* prev_p;   // previous interaction point xyz, bits(path id) (guided: row of the vertex store)
};
constexpr size_t kPathBytes = 8 * 16;  // bytes of state per slot

struct HitCols {
    float4* tb;      // t, b0, b1, b2
    uint2* prim;     // {leaf-order triangle index or kNoHit, the triangle's tag (material + 1 | has_normals << 31)}:
                     // with the tag in the hit record, shade's material and spectrum loads do not wait for the triangle fetch
};
constexpr size_t kHitBytes = 16 + 8;

// Shadow / connection ray queue entry (64 B):
//   q0 = {from.xyz, bits(target)}     target = slot | kTerminalBit (path), pixel (lightpath), vertex row (guided)
//   q1 = {to.xyz, bits(aux)}
//   q2,q3 = contribution, 8 wavelengths (path, guided) or 3 doubles RGB (lightpath)
struct ShadowCols {
    float4* q0;
    float4* q1;
    float4* q2;
    float4* q3;
};
constexpr size_t kShadowBytes = 4 * 16;
constexpr uint32_t kTerminalBit = 0x80000000u;

// Per-round device counters; round r's extend/shade read c[r].n_active, shade allocates from c[r+1].
struct RoundCounters {
    uint32_t n_active, n_shadow, n_terminal, n_segments;
};
constexpr int kMaxRounds = 1 << 16;
// Dense index j of a round's queue -> slot of the state set: the first n_bottom entries sit at the bottom, the rest
// count down from the top (capacity - 1, capacity - 2, ...).
PT_HD uint32_t slot_of_index(uint32_t j, uint32_t n_bottom, uint32_t capacity) { return j < n_bottom ? j : capacity - 1u - (j - n_bottom); }

// Guided integrator: per-path vertex store (PathVertex, path_guiding/mod.rs:281-289). One vertex = one 128-byte
// line of eight float4 {p.xyz|wi_pdf, wi.xyz|-, f/pdf (2), emitted (2), li_sample_contribution (2)}; vertex (row, path)
// sits at line row * n_paths + path. Survivors are compacted every round, so the lanes of a wave hold unrelated paths
// at different depths and the shade kernel's stores are scattered lines whatever the order; the record scan, which has
// one lane per path and walks the rows back to front, finds row v of 64 neighbouring paths in 8 KB of consecutive
// memory (with a path's rows back to back instead it read 128-byte pieces 2 KB apart: 2.2 TB/s, the rate of random
// DRAM accesses).
struct VertexStore {
    float4* lines;
    uint32_t n_rows;
    uint32_t n_paths;  // paths per batch (the context's capacity)
    PT_HD float4* at(uint32_t row, uint32_t path) const { return lines + ((size_t)row * n_paths + path) * 8; }
};
enum { VTX_P_PDF = 0, VTX_WI = 1, VTX_F0 = 2, VTX_F1 = 3, VTX_E0 = 4, VTX_E1 = 5, VTX_LI0 = 6, VTX_LI1 = 7 };
constexpr size_t kVertexBytes = 8 * 16;
constexpr size_t kRecordBytes = 40;  // record-queue entry per vertex (pt_guided.h: RecordQueue)
struct PathTail {     // per path id, written when the walk ends (guided)
    uint4* rng_px;    // final PCG state lo, hi, pixel, wavelength-table index
    uint32_t* nverts_flags;  // vertex count | FLAG_SECONDARY_TERMINATED
};

}  // namespace pt
