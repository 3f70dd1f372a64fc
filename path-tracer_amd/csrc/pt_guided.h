// pt_guided.h — device side of the `guided` integrator (Müller et al. practical path guiding as the reference
// implements it): SD-tree lookups, guided direction sampling, the per-path vertex store and the back-to-front
// record pass. Restates pt/src/integrator/path_guiding/{mod,stree,dtree}.rs; quirks kept on purpose are marked.
#pragma once
#include "pt_kernels.h"

namespace pt {

// Build with -DPT_DBG_TIMING to attribute k_shade_guided's wave time to its sections (s_memtime deltas summed over
// all waves into g_dbg_cycles; ptgpu_guided_end prints them). Compiled out otherwise.
#ifdef PT_DBG_TIMING
// Wave-level accounting: the time since the wave's previous tick (whichever lanes took it) goes to section k, once as
// cycles and once weighted by the lanes active at the tick. Kept in LDS per wave by the first active lane, so ticks
// inside divergent code are charged in the order the wave really executes them.
constexpr int kDbgSections = 24;
__device__ unsigned long long g_dbg_cycles[2 * kDbgSections];
PT_D void dbg_tick(unsigned long long* w, int k) {
    const unsigned long long now = (unsigned long long)clock64();
    const unsigned long long ex = __ballot(true);
    if ((int)__lane_id() == __ffsll((long long)ex) - 1) {
        const unsigned long long dt = now - w[0];
        w[0] = now;
        w[1 + k] += dt;
        w[1 + kDbgSections + k] += dt * (unsigned long long)__popcll(ex);
    }
}
#define PT_TICK_INIT() __shared__ unsigned long long sm_dbg[4 * (1 + 2 * kDbgSections)]; unsigned long long* dbg_w = sm_dbg + (threadIdx.x / 64u) * (1 + 2 * kDbgSections); \
    if (__lane_id() == 0) { for (int k_ = 1; k_ < 1 + 2 * kDbgSections; k_++) dbg_w[k_] = 0; dbg_w[0] = (unsigned long long)clock64(); }
#define PT_TICK(k) dbg_tick(dbg_w, k)
#define PT_TICK_FLUSH() do { if (__lane_id() == 0) for (int k_ = 0; k_ < 2 * kDbgSections; k_++) if (dbg_w[1 + k_]) atomicAdd(&g_dbg_cycles[k_], dbg_w[1 + k_]); } while (0)
#define PT_DBG_PARAMS , unsigned long long* dbg_w
#define PT_DBG_ARGS , dbg_w
#else
#define PT_DBG_PARAMS
#define PT_DBG_ARGS
#define PT_TICK_INIT() do {} while (0)
#define PT_TICK(k) do {} while (0)
#define PT_TICK_FLUSH() do {} while (0)
#endif

#ifndef PT_SORT_SHADE_GUIDED
#define PT_SORT_SHADE_GUIDED 1
#endif
#ifndef PT_SCAN_WAVES
#define PT_SCAN_WAVES 3  // k_guide_scan holds two rows of lines (64 registers): 4 waves per SIMD spill 152 bytes and run slower (30.6 against 28.2 ms per step)
#endif
#ifndef PT_GUIDED_DIFFUSE_WAVES
#define PT_GUIDED_DIFFUSE_WAVES 3
#endif

struct GuidedState;  // host-side (ptgpu_guided_api.inc)
struct GuidedPersist;
void guided_free(GuidedState* g);
void guided_persist_free(GuidedPersist* p);

// Flat SD-tree in HBM (built by HostSdTree::flatten).
//   snodes[i]    = {left, right, split_axis, leaf_id}; left == 0 <=> leaf
//   leaf_info[l] = {first node of the leaf's SAMPLING quadtree in s_*, first node of its COLLECTOR quadtree in c_*,
//                   sampler mean == 0, -}
//   quadtree node k of a tree at offset o: rad[o+k] = the 4 child sums, child[o+k] = 4 x u16 links (0 = leaf child);
//   child numbering bit0 = x >= mid, bit1 = y >= mid (dtree.rs:331-352).
// The collector sums c_rad and the per-leaf weights c_weight share ONE float buffer (c_rad first), which is what
// ptgpu_guided_stats hands to the multi-GPU driver for its all-reduce.
struct alignas(32) SNode {
    float4 rad;
    uint2 child;
    uint2 pad;
};
// Two levels of a sampling quadtree in one 128-byte line (PT_TWO_LEVEL): the node's own four sums and links, then the four
// sums and the links of each of its interior children. A record exists for every interior node at an even depth; links
// between records (glink) are record indices relative to the leaf's first record, which sits at the arena offset of the
// leaf's first node (a leaf has no more records than nodes). A descent fetches one line per two levels: the second level's
// 24 bytes are in the line the first level's fetch has just brought in. Built from the one-level arena by k_build_two_level.
struct alignas(128) SNode2 {
    float4 rad;
    uint2 child;   // the node's four one-level links: only == 0 (leaf cell) matters here
    uint2 aux;     // .x: the node's index in the one-level arena (the builder's queue)
    float4 grad[4];
    uint2 glink[4];
};
static_assert(sizeof(SNode2) == 128, "one line");
#ifndef PT_TWO_LEVEL
#define PT_TWO_LEVEL 0
#endif

struct SdTreeDev {
    const uint4* snodes;
    const uint2* top;  // {entry, sampler word of the leaf}: jump table over the first kStreeTopLevels decisions (pt_refine.h: k_build_top)
    const uint4* leaf_info;
    const SNode* s_node;  // sampling arena: a node's four child sums and four child links in one 32-byte record (one line per level)
    const SNode2* s_node2;  // the same trees, two levels per 128-byte record (PT_TWO_LEVEL)
    float4* c_rad;
    const uint2* c_child;
    float4* c_lazy;          // per collector node: radiance owed to ALL cells below child c (binned record pass; k_push_lazy)
    const uint32_t* c_size;  // per leaf: nodes of its collector quadtree (first node: leaf_info[l].y)
    float* c_weight;
    float* w_private;     // kWeightCopies x n_leaves: record counts per leaf, one copy per group of workgroups (folded into
                          // c_weight by k_fold_weights when the wave is done) — every record bumps its leaf's count, and
                          // same-address atomics serialise at the memory side
    uint32_t n_leaves;
    V3 blo, bhi;
    uint32_t iteration;  // Guider::iteration: 0 = no guiding yet
};

PT_HD uint32_t child_link(uint2 packed, int c) { return (c < 2 ? packed.x >> (16 * c) : packed.y >> (16 * (c - 2))) & 0xffffu; }
PT_HD float comp4(float4 v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); }

PT_D V3 square_to_dir(V2 p) {  // path_guiding/mod.rs:404-410 (cylindrical, area preserving; z = world z)
    float ct = 2.0f * p.x - 1.0f;
    float phi = kTwoPi * p.y;
    float st = safe_sqrtf(1.0f - ct * ct);
    float s, c;
    sincosf(phi, &s, &c);
    return mk3(st * c, st * s, ct);
}
PT_D V2 dir_to_square(V3 d) {  // path_guiding/mod.rs:412-421
    float ct = clampf(d.z, -1.0f, 1.0f);
    float phi = atan2f(d.y, d.x);
    while (phi < 0.0f) phi += kTwoPi;
    return clamp2(V2{(ct + 1.0f) / 2.0f, phi * kInvTwoPi}, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
}

// STree::leaf (stree.rs:202-229): halve along the node's axis until a leaf; also returns the leaf's box.
// The first kStreeTopLevels halvings are done without touching the tree (axis = depth % 3, split at the midpoint:
// the decisions are the reference's, bit for bit) and index a jump table; only deeper levels chase node links.
// `sampler_word` (optional) receives the leaf's sampling-arena offset | mean_is_zero << 31.
template <bool WANT_BOX>
PT_D uint32_t stree_leaf(const SdTreeDev& t, V3 p, V3& lo, V3& hi, uint32_t* sampler_word = nullptr) {
    lo = t.blo;
    hi = t.bhi;
    uint32_t code = 0;
#pragma unroll
    for (uint32_t d = 0; d < kStreeTopLevels; d++) {
        const int ax = (int)(d % 3u);
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        const bool right = !(get(p, ax) < mid);
        code = (code << 1) | (right ? 1u : 0u);
        if (right) set(lo, ax, mid);
        else set(hi, ax, mid);
    }
    const uint2 entry2 = t.top[code];
    const uint32_t entry = entry2.x;
    if (entry & 0x80000000u) {
        if (sampler_word) *sampler_word = entry2.y;
        if (WANT_BOX) {  // the leaf sits above the table's depth: redo its (fewer) halvings for the box
            const uint32_t depth = entry & 31u;
            lo = t.blo;
            hi = t.bhi;
            for (uint32_t d = 0; d < depth; d++) {
                const int ax = (int)(d % 3u);
                const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
                if (get(p, ax) < mid) set(hi, ax, mid);
                else set(lo, ax, mid);
            }
        }
        return (entry >> 5) & 0x3ffffffu;
    }
    uint32_t ni = entry >> 5;
    for (int guard = 0; guard < 128; guard++) {
        const uint4 n = t.snodes[ni];
        if (n.x == 0) {
            if (sampler_word) *sampler_word = n.y;
            return n.w;
        }
        const int ax = (int)n.z;
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        if (get(p, ax) < mid) {
            set(hi, ax, mid);
            ni = n.x;
        } else {
            set(lo, ax, mid);
            ni = n.y;
        }
    }
    if (sampler_word) *sampler_word = t.snodes[ni].y;
    return t.snodes[ni].w;
}

// The same descent cut in two for the shade kernel, which wants the table load in flight while it does other work:
// stree_code = the register halvings (index of the jump-table entry), stree_resolve = what follows the load. Only the
// leaf's sampler word (sampling-arena offset | mean_is_zero << 31) comes out.
PT_D uint32_t stree_code(const SdTreeDev& t, V3 p) {
    V3 lo = t.blo, hi = t.bhi;
    uint32_t code = 0;
#pragma unroll
    for (uint32_t d = 0; d < kStreeTopLevels; d++) {
        const int ax = (int)(d % 3u);
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        const bool right = !(get(p, ax) < mid);
        code = (code << 1) | (right ? 1u : 0u);
        if (right) set(lo, ax, mid);
        else set(hi, ax, mid);
    }
    return code;
}
PT_D uint32_t stree_resolve(const SdTreeDev& t, uint2 entry2, V3 p) {
    if (entry2.x & 0x80000000u) return entry2.y;
    // deeper than the table: redo the halvings for the box (rare; cheaper than keeping it in registers), then chase links
    V3 lo = t.blo, hi = t.bhi;
    for (uint32_t d = 0; d < kStreeTopLevels; d++) {
        const int ax = (int)(d % 3u);
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        if (get(p, ax) < mid) set(hi, ax, mid);
        else set(lo, ax, mid);
    }
    uint32_t ni = entry2.x >> 5;
    for (int guard = 0; guard < 128; guard++) {
        const uint4 n = t.snodes[ni];
        if (n.x == 0) return n.y;
        const int ax = (int)n.z;
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        if (get(p, ax) < mid) {
            set(hi, ax, mid);
            ni = n.x;
        } else {
            set(lo, ax, mid);
            ni = n.y;
        }
    }
    return t.snodes[ni].y;
}

// Quadtree cells are dyadic: cell (ix, iy) at `level` spans [ix, ix+1] * 2^-level. The reference halves boxes with
// (min+max)/2, which is exact for these values, so integer cell coordinates reproduce its boxes bit for bit.
struct Cell {
    uint32_t ix, iy, level;
};
PT_HD Box2 cell_box(Cell c) {
    const float s = u2f((127u - c.level) << 23);  // 2^-level
    return Box2{{(float)c.ix * s, (float)c.iy * s}, {(float)(c.ix + 1) * s, (float)(c.iy + 1) * s}};
}
PT_HD int cell_child_of(Cell c, V2 p, Cell& out) {  // child_index (dtree.rs:331-352)
    const Box2 b = cell_box(c);
    const V2 mid = box2_centroid(b);
    int r = 0;
    if (!(p.x < mid.x)) r |= 1;
    if (!(p.y < mid.y)) r |= 2;
    out = Cell{2 * c.ix + (uint32_t)(r & 1), 2 * c.iy + (uint32_t)(r >> 1), c.level + 1};
    return r;
}
constexpr uint32_t kMaxQuadLevel = 24;
constexpr uint32_t kWeightCopies = 16;
// The per-level pdf factors of a descent wait in LDS (dtree_pdf below). Each wave owns a [kMaxQuadLevel + 1][64] block
// of the workgroup's array — lane-interleaved, conflict-free — so that a wave may reuse its block for something else
// once its descent is over without caring where the workgroup's other waves are.
constexpr uint32_t kScaleStride = 64;
constexpr uint32_t kScaleWaveFloats = (kMaxQuadLevel + 1) * kScaleStride;
PT_D float* scales_of(float* sm, uint32_t tid) { return sm + (tid / 64u) * kScaleWaveFloats + (tid % 64u); }

// DTree::sample (dtree.rs:98-153): one float per level, two at the leaf.
PT_D V2 dtree_sample(const SdTreeDev& t, uint4 leaf, Rng& rng) {
    if (leaf.z) {
        const float a = rng_f32(rng), b = rng_f32(rng);
        return V2{a, b};
    }
    uint32_t ni = 0;
    Cell cell{0, 0, 0};
    for (;;) {
        const float4 r = t.s_node[leaf.x + ni].rad;
        const uint2 links = t.s_node[leaf.x + ni].child;
        const int c = sample_discrete4(rng_f32(rng), r.x, r.y, r.z, r.w);
        const Cell child{2 * cell.ix + (uint32_t)(c & 1), 2 * cell.iy + (uint32_t)(c >> 1), cell.level + 1};
        const uint32_t link = child_link(links, c);
        if (link == 0 || child.level >= kMaxQuadLevel) {
            const float a = rng_f32(rng), b = rng_f32(rng);
            const Box2 cb = cell_box(child);
            V2 s = cb.lo + V2{a, b} * box2_extent(cb);
            s = min2(s, cb.hi - V2{kEps, kEps});
            return clamp2(s, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
        }
        ni = link;
        cell = child;
    }
}
// DTree::pdf (dtree.rs:155-179): prod over levels of 4 * child / sum, times 1 / 4 pi.
// `scales` is this lane's column of its wave's [kMaxQuadLevel + 1][64] LDS block (scales_of): the per-level factors are multiplied
// innermost first, as the reference's recursion does, so they have to be kept until the leaf is reached.
PT_D float dtree_pdf(const SdTreeDev& t, uint4 leaf, V2 p, float* scales) {
    if (leaf.z) return kInvFourPi;
    int n = 0;
    uint32_t ni = 0;
    Cell cell{0, 0, 0};
    float result;
    for (;;) {
        const float4 r = t.s_node[leaf.x + ni].rad;
        const uint2 links = t.s_node[leaf.x + ni].child;
        Cell child;
        const int c = cell_child_of(cell, p, child);
        const float rc = comp4(r, c);
        if (rc == 0.0f) {
            result = 0.0f;
            break;
        }
        const float scale = 4.0f * rc / ((((0.0f + r.x) + r.y) + r.z) + r.w);
        const uint32_t link = child_link(links, c);
        if (link == 0 || child.level >= kMaxQuadLevel) {
            result = scale;
            break;
        }
        scales[(n++) * kScaleStride] = scale;
        ni = link;
        cell = child;
    }
    for (int i = n - 1; i >= 0; i--) result = scales[i * kScaleStride] * result;  // innermost first, as the recursion multiplies
    return result * kInvFourPi;
}

// One descent that serves both halves of Guider::sample: lanes with `sample` set draw a direction (DTree::sample),
// the others look up the pdf of the direction they bring in `sq` (DTree::pdf). A sampled point lies in the cell it
// was drawn from, so DTree::pdf of it would visit the very nodes the sampling descent visited and multiply the same
// factors: the sampling lanes collect them on the way down instead of descending a second time. The rare cases where
// that argument does not hold (the point was pushed out of its cell by the epsilon/clamp adjustments, or a child of
// zero weight was drawn) fall back to the separate pdf descent. Returns the tree pdf; `sq` is an output for sampling lanes.
// `root` is the tree's first node, loaded by the caller ahead of time (one memory round trip off the dependent chain).
PT_D float dtree_walk(const SdTreeDev& t, uint4 leaf, bool sample, Rng& rng, V2& sq, float* scales, const SNode& root) {
    if (leaf.z) {
        if (sample) {
            const float a = rng_f32(rng), b = rng_f32(rng);
            sq = V2{a, b};
        }
        return kInvFourPi;
    }
    int n = 0;
    uint32_t ni = 0;
    Cell cell{0, 0, 0};
    float result;
    bool redo = false;
    float4 r = root.rad;
    uint2 links = root.child;
    for (;;) {
        int c;
        Cell child;
        if (sample) {
            c = sample_discrete4(rng_f32(rng), r.x, r.y, r.z, r.w);
            child = Cell{2 * cell.ix + (uint32_t)(c & 1), 2 * cell.iy + (uint32_t)(c >> 1), cell.level + 1};
        } else {
            c = cell_child_of(cell, sq, child);
        }
        const float rc = comp4(r, c);
        if (rc == 0.0f) {
            if (!sample) {
                result = 0.0f;
                break;
            }
            redo = true;  // DTree::pdf would stop here; DTree::sample goes on
        }
        const float scale = 4.0f * rc / ((((0.0f + r.x) + r.y) + r.z) + r.w);
        const uint32_t link = child_link(links, c);
        cell = child;
        if (link == 0 || child.level >= kMaxQuadLevel) {
            result = scale;
            break;
        }
        scales[(n++) * kScaleStride] = scale;
        ni = link;
        r = t.s_node[leaf.x + ni].rad;
        links = t.s_node[leaf.x + ni].child;
    }
    if (sample) {
        const float a = rng_f32(rng), b = rng_f32(rng);
        const Box2 cb = cell_box(cell);
        V2 s = cb.lo + V2{a, b} * box2_extent(cb);
        s = min2(s, cb.hi - V2{kEps, kEps});
        sq = clamp2(s, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
        if (redo || !(sq.x >= cb.lo.x && sq.x < cb.hi.x && sq.y >= cb.lo.y && sq.y < cb.hi.y)) return dtree_pdf(t, leaf, sq, scales);
    }
    for (int i = n - 1; i >= 0; i--) result = scales[i * kScaleStride] * result;  // innermost first, as the recursion multiplies
    return result * kInvFourPi;
}

// dtree_pdf / dtree_walk over the two-level records: the same choices, the same factors in the same order.
PT_D float dtree_pdf2(const SdTreeDev& t, uint4 leaf, V2 p, float* scales) {
    if (leaf.z) return kInvFourPi;
    int n = 0;
    const SNode2* rec = t.s_node2 + leaf.x;
    Cell cell{0, 0, 0};
    float4 r = rec->rad;
    uint2 links = rec->child;
    float result;
    for (bool second = false;; second = !second) {
        Cell child;
        const int c = cell_child_of(cell, p, child);
        const float rc = comp4(r, c);
        if (rc == 0.0f) {
            result = 0.0f;
            break;
        }
        const float scale = 4.0f * rc / ((((0.0f + r.x) + r.y) + r.z) + r.w);
        const uint32_t link = child_link(links, c);
        if (link == 0 || child.level >= kMaxQuadLevel) {
            result = scale;
            break;
        }
        scales[(n++) * kScaleStride] = scale;
        cell = child;
        if (!second) {
            r = rec->grad[c];
            links = rec->glink[c];
        } else {
            rec = t.s_node2 + leaf.x + link;
            r = rec->rad;
            links = rec->child;
        }
    }
    for (int i = n - 1; i >= 0; i--) result = scales[i * kScaleStride] * result;
    return result * kInvFourPi;
}
PT_D float dtree_walk2(const SdTreeDev& t, uint4 leaf, bool sample, Rng& rng, V2& sq, float* scales, float4 root_rad, uint2 root_child) {
    if (leaf.z) {
        if (sample) {
            const float a = rng_f32(rng), b = rng_f32(rng);
            sq = V2{a, b};
        }
        return kInvFourPi;
    }
    int n = 0;
    const SNode2* rec = t.s_node2 + leaf.x;
    Cell cell{0, 0, 0};
    float result;
    bool redo = false;
    float4 r = root_rad;
    uint2 links = root_child;
    for (bool second = false;; second = !second) {
        int c;
        Cell child;
        if (sample) {
            c = sample_discrete4(rng_f32(rng), r.x, r.y, r.z, r.w);
            child = Cell{2 * cell.ix + (uint32_t)(c & 1), 2 * cell.iy + (uint32_t)(c >> 1), cell.level + 1};
        } else {
            c = cell_child_of(cell, sq, child);
        }
        const float rc = comp4(r, c);
        if (rc == 0.0f) {
            if (!sample) {
                result = 0.0f;
                break;
            }
            redo = true;
        }
        const float scale = 4.0f * rc / ((((0.0f + r.x) + r.y) + r.z) + r.w);
        const uint32_t link = child_link(links, c);
        cell = child;
        if (link == 0 || child.level >= kMaxQuadLevel) {
            result = scale;
            break;
        }
        scales[(n++) * kScaleStride] = scale;
        if (!second) {  // the child's sums and links: same line as the node's
            r = rec->grad[c];
            links = rec->glink[c];
        } else {  // the grandchild's record
            rec = t.s_node2 + leaf.x + link;
            r = rec->rad;
            links = rec->child;
        }
    }
    if (sample) {
        const float a = rng_f32(rng), b = rng_f32(rng);
        const Box2 cb = cell_box(cell);
        V2 s = cb.lo + V2{a, b} * box2_extent(cb);
        s = min2(s, cb.hi - V2{kEps, kEps});
        sq = clamp2(s, V2{0.0f, 0.0f}, V2{1.0f, 1.0f});
        if (redo || !(sq.x >= cb.lo.x && sq.x < cb.hi.x && sq.y >= cb.lo.y && sq.y < cb.hi.y)) return dtree_pdf2(t, leaf, sq, scales);
    }
    for (int i = n - 1; i >= 0; i--) result = scales[i * kScaleStride] * result;
    return result * kInvFourPi;
}

// Builds the two-level records of every spatial leaf's sampling quadtree from the one-level arena: one thread per leaf,
// breadth first over the interior nodes at even depths, the records themselves serving as the queue (aux.x = node).
__global__ __launch_bounds__(kBlock) void k_build_two_level(const SNode* s1, SNode2* s2, const uint32_t* s_off, uint32_t n_leaves) {
    const uint32_t leaf = blockIdx.x * blockDim.x + threadIdx.x;
    if (leaf >= n_leaves) return;
    const uint32_t base = s_off[leaf];
    const SNode* in = s1 + base;
    SNode2* out = s2 + base;
    out[0].aux = make_uint2(0u, 0u);
    uint32_t next = 1;
    for (uint32_t r = 0; r < next; r++) {
        const uint32_t node = out[r].aux.x;
        const SNode N = in[node];
        SNode2 rec;
        rec.rad = N.rad;
        rec.child = N.child;
        rec.aux = make_uint2(node, 0u);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            rec.grad[c] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            uint32_t gl[4] = {0u, 0u, 0u, 0u};
            const uint32_t link = child_link(N.child, c);
            if (link != 0) {
                const SNode C = in[link];
                rec.grad[c] = C.rad;
                for (int g = 0; g < 4; g++) {
                    const uint32_t l2 = child_link(C.child, g);
                    if (l2 != 0) {
                        out[next].aux = make_uint2(l2, 0u);
                        gl[g] = next++;
                    }
                }
            }
            rec.glink[c] = make_uint2(gl[0] | (gl[1] << 16), gl[2] | (gl[3] << 16));
        }
        out[r] = rec;
    }
}

// Guider::sample (path_guiding/mod.rs:315-368). `sampler_word` and `root` are the spatial leaf's sampler word and its
// quadtree's first node, fetched by the caller while it did other work (unused for iteration 0 / specular surfaces).
PT_D bool guider_sample(const SdTreeDev& t, const Bsdf& bsdf, const Frame& fr, V3 ng, V3 wo, Rng& rng, float* lds_scales, uint32_t sampler_word, const SNode& root,
                        BsdfSample& s PT_DBG_PARAMS) {
    if (t.iteration == 0 || bsdf_is_specular(bsdf)) {
        const float u1d = rng_f32(rng), ua = rng_f32(rng), ub = rng_f32(rng);
        const bool ok = bsdf_sample(bsdf, fr, ng, wo, u1d, ua, ub, true, s);
        PT_TICK(14);
        return ok;
    }
    const uint4 leaf = make_uint4(sampler_word & 0x7fffffffu, 0u, sampler_word >> 31, 0u);  // leaf_info's x and z
    PT_TICK(15);
    // Both branches need the tree's pdf at the chosen direction: one quadtree descent per wave serves the lanes
    // that sample from the tree and the lanes that only evaluate it (dtree_walk). Values are unchanged.
    const bool use_tree = rng_f32(rng) < 0.5f;
    V2 sq = V2{0.0f, 0.0f};
    if (!use_tree) {
        const float u1d = rng_f32(rng), ua = rng_f32(rng), ub = rng_f32(rng);
        if (!bsdf_sample(bsdf, fr, ng, wo, u1d, ua, ub, true, s)) return false;
        sq = dir_to_square(s.wi);
        PT_TICK(5);
    }
    PT_TICK(16);
#if PT_TWO_LEVEL
    const float tree_pdf = dtree_walk2(t, leaf, use_tree, rng, sq, lds_scales, root.rad, root.child);
#else
    const float tree_pdf = dtree_walk(t, leaf, use_tree, rng, sq, lds_scales, root);
#endif
    PT_TICK(6);
    if (use_tree) {
        s.wi = square_to_dir(sq);
        const float bp = bsdf_pdf(bsdf, fr, wo, s.wi) * 0.5f;
        if (bp == 0.0f) return false;  // the tree proposed a direction the BSDF cannot produce: the walk ends here
        s.pdf = bp + tree_pdf * 0.5f;
        s.f = bsdf_eval(bsdf, fr, ng, wo, s.wi, true);
        s.specular = false;
        s.terminate_secondary = false;
        PT_TICK(7);
    } else {
        s.pdf = 0.5f * s.pdf + 0.5f * tree_pdf;
    }
    PT_TICK(17);
    return true;
}

// DTree::record with the box filter (dtree.rs:200-207,235-289): weight += 1; splat `radiance` over every collector
// leaf cell the footprint (a box of the size of the cell containing p, centred on p, clipped to the unit square)
// overlaps, in proportion to the overlap. f32 atomics straight into HBM.
//
// Called by the WHOLE wave (`active` marks lanes that carry a record). What bounds this kernel is the memory-side
// float atomic path, in two currencies (MI355X_MICROARCH.md "Global float atomics"): 64-byte REQUESTS chip-wide and
// atomic wave-INSTRUCTIONS per CU (one per ~50 ns, however few lanes it carries). So
//  * the four child sums of a node (16 contiguous bytes) are added by the four lanes of a quad in one instruction —
//    the quad takes its members' nodes in turn, lane j adding child j's share: sibling cells cost one request;
//  * the adds are not issued as the walk finds them (every step of every lane would cost the wave four sparse
//    instructions) but parked per lane in LDS (node, four shares) and flushed when the walk is over, entry k of all
//    lanes together: 4 x (the largest number of touched nodes in the wave) instructions per record instead of
//    4 x (the longest walk).
constexpr uint32_t kSplatPending = 4;                       // parked (node, shares) entries per lane
constexpr uint32_t kSplatStackLds = 8;                      // walk-stack entries kept in LDS (two dwords each); deeper ones spill to scratch
constexpr uint32_t kSplatPendingWords = 5 * kSplatPending;  // LDS dwords per lane: [entry][field][lane]
PT_D void splat_flush(const SdTreeDev& t, uint32_t* pend, uint32_t& n_pend) {
    const uint32_t lane = __lane_id(), quad_base = lane & ~3u, me = lane & 3u;
    uint32_t most = n_pend;
    for (int off = 32; off > 0; off >>= 1) most = max(most, (uint32_t)__shfl_xor((int)most, off));
    for (uint32_t k = 0; k < most; k++) {
        const bool have = k < n_pend;
        const uint32_t node = have ? pend[(5 * k) * kBlock] : 0u;
        float share[4];
#pragma unroll
        for (int c = 0; c < 4; c++) share[c] = have ? u2f(pend[(5 * k + 1 + c) * kBlock]) : 0.0f;
#pragma unroll
        for (uint32_t r = 0; r < 4; r++) {
            const int src = (int)(quad_base + r);
            const uint32_t node_r = (uint32_t)__shfl((int)node, src);
            const float s0 = __shfl(share[0], src), s1 = __shfl(share[1], src), s2 = __shfl(share[2], src), s3 = __shfl(share[3], src);
            const float mine = me == 0u ? s0 : (me == 1u ? s1 : (me == 2u ? s2 : s3));
            if (mine != 0.0f) unsafeAtomicAdd(reinterpret_cast<float*>(t.c_rad + node_r) + me, mine);
        }
    }
    n_pend = 0;
}
PT_D void dtree_record_box(const SdTreeDev& t, uint32_t leaf_id, uint4 leaf, V2 p, float radiance, bool active, uint32_t* pend) {
    // Walk stack: (node, cell) packed in two dwords — node 16 bits, iy 23 bits, and ix behind a leading 1 whose position
    // is the level (24 bits): every cell that can be pushed (level < kMaxQuadLevel = 24) fits. The first kSplatStackLds
    // entries live in LDS, the rest in scratch.
    struct Item {
        uint32_t node;
        Cell cell;
    };
    uint2 deep[3 * kMaxQuadLevel + 4 - kSplatStackLds];
    uint32_t* lds_stack = pend + kSplatPendingWords * kBlock;
    int sp = 0;
    auto push = [&](const Item& it) {
        const uint2 w = make_uint2(it.node | (it.cell.iy << 16), ((1u << it.cell.level) | it.cell.ix) | ((it.cell.iy >> 16) << 24));
        if (sp < (int)kSplatStackLds) {
            lds_stack[(2 * sp) * kBlock] = w.x;
            lds_stack[(2 * sp + 1) * kBlock] = w.y;
        } else
            deep[sp - (int)kSplatStackLds] = w;
        sp++;
    };
    auto pop = [&]() {
        --sp;
        const uint2 w = sp < (int)kSplatStackLds ? make_uint2(lds_stack[(2 * sp) * kBlock], lds_stack[(2 * sp + 1) * kBlock]) : deep[sp - (int)kSplatStackLds];
        const uint32_t sx = w.y & 0xffffffu, level = 31u - (uint32_t)__clz((int)sx);
        return Item{w.x & 0xffffu, Cell{sx ^ (1u << level), (w.x >> 16) | ((w.y >> 24) << 16), level}};
    };
    Box2 fp{{0.0f, 0.0f}, {0.0f, 0.0f}};
    float fp_area = 0.0f;
    if (active) {
        unsafeAtomicAdd(t.w_private + (size_t)(blockIdx.x % kWeightCopies) * t.n_leaves + leaf_id, 1.0f);
        // DTree::leaf (dtree.rs:81-94): the cell that holds p
        Cell cell{0, 0, 0};
        uint32_t ni = 0;
        for (;;) {
            Cell child;
            const int c = cell_child_of(cell, p, child);
            const uint32_t link = child_link(t.c_child[leaf.y + ni], c);
            cell = child;
            if (link == 0 || child.level >= kMaxQuadLevel) break;
            ni = link;
        }
        const Box2 lb = cell_box(cell);
        fp = box2_around(p, box2_extent(lb));
        fp = Box2{clamp2(V2{0.0f, 0.0f}, fp.lo, fp.hi), clamp2(V2{1.0f, 1.0f}, fp.lo, fp.hi)};  // Bounds2::clamp (bounds.rs:46-50)
        fp_area = box2_area(fp);
        push(Item{0, Cell{0, 0, 0}});
    }
    uint32_t n_pend = 0;
    // depth-first over (node, cell) pairs; children visited in index order like the recursion
    for (;;) {
        while (sp > 0 && n_pend < kSplatPending) {
            const Item it = pop();
            const uint32_t node_index = leaf.y + it.node;
            const uint2 links = t.c_child[node_index];
            // The four children share three x and three y coordinates, so the overlap with the footprint is two 1-D
            // extents per axis; the values are those of Bounds2::overlap on cell_box(child) (same operands, same order).
            const float cs = u2f((127u - (it.cell.level + 1)) << 23);  // child cell size
            const float x0 = (float)(2 * it.cell.ix) * cs, xm = (float)(2 * it.cell.ix + 1) * cs, x1 = (float)(2 * it.cell.ix + 2) * cs;
            const float y0 = (float)(2 * it.cell.iy) * cs, ym = (float)(2 * it.cell.iy + 1) * cs, y1 = (float)(2 * it.cell.iy + 2) * cs;
            const float ex[2] = {fminf(fp.hi.x, xm) - fmaxf(fp.lo.x, x0), fminf(fp.hi.x, x1) - fmaxf(fp.lo.x, xm)};
            const float ey[2] = {fminf(fp.hi.y, ym) - fmaxf(fp.lo.y, y0), fminf(fp.hi.y, y1) - fmaxf(fp.lo.y, ym)};
            float share[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            bool any = false;
#pragma unroll
            for (int c = 3; c >= 0; c--) {  // pushed in reverse so that child 0 is expanded first
                const float ox = ex[c & 1], oy = ey[c >> 1];
                if (ox <= 0.0f || oy <= 0.0f) continue;  // overlap 0 => weight 0 (or 0/0): skipped by the reference too
                const float w = (ox * oy) / fp_area;
                if (w == 0.0f || !(w == w)) continue;
                const uint32_t link = child_link(links, c);
                const uint32_t child_level = it.cell.level + 1;
                if (link == 0 || child_level >= kMaxQuadLevel) {
                    share[c] = radiance * w;
                    any = true;
                } else if (sp < (int)(3 * kMaxQuadLevel + 4))
                    push(Item{link, Cell{2 * it.cell.ix + (uint32_t)(c & 1), 2 * it.cell.iy + (uint32_t)(c >> 1), child_level}});
            }
            if (any) {
                pend[(5 * n_pend) * kBlock] = node_index;
#pragma unroll
                for (int c = 0; c < 4; c++) pend[(5 * n_pend + 1 + c) * kBlock] = f2u(share[c]);
                n_pend++;
            }
        }
        splat_flush(t, pend, n_pend);  // whole wave
        if (__ballot(sp > 0) == 0ull) break;
    }
}

// STree::record with the stochastic spatial filter (stree.rs:102-116,186-201); the caller has checked that
// `radiance` is finite and non-zero and has drawn the three jitter numbers.
PT_D void stree_record_jittered(const SdTreeDev& t, V3 p, V3 wi, float radiance, float u0, float u1, float u2, bool active, uint32_t* pend) {
    V2 p2 = V2{0.0f, 0.0f};
    uint32_t leaf_id = 0;
    uint4 leaf = make_uint4(0, 0, 0, 0);
    if (active) {
        p2 = dir_to_square(wi);
        V3 lo, hi;
        (void)stree_leaf<true>(t, p, lo, hi);
        const V3 ext = hi - lo;
        const V3 jlo = p - ext / 2.0f;  // Bounds3::around(p, extent)
        const V3 jhi = p + ext / 2.0f;
        const V3 q = clamp3(jlo + mk3(u0, u1, u2) * (jhi - jlo), t.blo, t.bhi);
        leaf_id = stree_leaf<false>(t, q, lo, hi);
        leaf = t.leaf_info[leaf_id];
    }
    dtree_record_box(t, leaf_id, leaf, p2, radiance, active, pend);  // wave-cooperative
}

struct GuidedParams {
    VertexStore vtx;
    PathTail tail;
    uint32_t n_rows;       // max_depth + 1 vertex rows
    uint32_t row_stride;   // = capacity
    uint32_t write_film;   // only the last wave reaches the image (film.clear() per wave, mod.rs:80)
    uint32_t record;       // 0: the scan only folds the paths into the film (ptgpu_set_final_training(0), last wave)
    uint32_t shard_index, shard_count;
    uint32_t tiles_x, n_tiles;
    uint32_t spatial_filter;       // STree filter: 2 = Stochastic (stree.rs:186-201: three draws per record), 0 = Nearest (stree.rs:117-126), 1 = Box (stree.rs:127-185: k_guide_splat_spatial_box)
    uint32_t directional_filter;   // DTree filter: 1 = Box (dtree.rs:235-289), 0 = Nearest (dtree.rs:208-222), 2 = Stochastic (dtree.rs:223-234: two more draws per record)
};

PT_HD uint32_t tile_of_pixel(uint32_t pixel, uint32_t resx, uint32_t tiles_x) { return (pixel / resx / 16u) * tiles_x + (pixel % resx) / 16u; }
// Index of the pixel inside its 16x16 film tile in TileIter order (camera.rs:285-299) — what the reference's
// tile-sequential wavelength recurrence depends on.
PT_HD uint32_t index_in_tile(uint32_t pixel, uint32_t resx) {
    const uint32_t x = pixel % resx, y = pixel / resx;
    const uint32_t tx0 = (x / 16u) * 16u;
    const uint32_t tw = resx - tx0 < 16u ? resx - tx0 : 16u;
    return (x - tx0) + (y % 16u) * tw;
}

// Per-pass head of PathGuidingIntegrator::render (mod.rs:90-96). Tile t of pass k belongs to shard (t + k) %
// shard_count, so waves of 1, 2, 4 passes still spread over all GPUs. The launch enumerates only the (pass, tile,
// pixel-in-tile) triples this shard owns — tiles first_tile(pass) + j * shard_count — so an N-GPU run does 1/N of the
// work here too; paths come out tile by tile (coherent first bounces).
__global__ __launch_bounds__(kBlock) void k_raygen_guided(SceneDev sc, CameraDev cam, RenderParams rp, GuidedParams gp, PathCols st, RoundCounters* cnt,
                                                          uint32_t n_pairs, uint32_t tiles_per_pass, uint64_t first_pass, uint32_t dense) {
    // dense: the host has checked that every enumerated triple is a pixel of this shard (image a whole number of tiles,
    // tiles a whole number of rounds of the shards), so path g takes slot g and nobody queues at the slot counter — one
    // returning atomic per workgroup trip on a single word (~80 per microsecond) was three quarters of this kernel's time.
    __shared__ BlockAlloc sm_alloc;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n_pairs + stride - 1) / stride;
    const uint32_t per_pass = tiles_per_pass * 256u;
    for (uint32_t it = 0; it < n_iter; it++) {
        const uint32_t g = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool mine = false;
        uint32_t pixel = 0;
        uint64_t pass = 0;
        if (g < n_pairs) {
            pass = first_pass + g / per_pass;
            const uint32_t r = g % per_pass, j = r / 256u, q = r % 256u;
            const uint32_t first_tile = (gp.shard_index + gp.shard_count - (uint32_t)(pass % gp.shard_count)) % gp.shard_count;
            const uint32_t tile = first_tile + j * gp.shard_count;
            const uint32_t x = (tile % gp.tiles_x) * 16u + q % 16u, y = (tile / gp.tiles_x) * 16u + q / 16u;
            mine = tile < gp.n_tiles && x < cam.resx && y < cam.resy;
            pixel = y * cam.resx + x;
        }
        uint32_t slot = g, unused_slot;
        if (!dense) block_alloc2(sm_alloc, &cnt[0], mine, false, slot, unused_slot);
        else if (g == 0) cnt[0].n_active = n_pairs;
        if (mine) {
            const uint32_t ix = pixel % cam.resx, iy = pixel / cam.resx;
            Rng rng = rng_init_pixel(ix, iy, pass);
            const float u0 = rng_f32(rng), u1 = rng_f32(rng);
            V3 o, d;
            camera_sample_we(cam, ix, iy, u0, u1, o, d);
            const uint32_t lam = index_in_tile(pixel, cam.resx) + (pass == 0 ? 0u : 256u);
            st.o_pdf[slot] = make_float4(o.x, o.y, o.z, u2f(slot));  // w: path id (vertex-store column); the previous vertex' point
                                                                     // and pdf, which `path` keeps here, are read back from its line
            st.d_flags[slot] = make_float4(d.x, d.y, d.z, u2f(FLAG_FRESH));  // throughput 1: k_shade_guided knows
            st.rng_px[slot] = make_uint4((uint32_t)rng.state, (uint32_t)(rng.state >> 32), pixel, lam);  // w: lambda index | nverts << 16
        }
    }
}

// A stored vertex leaves the kernel as ONE 128-byte line written by eight neighbouring lanes. Compaction scatters paths
// over lanes, so a lane that writes its own line in 16-byte pieces touches a line of its own with every store
// instruction (64 partly written lines per instruction, six instructions per vertex): measured, those stores were 50 ms
// of the 166 ms the kernel takes per step. Instead every lane parks its line in LDS — the emitted light and the
// direct-light sample as soon as they are known (they would otherwise hold 16 registers over the long sampling code),
// position, direction and throughput at the end, in the space the wave's pdf factors have vacated by then — and the wave
// then writes the parked lines out, eight lanes per line: full lines, one request each, nothing for the memory side
// to merge. The direct-light sample is written as if visible; k_occlude_guided clears it when the shadow ray is blocked.
// The emitted light of a vertex is zero unless the path has just hit an emitter, its direct light is zero when no light
// sample was made: the record pass adds each only when present (`has_emitted`, `has_li`: bits 0 and 1 of the line's
// spare word); an absent field of the line holds whatever the LDS held.
// Piece k (VTX_*) of lane l sits at [k][(l + k) % 64] of the wave's block: lanes write consecutive 16-byte cells, and
// the eight lanes that later read pieces 0..7 of one line fall into eight different bank groups.
constexpr uint32_t kNoLine = 0xffffffffu;
PT_D float4* stage_cell(float4* lo, float4* hi, uint32_t piece, uint32_t lane) { return (piece < 4u ? lo : hi - 4u * 64u) + piece * 64u + ((lane + piece) & 63u); }
PT_D void stage_vertex_light(float4* hi, uint32_t lane, const Spec& e, const Spec& li, bool has_emitted, bool has_li) {
    if (has_emitted) {
        *stage_cell(nullptr, hi, VTX_E0, lane) = make_float4(e.v[0], e.v[1], e.v[2], e.v[3]);
        *stage_cell(nullptr, hi, VTX_E1, lane) = make_float4(e.v[4], e.v[5], e.v[6], e.v[7]);
    }
    if (has_li) {
        *stage_cell(nullptr, hi, VTX_LI0, lane) = make_float4(li.v[0], li.v[1], li.v[2], li.v[3]);
        *stage_cell(nullptr, hi, VTX_LI1, lane) = make_float4(li.v[4], li.v[5], li.v[6], li.v[7]);
    }
}
PT_D void stage_vertex(float4* lo, uint32_t lane, V3 p, V3 wi, float pdf, const Spec& f, bool has_emitted, bool has_li) {
    *stage_cell(lo, nullptr, VTX_P_PDF, lane) = make_float4(p.x, p.y, p.z, pdf);
    *stage_cell(lo, nullptr, VTX_WI, lane) = make_float4(wi.x, wi.y, wi.z, u2f((has_emitted ? 1u : 0u) | (has_li ? 2u : 0u)));
    *stage_cell(lo, nullptr, VTX_F0, lane) = make_float4(f.v[0], f.v[1], f.v[2], f.v[3]);
    *stage_cell(lo, nullptr, VTX_F1, lane) = make_float4(f.v[4], f.v[5], f.v[6], f.v[7]);
}
// Called by the whole wave: `line` = index of this lane's parked line in the vertex store, or kNoLine.
PT_D void stage_write_out(float4* lines, float4* lo, float4* hi, uint32_t line) {
    asm volatile("" ::: "memory");  // the parking stores above stay above
    const unsigned long long have = __ballot(line != kNoLine);
    if (have == 0ull) return;
    const uint32_t lane = __lane_id(), piece = lane & 7u;
#pragma unroll
    for (uint32_t s = 0; s < 8; s++) {
        if (((have >> (8u * s)) & 0xffull) == 0ull) continue;  // wave-uniform
        const uint32_t src = 8u * s + (lane >> 3);
        const uint32_t at = (uint32_t)__shfl((int)line, (int)src);
        if (at != kNoLine) lines[(size_t)at * 8u + piece] = *stage_cell(lo, hi, piece, src);
    }
    asm volatile("" ::: "memory");  // ... and the block's next use below
}

// One iteration of PathGuidingIntegrator::random_walk's loop body (mod.rs:139-278).
template <int KINDS, bool SMALL>
__global__ __launch_bounds__(kBlock, KINDS != 7 ? PT_GUIDED_DIFFUSE_WAVES : PT_SHADE_WAVES) void k_shade_guided(SceneDev sc, SdTreeDev tree, RenderParams rp, GuidedParams gp, PathCols in, HitCols hit, PathCols out,
                                                         ShadowCols sh, const RoundCounters* cur, RoundCounters* nxt, const uint32_t* cur_top, uint32_t* nxt_top) {
    __shared__ BlockAlloc3 sm_alloc;
    __shared__ uint32_t sm_count;
    __shared__ BlockSort sm_sort;
    constexpr bool kSort = PT_SORT_SHADE_GUIDED && KINDS == 7;  // measured: a gain where conductors mix with other surfaces (veach-mis shade -9 %, veach-bidir -7 %), none or a loss on the Cornell box
    __shared__ __align__(16) float sm_scales[(kBlock / 64u) * kScaleWaveFloats];
    __shared__ float4 sm_stage_hi[(kBlock / 64u) * 4u * 64u];
    __shared__ float sm_shadow[6 * kBlock];
    static_assert(kScaleWaveFloats >= 4u * 64u * 4u && (kScaleWaveFloats * sizeof(float)) % 16u == 0, "pieces 0-3 of a wave's lines are parked in its block of pdf factors");
    // (recomputed where used rather than kept: the kernel sits at its register cap)
#define PT_MY_SCALES scales_of(sm_scales, threadIdx.x)
#define PT_STAGE_LO reinterpret_cast<float4*>(sm_scales + (threadIdx.x / 64u) * kScaleWaveFloats)
#define PT_STAGE_HI (sm_stage_hi + (threadIdx.x / 64u) * 4u * 64u)
    PT_TICK_INIT();
    __shared__ SmallSceneLds<SMALL> sm_scene;
    if constexpr (SMALL) stage_small_scene(sc, sm_scene);
    uint32_t n_shaded = 0, n_skipped = 0;
    const uint32_t n_bottom = cur->n_active;
    const uint32_t n = n_bottom + *cur_top;
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t n_iter = (n + stride - 1) / stride;
    for (uint32_t it = 0; it < n_iter; it++) {
        uint32_t j = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        if (kSort) {  // lanes regrouped by BSDF code (pt_kernels.h: block_sort_by_class)
            uint32_t cls = 3;
            if (j < n) {
                const uint2 pt = hit.prim[slot_of_index(j, n_bottom, rp.capacity)];
                if (pt.x != kNoHit && tag_material(pt.y) >= 0) cls = tag_kind(pt.y);
            }
            j = it * stride + blockIdx.x * blockDim.x + block_sort_by_class(sm_sort, cls);
        }
        const bool valid = j < n;
        const uint32_t i = slot_of_index(j, n_bottom, rp.capacity);
        bool alive = false, want_shadow = false, shaded = false, skipped_shadow = false, ended = false;
        V3 o = mk3(0, 0, 0), d = mk3(0, 0, 1);
        uint32_t flags = 0, pid = 0, nverts = 0;
        bool stored = false, miss_vertex = false;  // stored: this lane fills vertex row nverts - 1 of its path in this trip
        V3 park_p = mk3(0, 0, 0), park_wi = mk3(0, 0, 1);
        float park_pdf = 0.0f;
        Spec park_f = spec_splat(0.0f);
        uint32_t park_bits = 0;
        Spec beta = spec_splat(0.0f);
        uint4 rp4 = make_uint4(0, 0, 0, 0);
        if (valid) {
            const float4 o4 = in.o_pdf[i], d4 = in.d_flags[i];
            o = mk3(o4.x, o4.y, o4.z);
            d = mk3(d4.x, d4.y, d4.z);
            flags = f2u(d4.w);
            const bool fresh = (flags & FLAG_FRESH) != 0u;  // straight out of k_raygen_guided: throughput 1, not stored
            flags &= ~FLAG_FRESH;
            rp4 = in.rng_px[i];
            pid = f2u(o4.w);
            nverts = rp4.w >> 16;
            const uint2 prim_tag = hit.prim[i];
            const uint32_t prim = prim_tag.x;
            PT_TICK(0);
            if (prim == kNoHit) {
                // miss vertex (mod.rs:140-150): far point along the ray, nothing emitted without an envmap
                if (nverts < gp.n_rows) {
                    miss_vertex = stored = true;  // parked below, once every lane of the wave is done with its pdf factors
                    nverts++;
                } else
                    atomicAdd(sc.guards + kGuardVertexStore, 1u);
                ended = true;
                PT_TICK(12);
            } else {
                const float4 tb = hit.tb[i];
                beta = fresh ? spec_splat(1.0f) : load_spec(in.beta0, in.beta1, i);
                const uint32_t row = rp4.w & 0xffffu;  // wavelength-table row of this path
                Surface sf;
                if constexpr (SMALL) sf = make_surface_from(sc, sm_scene.tri, prim, tb.y, tb.z, tb.w);
                else sf = make_surface(sc, prim, tb.y, tb.z, tb.w);
                const int material = tag_material(prim_tag.y);
                uint32_t depth = flags & FLAG_DEPTH_MASK;
                if (material < 0) {
                    // pass-through: no bounce counted and, unlike `path`, specular_bounce is left alone (mod.rs:153-156)
                    o = offset_ray_origin(sf.blo, sf.bhi, sf.ng, d);
                    alive = passthrough_step(flags);
                    if (!alive) atomicAdd(sc.guards + kGuardPassThrough, 1u);
                    ended = !alive;
                } else {
                    shaded = true;
                    flags &= ~FLAG_PASS_MASK;
                    const V3 wo = -d;
                    const V3 p = surf_p(sf);
                    Bsdf bsdf;
                    if constexpr (SMALL) bsdf = make_bsdf_head<KINDS>(rp.spec, sm_scene.mat[material], material, row);
                    else bsdf = make_bsdf<KINDS>(sc, rp.spec, material, row);
                    const bool specular = bsdf_is_specular(bsdf);
                    const bool spec_bounce = (flags & FLAG_SPECULAR_BOUNCE) != 0;
                    Spec emitted = spec_splat(0.0f);
                    if (sf.tri.mesh < sc.n_emissive_meshes) {
                        const bool direct = (depth == 0) || spec_bounce;
                        if (rp.mode == MODE_PT || direct) emitted = light_emitted(rp.spec, sf.tri, sf.ng, p, o, row);
                        else if (rp.mode == MODE_MIS) {
                            // previous point and its sampling pdf: the last stored vertex (the previous bounce was not
                            // specular, or this would be `direct`; every non-specular bounce is stored)
                            const float4 pv = nverts > 0 ? gp.vtx.at(nverts - 1, pid)[VTX_P_PDF] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                            const float pb = pv.w;
                            const float pl = light_pdf_li(sf.tri, sf.ng, mk3(pv.x, pv.y, pv.z), p);
                            const float w = (pb * pb) / (pb * pb + pl * pl);
                            emitted = w * light_emitted(rp.spec, sf.tri, sf.ng, p, o, row);
                        }
                    }
                    Rng rng{(uint64_t)rp4.x | ((uint64_t)rp4.y << 32), rng_inc_for_pixel(rp4.z % rp.resx, rp4.z / rp.resx)};
                    PT_TICK(1);
                    // Memory round trips are what this kernel waits for (2-3 waves per SIMD, ~80 % of the time in
                    // s_waitcnt), so the loads of the two independent chains below — light pick -> its emission row,
                    // jump table -> first quadtree node — are issued pairwise before either result is used, and
                    // before the vertex stores (on gfx950 a wave cannot reuse a store's data registers, nor see a
                    // later load, until the store is acknowledged).
                    const bool guided_lane = tree.iteration != 0 && !specular;
                    uint2 top_entry = make_uint2(0u, 0u);
                    if (guided_lane) top_entry = tree.top[stree_code(tree, p)];
                    bool has_nee = false, have_light = false;
                    Spec contrib = spec_splat(0.0f);
                    Tri lt;
                    float pmf = 0.0f, ua = 0.0f, ub = 0.0f;
                    if (rp.mode != MODE_PT && !specular) {
                        const float u1d = rng_f32(rng);
                        ua = rng_f32(rng);
                        ub = rng_f32(rng);
                        if constexpr (SMALL) have_light = choose_light_from(sc, sm_scene.light, u1d, lt, pmf);
                        else have_light = choose_light(sc, u1d, lt, pmf);
                    }
                    Spec le = spec_splat(0.0f);
                    if (have_light) le = spec_lookup(rp.spec, rp.spec.emission_base + lt.mesh, row);
                    PT_TICK(13);
                    uint32_t sampler_word = 0x80000000u;
                    SNode root;
                    root.rad = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    root.child = make_uint2(0u, 0u);
                    if (guided_lane) {
                        sampler_word = stree_resolve(tree, top_entry, p);
#if PT_TWO_LEVEL
                        root.rad = tree.s_node2[sampler_word & 0x7fffffffu].rad;
                        root.child = tree.s_node2[sampler_word & 0x7fffffffu].child;
#else
                        root = tree.s_node[sampler_word & 0x7fffffffu];
#endif
                    }
                    PT_TICK(3);
                    if (have_light) {
                        LiSample ls;
                        if (light_sample_li(lt, p, ua, ub, ls)) {
                            Spec f = bsdf_eval(bsdf, sf.frame, sf.ng, wo, ls.wl, true) / (ls.pdf * pmf);
                            if (!spec_is_black(f)) {
                                const V3 lp = (ls.blo + ls.bhi) / 2.0f;
                                if (rp.mode == MODE_NEE) contrib = f * le;
                                else {
                                    const float pl = ls.pdf;
                                    const float pb = bsdf_pdf(bsdf, sf.frame, wo, ls.wl);
                                    const float w = (pl * pl) / (pl * pl + pb * pb);
                                    contrib = w * f * le;
                                }
                                const V3 w = normalize(lp - p);
                                // the shadow ray waits in LDS for its queue slot, not in six registers over the sampling code
                                const V3 sh_from = offset_ray_origin(sf.blo, sf.bhi, sf.ng, w);
                                sm_shadow[0 * kBlock + threadIdx.x] = sh_from.x;
                                sm_shadow[1 * kBlock + threadIdx.x] = sh_from.y;
                                sm_shadow[2 * kBlock + threadIdx.x] = sh_from.z;
                                sm_shadow[3 * kBlock + threadIdx.x] = lp.x;
                                sm_shadow[4 * kBlock + threadIdx.x] = lp.y;
                                sm_shadow[5 * kBlock + threadIdx.x] = lp.z;
                                has_nee = true;
                            }
                        }
                    }
                    const float off_mag = offset_magnitude(sf.blo, sf.bhi, sf.ng);  // what the next ray's origin needs of the error box
                    const bool has_emitted = !spec_is_black(emitted);
                    if (!specular && nverts < gp.n_rows) stage_vertex_light(PT_STAGE_HI, __lane_id(), emitted, contrib, has_emitted, has_nee);
                    PT_TICK(2);
                    BsdfSample bs;
                    const bool ok = guider_sample(tree, bsdf, sf.frame, sf.ng, wo, rng, PT_MY_SCALES, sampler_word, root, bs PT_DBG_ARGS);
                    rp4.x = (uint32_t)rng.state;
                    rp4.y = (uint32_t)(rng.state >> 32);
                    if (!ok) {
                        // QUIRK kept: the walk breaks before the vertex is pushed, so this surface's emitted light
                        // and its direct-light sample are lost (mod.rs:245-266). The reference still traced the
                        // shadow ray; here it is only counted.
                        skipped_shadow = has_nee;
                        ended = true;
                    } else {
                        flags = specular ? (flags | FLAG_SPECULAR_BOUNCE) : (flags & ~FLAG_SPECULAR_BOUNCE);
                        if (specular && dot(bs.wi, sf.ng) * dot(wo, sf.ng) < 0.0f) flags ^= FLAG_INSIDE;  // went through the surface (queueing hint only)
                        Spec fw;
#pragma unroll
                        for (int k = 0; k < NSW; k++) fw.v[k] = bs.f.v[k] / bs.pdf;
                        if (!specular && nverts < gp.n_rows) {
                            // pieces 0-3 of the line share LDS with the wave's pdf factors: they are parked below, at the
                            // wave-uniform point behind this divergent block, where every lane is done with its factors
                            park_p = p;
                            park_wi = bs.wi;
                            park_pdf = bs.pdf;
                            park_f = fw;
                            park_bits = (has_emitted ? 1u : 0u) | (has_nee ? 2u : 0u);
                            stored = true;
                            want_shadow = has_nee;
                            nverts++;
                        } else {
                            skipped_shadow = has_nee;
                            if (!specular) atomicAdd(sc.guards + kGuardVertexStore, 1u);
                        }
                        beta = beta * fw;
                        if (spec_is_black(beta)) ended = true;
                        else {
                            if (bs.terminate_secondary) flags |= FLAG_SECONDARY_TERMINATED;
                            o = offset_ray_origin_from(p, sf.ng, off_mag, bs.wi);
                            d = bs.wi;
                            depth += 1;
                            flags = (flags & ~FLAG_DEPTH_MASK) | (depth & FLAG_DEPTH_MASK);
                            alive = depth < rp.max_depth;
                            ended = !alive;
                        }
                    }
                    PT_TICK(18);
                }
            }
            rp4.w = (rp4.w & 0xffffu) | (nverts << 16);
        }
        // Every lane of the wave is here (the block above is closed, and what follows is wave-wide ballots and shuffles):
        // no lane can still be reading its column of pdf factors when another lane's line pieces land on it.
        __builtin_amdgcn_wave_barrier();
        if (miss_vertex) stage_vertex(PT_STAGE_LO, __lane_id(), o + d * 10000.0f, d, 1.0f, spec_splat(1.0f), false, false);
        else if (stored) stage_vertex(PT_STAGE_LO, __lane_id(), park_p, park_wi, park_pdf, park_f, (park_bits & 1u) != 0, (park_bits & 2u) != 0);
        stage_write_out(gp.vtx.lines, PT_STAGE_LO, PT_STAGE_HI, stored ? (nverts - 1u) * gp.vtx.n_paths + pid : kNoLine);
        PT_TICK(8);
        uint32_t a_slot, s_slot;
        const bool to_top = alive && (flags & FLAG_INSIDE) != 0;  // the next surface is, in all likelihood, the body's own: specular
#ifdef PT_DBG_TIMING
        lds_barrier();
        PT_TICK(19);  // waiting for the block's other waves to arrive
#endif
        block_alloc3(sm_alloc, nxt, nxt_top, rp.capacity, alive && !to_top, to_top, want_shadow, a_slot, s_slot);
        PT_TICK(9);
        n_shaded += shaded ? 1u : 0u;
        n_skipped += skipped_shadow ? 1u : 0u;
        if (alive) {
            out.o_pdf[a_slot] = make_float4(o.x, o.y, o.z, u2f(pid));
            out.d_flags[a_slot] = make_float4(d.x, d.y, d.z, u2f(flags));
            store_spec(out.beta0, out.beta1, a_slot, beta);
            out.rng_px[a_slot] = rp4;
        }
        if (valid && ended) {
            gp.tail.rng_px[pid] = rp4;
            gp.tail.nverts_flags[pid] = flags;
        }
        if (want_shadow) {
            sh.q0[s_slot] = make_float4(sm_shadow[0 * kBlock + threadIdx.x], sm_shadow[1 * kBlock + threadIdx.x], sm_shadow[2 * kBlock + threadIdx.x], u2f(pid));
            sh.q1[s_slot] = make_float4(sm_shadow[3 * kBlock + threadIdx.x], sm_shadow[4 * kBlock + threadIdx.x], sm_shadow[5 * kBlock + threadIdx.x], u2f(nverts - 1u));  // the row just stored
        }
        PT_TICK(10);
    }
    block_count_flush(&sm_count, &nxt->n_segments, n_shaded);
    block_count_flush(&sm_count, &nxt->n_terminal, n_skipped);
    PT_TICK(11);
    PT_TICK_FLUSH();
}

// NEE visibility for the guided walk: shade wrote the sample into the vertex as if visible; a blocked one is cleared.
template <bool CENSUS, bool WIDE>
__global__ __launch_bounds__(kBlock) void k_occlude_guided(SceneDev sc, GuidedParams gp, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;  // late rounds: most workgroups have nothing to do, not even staging nodes
    bvh_stage_nodes<WIDE>(sc);
    const uint32_t stride = gridDim.x * blockDim.x;
    uint32_t cn = 0, ct = 0;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        const float4 a = sh.q0[j], b = sh.q1[j];
        if (scene_occluded<CENSUS, WIDE>(sc, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), &cn, &ct)) {
            float4* v = gp.vtx.at(f2u(b.w), f2u(a.w));
            v[VTX_LI0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            v[VTX_LI1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    if (CENSUS) census_add(census, 2, cn, ct);
}

// The same over the shadow queue as a stream (bvh_stream_wide), for the four-wide walk of large scenes.
template <bool CENSUS>
__global__ __launch_bounds__(kBlock, PT_STREAM_WAVES) void k_occlude_guided_stream(SceneDev sc, GuidedParams gp, ShadowCols sh, const RoundCounters* cnt, unsigned long long* census) {
    const uint32_t n = cnt->n_shadow;
    if (blockIdx.x * blockDim.x >= n) return;
    bvh_stage_nodes<true>(sc);
    uint32_t cn = 0, ct = 0;
    const auto fetch = [&](uint32_t j, V3& o, V3& d) {
        const float4 a = sh.q0[j], q = sh.q1[j];
        o = mk3(a.x, a.y, a.z);
        d = mk3(q.x, q.y, q.z) - o;  // Scene::occluded: unnormalised segment, t_max 0.9999
    };
    const auto emit = [&](bool done, uint32_t j, const Hit& h) {
        if (!done || h.prim == kNoHit) return;
        float4* v = gp.vtx.at(f2u(sh.q1[j].w), f2u(sh.q0[j].w));
        v[VTX_LI0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        v[VTX_LI1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    if (sc.lds_wnodes) bvh_stream_wide<CENSUS, true, true>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    else bvh_stream_wide<CENSUS, true, false>(sc, n, 0.9999f, fetch, emit, &cn, &ct);
    if (CENSUS) census_add(census, 2, cn, ct);
}

// The two walkers of the guided integrator in ONE launch: the shadow rays of round r and the bounce rays of round r + 1
// both exist once shade(r) has run and do not depend on each other. Late rounds and small waves (one GPU's share of a
// multi-GPU job) are bound by launch count and by the tail of each launch, not by work: two launches per round instead of
// three, and either walker's tail is filled by the other's workgroups. The grid is divided between the two roles in
// proportion to their work (a bounce ray costs about twice a shadow ray); every workgroup derives the split from the
// round's counters, so it needs no host look.
#ifndef PT_WALK_WAVES
#define PT_WALK_WAVES 7  // 8 (64 VGPRs) spills 28 B with both walkers' arguments live
#endif
template <bool CENSUS, bool WIDE>
__global__ __launch_bounds__(kBlock, WIDE ? PT_WIDE_WAVES : PT_WALK_WAVES) void k_walk_guided(SceneDev sc, GuidedParams gp, PathCols st, HitCols hit, ShadowCols sh, const RoundCounters* cnt,
                                                                               const uint32_t* top, uint32_t capacity, unsigned long long* census) {
    const uint32_t n_bottom = cnt->n_active, n_ext = n_bottom + *top, n_sh = cnt->n_shadow;
    const uint32_t G = gridDim.x;
    uint32_t g_ext = 0;
    if (n_ext != 0 && n_sh == 0) g_ext = G;
    else if (n_ext != 0) {
        const float share = 2.0f * (float)n_ext / (2.0f * (float)n_ext + (float)n_sh);
        g_ext = (uint32_t)(share * (float)G + 0.5f);
        g_ext = G > 1 ? (g_ext < 1u ? 1u : (g_ext > G - 1u ? G - 1u : g_ext)) : 1u;  // (a grid of one workgroup does both, below)
    }
    uint32_t cn = 0, ct = 0;
    const bool both = G == 1 && n_ext != 0 && n_sh != 0;
    const bool extend_role = blockIdx.x < g_ext;
    if (extend_role || both) {
        const uint32_t b = blockIdx.x, stride = g_ext * blockDim.x;
        if (b * blockDim.x < n_ext) {
            bvh_stage_nodes<WIDE>(sc);
            for (uint32_t j = b * blockDim.x + threadIdx.x; j < n_ext; j += stride) {
                const uint32_t i = slot_of_index(j, n_bottom, capacity);
                float4 o = st.o_pdf[i], d = st.d_flags[i];
                asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(d.x), "+v"(d.y), "+v"(d.z));  // one load each (see bvh_intersect)
                const Hit h = bvh_intersect<CENSUS, false, WIDE>(sc, mk3(o.x, o.y, o.z), mk3(d.x, d.y, d.z), f_inf(), &cn, &ct);
                hit.tb[i] = make_float4(h.th.t, h.th.b0, h.th.b1, h.th.b2);
                hit.prim[i] = make_uint2(h.prim, h.tag);
            }
            if (CENSUS) census_add(census, 0, cn, ct);
        }
        if (!both) return;
        cn = ct = 0;
        __syncthreads();
    }
    {
        const uint32_t g_occ = both ? 1u : G - g_ext, b = both ? 0u : blockIdx.x - g_ext, stride = g_occ * blockDim.x;
        if (b * blockDim.x >= n_sh) return;
        bvh_stage_nodes<WIDE>(sc);
        for (uint32_t j = b * blockDim.x + threadIdx.x; j < n_sh; j += stride) {
            const float4 a = sh.q0[j], q = sh.q1[j];
            if (scene_occluded<CENSUS, WIDE>(sc, mk3(a.x, a.y, a.z), mk3(q.x, q.y, q.z), &cn, &ct)) {
                float4* v = gp.vtx.at(f2u(q.w), f2u(a.w));
                v[VTX_LI0] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                v[VTX_LI1] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
        if (CENSUS) census_add(census, 2, cn, ct);
    }
}

// Back-to-front pass over each finished path (mod.rs:106-116): what arrives at a vertex from further along the path
// is recorded into the SD-tree, then folded through the vertex. Two kernels:
//   k_guide_scan  — one lane per path walks its vertices back to front (row v of all paths is contiguous), draws the
//                   three jitter numbers of every record that will happen (STree::record draws only for a finite,
//                   non-zero radiance, so the PCG sequence is the reference's) and appends {p, radiance, wi, jitter}
//                   to the record queue; it also adds the finished path to the film.
//   k_guide_splat — one lane per queued record does the tree work (two spatial descents, the quadtree box splat).
// Splitting keeps the divergent, latency-bound tree work on dense lanes. The queue needs no global atomics: chunk c
// (path ids [256c, 256c+256)) owns entries [c*256*n_rows, ...) and its length lands in count[c].
struct RecordQueue {
    float4* q0;       // p.xyz, radiance
    float4* q1;       // wi.xyz, u0
    float2* q2;       // u1, u2
    uint32_t* count;  // records per chunk
    uint4* who;       // parity hook only (ptgpu_guided_capture_records), else NULL: {path id, vertex row, pixel, -} per record
    float2* q3;       // the directional stochastic filter's two draws (dtree.rs:232), else NULL
};

__global__ __launch_bounds__(kBlock, PT_SCAN_WAVES) void k_guide_scan(SceneDev sc, RenderParams rp, GuidedParams gp, RecordQueue rq, uint32_t n_paths, uint32_t first_chunk, uint32_t end_chunk) {
    __shared__ uint32_t sm_n;
    __shared__ float4 sm_stage[(kBlock / 64u) * 8u * 64u];  // per wave: the 64 lines of one row, on their way from the lanes that fetched them to their paths
    const uint32_t lane = __lane_id();
    for (uint32_t chunk = first_chunk + blockIdx.x; chunk < end_chunk; chunk += gridDim.x) {  // (chunks [first, end): the whole batch, or one slice of it)
        if (threadIdx.x == 0) sm_n = 0;
        __syncthreads();
        const uint32_t pid = chunk * kBlock + threadIdx.x;
        const bool valid = pid < n_paths;
        uint4 rp4 = make_uint4(0, 0, 0, 0);
        uint32_t flags = 0;
        if (valid) {
            rp4 = gp.tail.rng_px[pid];
            flags = gp.tail.nverts_flags[pid];
        }
        const uint32_t nverts = rp4.w >> 16;
        uint32_t wave_rows = nverts;
        for (int off = 32; off > 0; off >>= 1) wave_rows = max(wave_rows, (uint32_t)__shfl_xor((int)wave_rows, off));
        Rng rng{(uint64_t)rp4.x | ((uint64_t)rp4.y << 32), rng_inc_for_pixel(rp4.z % rp.resx, rp4.z / rp.resx)};
        Spec l = spec_splat(0.0f);
        const size_t base = (size_t)chunk * kBlock * gp.n_rows;
        // The lines of a path are read back to front, and nothing in a line's address depends on the line before it: the
        // next row is requested whole before this one is folded in (two rows in flight per wave). Whole, because a
        // half asked for one trip later has left L2 by then and comes from HBM a second time (measured: +24 % traffic).
        // A row is fetched as it was written (stage_write_out): eight neighbouring lanes read the eight pieces of one
        // line — a lane that reads its own line in 16-byte pieces keeps the address pipeline busy four cycles per piece
        // (64 lines per instruction), which bounded this kernel at 2.2 TB/s (TA busy 0.9) — and the lines change hands in LDS.
        const uint32_t pid0 = chunk * kBlock + (threadIdx.x & ~63u);  // the wave's first path: lane s holds path pid0 + s
        float4* const stage = sm_stage + (threadIdx.x / 64u) * 8u * 64u;
        float4 nx[8];
        const auto fetch_row = [&](uint32_t row) {
            const unsigned long long has = __ballot(row < nverts);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) {
                const uint32_t src = 8u * k + (lane >> 3);
                nx[k] = make_float4(0, 0, 0, 0);
                if ((has >> src) & 1ull) nx[k] = gp.vtx.at(row, pid0)[64u * k + lane];  // = piece lane % 8 of path pid0 + src: consecutive lanes, consecutive addresses
            }
        };
        if (wave_rows > 0) fetch_row(wave_rows - 1);
        for (uint32_t v = wave_rows; v-- > 0;) {
            bool want = false;
            float4 cur[8];
            asm volatile("" ::: "memory");
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) *stage_cell(stage, stage + 4u * 64u, lane & 7u, 8u * k + (lane >> 3)) = nx[k];
            asm volatile("" ::: "memory");
            if (v > 0) fetch_row(v - 1);
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) cur[k] = *stage_cell(stage, stage + 4u * 64u, k, lane);  // (read where used: the compiler sinks them)
            const float4 pp = cur[VTX_P_PDF], wi = cur[VTX_WI];
            float radiance = 0.0f, u0 = 0.0f, u1 = 0.0f, u2 = 0.0f, d0 = 0.0f, d1 = 0.0f;
            if (v < nverts) {
                radiance = spec_average(l) / pp.w;
                want = isfinite(radiance) && radiance != 0.0f;  // STree::record's early return (stree.rs:102-116)
                if (want && gp.spatial_filter == 2u) {  // the stochastic filter's three draws (stree.rs:195-199); the other filters draw nothing
                    u0 = rng_f32(rng);
                    u1 = rng_f32(rng);
                    u2 = rng_f32(rng);
                }
                if (want && gp.directional_filter == 2u) {  // ... then DTree::record_stochastic's two (dtree.rs:232), by the same sampler
                    d0 = rng_f32(rng);
                    d1 = rng_f32(rng);
                }
                l = l * spec_from(cur[VTX_F0], cur[VTX_F1]);
                if (f2u(wi.w) & 2u) l = l + spec_from(cur[VTX_LI0], cur[VTX_LI1]);  // fields without their bit were never written
                if (f2u(wi.w) & 1u) l = l + spec_from(cur[VTX_E0], cur[VTX_E1]);
            }
            const unsigned long long m = gp.record ? __ballot(want) : 0ull;
            if (m != 0ull) {
                uint32_t wbase = 0;
                if (lane == 0) wbase = atomicAdd(&sm_n, (uint32_t)__popcll(m));
                wbase = __shfl(wbase, 0);
                if (want) {
                    const size_t slot = base + wbase + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    rq.q0[slot] = make_float4(pp.x, pp.y, pp.z, radiance);
                    rq.q1[slot] = make_float4(wi.x, wi.y, wi.z, u0);
                    rq.q2[slot] = make_float2(u1, u2);
                    if (rq.who) rq.who[slot] = make_uint4(pid, v, rp4.z, 0u);
                    if (rq.q3) rq.q3[slot] = make_float2(d0, d1);
                }
            }
        }
        if (gp.write_film) film_add_wave(sc, rp.film, rp.lambda_tab, rp4.w & 0xffffu, rp4.z, l, (flags & FLAG_SECONDARY_TERMINATED) != 0, valid);
        __syncthreads();
        if (threadIdx.x == 0) rq.count[chunk] = sm_n;
    }
}

// STree::leaf by plain node chasing from the root (stree.rs:202-229), as the parity hook's yardstick for stree_leaf.
PT_D uint32_t stree_leaf_plain(const SdTreeDev& t, V3 p, V3& lo, V3& hi) {
    lo = t.blo;
    hi = t.bhi;
    uint32_t ni = 0;
    for (int guard = 0; guard < 128; guard++) {
        const uint4 n = t.snodes[ni];
        if (n.x == 0) return n.w;
        const int ax = (int)n.z;
        const float mid = (get(lo, ax) + get(hi, ax)) / 2.0f;
        if (get(p, ax) < mid) {
            set(hi, ax, mid);
            ni = n.x;
        } else {
            set(lo, ax, mid);
            ni = n.y;
        }
    }
    return t.snodes[ni].w;
}

// Parity hook (ptgpu_guided_selfcheck): fused walk vs the two separate descents it replaces, on the live tree.
__global__ __launch_bounds__(kBlock) void k_guided_selfcheck(SdTreeDev tree, uint32_t n_probes, uint64_t seed, unsigned long long* counts) {
    __shared__ float sm_scales[(kMaxQuadLevel + 1) * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_probes) return;
    Rng rng = rng_init_pixel(i & 0xffffu, i >> 16, seed);
    const V3 ext = tree.bhi - tree.blo;
    const V3 p = tree.blo + mk3(rng_f32(rng), rng_f32(rng), rng_f32(rng)) * ext;
    V3 lo, hi;
    uint32_t word = 0;
    const uint32_t leaf_id = stree_leaf<true>(tree, p, lo, hi, &word);
    {
        // jump-table descent vs node chasing: same leaf, same box bits, and the sampler word is the leaf's
        V3 plo, phi;
        const uint32_t plain = stree_leaf_plain(tree, p, plo, phi);
        const uint4 li = tree.leaf_info[plain];
        const bool same_box = f2u(lo.x) == f2u(plo.x) && f2u(lo.y) == f2u(plo.y) && f2u(lo.z) == f2u(plo.z) && f2u(hi.x) == f2u(phi.x) &&
                              f2u(hi.y) == f2u(phi.y) && f2u(hi.z) == f2u(phi.z);
        if (plain != leaf_id || !same_box || (word & 0x7fffffffu) != li.x || (word >> 31) != li.z) atomicAdd(counts + 4, 1ull);
    }
    const uint4 leaf = make_uint4(word & 0x7fffffffu, 0u, word >> 31, 0u);
    float* scales = scales_of(sm_scales, threadIdx.x);
    // (a) separate descents
    Rng ra = rng;
    const V2 sq_a = dtree_sample(tree, leaf, ra);
    const float pdf_a = dtree_pdf(tree, leaf, sq_a, scales);
    // (b) fused
    Rng rb = rng;
    V2 sq_b = V2{0.0f, 0.0f};
#if PT_TWO_LEVEL
    const float pdf_b = dtree_walk2(tree, leaf, true, rb, sq_b, scales, tree.s_node2[leaf.x].rad, tree.s_node2[leaf.x].child);
#else
    const float pdf_b = dtree_walk(tree, leaf, true, rb, sq_b, scales, tree.s_node[leaf.x]);
#endif
    if (!leaf.z) atomicAdd(counts + 0, 1ull);
    if (f2u(sq_a.x) != f2u(sq_b.x) || f2u(sq_a.y) != f2u(sq_b.y) || ra.state != rb.state) atomicAdd(counts + 1, 1ull);
    if (f2u(pdf_a) != f2u(pdf_b)) atomicAdd(counts + 2, 1ull);
    // pdf-only lanes
    V2 sq_c = V2{rng_f32(rng), rng_f32(rng)};
    const float pdf_c = dtree_pdf(tree, leaf, sq_c, scales);
    Rng rc = rng;
#if PT_TWO_LEVEL
    const float pdf_d = dtree_walk2(tree, leaf, false, rc, sq_c, scales, tree.s_node2[leaf.x].rad, tree.s_node2[leaf.x].child);
#else
    const float pdf_d = dtree_walk(tree, leaf, false, rc, sq_c, scales, tree.s_node[leaf.x]);
#endif
    if (f2u(pdf_c) != f2u(pdf_d) || rc.state != rng.state) atomicAdd(counts + 3, 1ull);
}

// Parity hook (ptgpu_guided_probe): what the shade kernel's lookups return for caller-supplied points, for comparison with
// the CPU oracle's STree::leaf / DTree::sample / DTree::pdf on the same tree. Per probe 16 words:
//   [0] spatial leaf id  [1..2] sampled square point  [3] tree pdf of that point  [4] tree pdf of the query point
//   [5..6] PCG state after the sampling draws  [7] 1 if the shade kernel's split descent (stree_code + stree_resolve)
//   disagrees with stree_leaf  [8..13] the leaf's box (lo, hi)  [14..15] dir_to_square of the point read as a direction
//   (the one libm call, atan2f, on the record path: tests use it to tell last-ulp libm differences from real ones)
__global__ __launch_bounds__(kBlock) void k_guided_probe(SdTreeDev tree, const float* points, const float* query, uint32_t n_probes, uint64_t seed, uint32_t* out) {
    __shared__ float sm_scales[(kMaxQuadLevel + 1) * kBlock];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_probes) return;
    const V3 p = mk3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]);
    V3 lo, hi;
    uint32_t word = 0;
    const uint32_t leaf_id = stree_leaf<true>(tree, p, lo, hi, &word);
    const uint32_t word2 = stree_resolve(tree, tree.top[stree_code(tree, p)], p);
    const uint4 leaf = make_uint4(word & 0x7fffffffu, 0u, word >> 31, 0u);
    float* scales = scales_of(sm_scales, threadIdx.x);
    Rng rng = rng_init_pixel(i & 0xffffu, i >> 16, seed);
    V2 sq = V2{0.0f, 0.0f};
    V2 q = V2{query[2 * (size_t)i], query[2 * (size_t)i + 1]};
    Rng unused;
#if PT_TWO_LEVEL
    const float pdf_s = dtree_walk2(tree, leaf, true, rng, sq, scales, tree.s_node2[leaf.x].rad, tree.s_node2[leaf.x].child);
    unused = rng;
    const float pdf_q = dtree_walk2(tree, leaf, false, unused, q, scales, tree.s_node2[leaf.x].rad, tree.s_node2[leaf.x].child);
#else
    const SNode root = tree.s_node[leaf.x];
    const float pdf_s = dtree_walk(tree, leaf, true, rng, sq, scales, root);
    unused = rng;
    const float pdf_q = dtree_walk(tree, leaf, false, unused, q, scales, root);
#endif
    uint32_t* o = out + 16 * (size_t)i;
    o[0] = leaf_id; o[1] = f2u(sq.x); o[2] = f2u(sq.y); o[3] = f2u(pdf_s); o[4] = f2u(pdf_q);
    o[5] = (uint32_t)rng.state; o[6] = (uint32_t)(rng.state >> 32); o[7] = word2 != word ? 1u : 0u;
    o[8] = f2u(lo.x); o[9] = f2u(lo.y); o[10] = f2u(lo.z); o[11] = f2u(hi.x); o[12] = f2u(hi.y); o[13] = f2u(hi.z);
    const V2 dsq = dir_to_square(p);
    o[14] = f2u(dsq.x); o[15] = f2u(dsq.y);
}

// c_weight[l] += the private counts of leaf l (whole numbers: the order of the sum does not matter); the private copies
// are left at zero, so folding twice (ptgpu_guided_record_batch after a wave) counts nothing twice.
__global__ __launch_bounds__(kBlock) void k_fold_weights(SdTreeDev tree) {
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= tree.n_leaves) return;
    float w = tree.c_weight[l];
    for (uint32_t k = 0; k < kWeightCopies; k++) {
        w += tree.w_private[(size_t)k * tree.n_leaves + l];
        tree.w_private[(size_t)k * tree.n_leaves + l] = 0.0f;
    }
    tree.c_weight[l] = w;
}

// STree::record with the stochastic spatial filter (stree.rs:102-116,186-201) for the queued records.
__global__ __launch_bounds__(kBlock) void k_guide_splat(SdTreeDev tree, GuidedParams gp, RecordQueue rq, uint32_t n_paths) {
    __shared__ uint32_t sm_pend[(kSplatPendingWords + 2 * kSplatStackLds) * kBlock];  // parked adds, then the walk stack
    const uint32_t n_chunks = (n_paths + kBlock - 1) / kBlock;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t n = rq.count[chunk];
        const size_t base = (size_t)chunk * kBlock * gp.n_rows;
        for (uint32_t j0 = 0; j0 < n; j0 += kBlock) {  // block-uniform trip count: the record is wave-cooperative
            const uint32_t j = j0 + threadIdx.x;
            const bool active = j < n;
            float4 a = make_float4(0, 0, 0, 0), b = make_float4(0, 0, 0, 0);
            float2 c = make_float2(0, 0);
            if (active) {
                a = rq.q0[base + j];
                b = rq.q1[base + j];
                c = rq.q2[base + j];
            }
            stree_record_jittered(tree, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, b.w, c.x, c.y, active, sm_pend + threadIdx.x);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Binned record pass. The direct splat above pays one memory-side atomic request per touched quadtree node per
// record (~6 per record, 4 x 10^7 records per batch): that path, not bandwidth, is what bounds it. But a batch brings
// thousands of records to every spatial leaf, and a leaf's collector quadtree is small (~10^2 nodes): so the records
// are first grouped by the leaf they land in, and each group is splatted into a copy of that quadtree in LDS (LDS
// float atomics), which is added to HBM once. Atomic requests to memory drop by two orders of magnitude.
//   k_guide_locate — per queued record: the two spatial descents of the stochastic filter -> leaf; position within
//                    the leaf's group from a counter per leaf; the record is rewritten as {dir square, radiance, leaf}.
//   k_exclusive_scan (pt_refine.h) over the counters -> group starts.
//   k_guide_scatter — moves each record to its place in the grouped array.
//   k_guide_splat_bins — workgroups take equal slices of the grouped array; per run of one leaf: quadtree -> LDS,
//                    box-filter splats with LDS atomics, sums -> HBM (one atomic per touched child sum), weight += run length.
// Sums are the reference's terms in a different order (as with any atomic accumulation).
struct RecordBins {
    uint32_t* count;   // per leaf (zeroed per batch)
    uint32_t* start;   // n_leaves + 1 (exclusive scan of count)
    float4* grouped;   // {sq.x, sq.y, radiance, bits(leaf)} in leaf order
    float2* grouped2;  // the directional stochastic filter's draws in the same order, else NULL
};
// One WAVE per workgroup: a batch brings a leaf ~10^3 records (fewer as the tree grows), and with one wave per leaf run
// neither do 256 threads share a run of a few hundred records nor do waves wait for one another at barriers.
constexpr uint32_t kBinBlock = 64;
constexpr uint32_t kBinStackLds = 4;   // walk-stack entries in LDS (owed subtrees keep the walk on the footprint's rim: shallow)
constexpr uint32_t kBinSpillDepth = 3 * kMaxQuadLevel + 4 - kBinStackLds;  // walk-stack entries beyond the LDS part
constexpr uint32_t kBinTreeCap = 192;  // quadtree nodes held in LDS (the 1 % subdivision threshold keeps trees near 10^2 nodes); larger trees take the direct path

__global__ __launch_bounds__(kBlock) void k_guide_locate(SdTreeDev tree, GuidedParams gp, RecordQueue rq, RecordBins bins, uint32_t first_chunk, uint32_t end_chunk) {
    const uint32_t lane = __lane_id();
    for (uint32_t chunk = first_chunk + blockIdx.x; chunk < end_chunk; chunk += gridDim.x) {
        const uint32_t n = rq.count[chunk];
        const size_t base = (size_t)chunk * kBlock * gp.n_rows;
        for (uint32_t j0 = 0; j0 < n; j0 += kBlock) {
            const uint32_t j = j0 + threadIdx.x;
            const bool active = j < n;
            uint32_t leaf_id = 0xffffffffu;
            V2 sq = V2{0.0f, 0.0f};
            float radiance = 0.0f;
            if (active) {
                const float4 a = rq.q0[base + j], b = rq.q1[base + j];
                const float2 c = rq.q2[base + j];
                const V3 p = mk3(a.x, a.y, a.z);
                radiance = a.w;
                sq = dir_to_square(mk3(b.x, b.y, b.z));
                V3 lo, hi;
                if (gp.spatial_filter == 0u) leaf_id = stree_leaf<false>(tree, p, lo, hi);  // STree::record_nearest (stree.rs:117-126)
                else {
                    (void)stree_leaf<true>(tree, p, lo, hi);
                    const V3 ext = hi - lo;
                    const V3 jlo = p - ext / 2.0f;  // Bounds3::around(p, extent)
                    const V3 jhi = p + ext / 2.0f;
                    const V3 q = clamp3(jlo + mk3(b.w, c.x, c.y) * (jhi - jlo), tree.blo, tree.bhi);
                    leaf_id = stree_leaf<false>(tree, q, lo, hi);
                }
            }
            // Position in the leaf's group. Young trees have few leaves (wave 0: one), and same-address atomics
            // serialise at the memory side: lanes that share the first active lane's leaf take ONE atomic together,
            // as long as such groups are big; everything else goes lane by lane in a single instruction.
            uint32_t pos = 0;
            bool todo = active;
            for (int rounds = 0; rounds < 8; rounds++) {
                const unsigned long long live = __ballot(todo);
                if (live == 0ull) break;
                const int first = __ffsll((long long)live) - 1;
                const uint32_t lead = (uint32_t)__shfl((int)leaf_id, first);
                const bool same = todo && leaf_id == lead;
                const unsigned long long m = __ballot(same);
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (cnt < 8u) break;
                uint32_t b0 = 0;
                if ((int)lane == first) b0 = atomicAdd(bins.count + lead, cnt);
                b0 = (uint32_t)__shfl((int)b0, first);
                if (same) {
                    pos = b0 + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    todo = false;
                }
            }
            if (todo) pos = atomicAdd(bins.count + leaf_id, 1u);
            if (active) {
                rq.q0[base + j] = make_float4(sq.x, sq.y, radiance, u2f(leaf_id));
                rq.q2[base + j] = make_float2(u2f(pos), 0.0f);
            }
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_guide_scatter(GuidedParams gp, RecordQueue rq, RecordBins bins, uint32_t first_chunk, uint32_t end_chunk) {
    for (uint32_t chunk = first_chunk + blockIdx.x; chunk < end_chunk; chunk += gridDim.x) {
        const uint32_t n = rq.count[chunk];
        const size_t base = (size_t)chunk * kBlock * gp.n_rows;
        for (uint32_t j = threadIdx.x; j < n; j += kBlock) {
            const float4 r = rq.q0[base + j];
            const uint32_t pos = f2u(rq.q2[base + j].x);
            const size_t at = (size_t)bins.start[f2u(r.w)] + pos;
            bins.grouped[at] = r;
            if (bins.grouped2) bins.grouped2[at] = rq.q3[base + j];
        }
    }
}

// DTree::record with the box filter for ONE lane against a quadtree whose links and sums are in LDS (LDS_TREE) or in
// HBM (trees beyond kBinTreeCap: plain global atomics, no batching). Same walk, same terms as dtree_record_box.
// The walk stack's first kBinStackLds entries are in LDS; deeper ones (rare) go to `spill`, this thread's column of a
// preallocated HBM array [kBinSpillDepth][threads] — not to a private array: scratch memory of that size, times the
// chip's wave slots, exceeds what the runtime keeps allocated, and it would allocate and free it on every dispatch (~1.5 ms).
template <bool LDS_TREE>
PT_D void dtree_record_box_lane(const uint2* links_of, float* sums_of, float* lazy_of, V2 p, float radiance, uint32_t* lds_stack, uint2* spill, size_t spill_stride) {
    int sp = 0;
    auto push = [&](uint32_t node, Cell cell) {
        const uint2 w = make_uint2(node | (cell.iy << 16), ((1u << cell.level) | cell.ix) | ((cell.iy >> 16) << 24));
        if (sp < (int)kBinStackLds) {
            lds_stack[(2 * sp) * kBinBlock] = w.x;
            lds_stack[(2 * sp + 1) * kBinBlock] = w.y;
        } else
            spill[(size_t)(sp - (int)kBinStackLds) * spill_stride] = w;
        sp++;
    };
    // DTree::leaf (dtree.rs:81-94): the cell that holds p
    Cell cell{0, 0, 0};
    uint32_t ni = 0;
    for (;;) {
        Cell child;
        const int c = cell_child_of(cell, p, child);
        const uint32_t link = child_link(links_of[ni], c);
        cell = child;
        if (link == 0 || child.level >= kMaxQuadLevel) break;
        ni = link;
    }
    const Box2 lb = cell_box(cell);
    Box2 fp = box2_around(p, box2_extent(lb));
    fp = Box2{clamp2(V2{0.0f, 0.0f}, fp.lo, fp.hi), clamp2(V2{1.0f, 1.0f}, fp.lo, fp.hi)};  // Bounds2::clamp (bounds.rs:46-50)
    const float fp_area = box2_area(fp);
    push(0u, Cell{0, 0, 0});
    while (sp > 0) {
        --sp;
        // (the LDS read is unconditional and the rare spilled entry overrides it: as a choice between the two the
        // compiler selects the POINTER and issues flat loads, slower than ds_read, on every pop)
        const int sp_lds = sp < (int)kBinStackLds ? sp : (int)kBinStackLds - 1;
        uint2 w = make_uint2(lds_stack[(2 * sp_lds) * kBinBlock], lds_stack[(2 * sp_lds + 1) * kBinBlock]);
        asm volatile("" : "+v"(w.x), "+v"(w.y));  // (pins the ds_read: otherwise the two loads are sunk into one flat load all the same)
        if (sp >= (int)kBinStackLds) w = spill[(size_t)(sp - (int)kBinStackLds) * spill_stride];
        const uint32_t sx = w.y & 0xffffffu, level = 31u - (uint32_t)__clz((int)sx);
        const uint32_t node = w.x & 0xffffu, ix = sx ^ (1u << level), iy = (w.x >> 16) | ((w.y >> 24) << 16);
        const uint2 links = links_of[node];
        const float cs = u2f((127u - (level + 1)) << 23);  // child cell size
        const float x0 = (float)(2 * ix) * cs, xm = (float)(2 * ix + 1) * cs, x1 = (float)(2 * ix + 2) * cs;
        const float y0 = (float)(2 * iy) * cs, ym = (float)(2 * iy + 1) * cs, y1 = (float)(2 * iy + 2) * cs;
        const float ex[2] = {fminf(fp.hi.x, xm) - fmaxf(fp.lo.x, x0), fminf(fp.hi.x, x1) - fmaxf(fp.lo.x, xm)};
        const float ey[2] = {fminf(fp.hi.y, ym) - fmaxf(fp.lo.y, y0), fminf(fp.hi.y, y1) - fmaxf(fp.lo.y, ym)};
#pragma unroll
        for (int c = 3; c >= 0; c--) {  // pushed in reverse so that child 0 is expanded first
            const float ox = ex[c & 1], oy = ey[c >> 1];
            if (ox <= 0.0f || oy <= 0.0f) continue;  // overlap 0 => weight 0 (or 0/0): skipped by the reference too
            const float wgt = (ox * oy) / fp_area;
            if (wgt == 0.0f || !(wgt == wgt)) continue;
            const uint32_t link = child_link(links, c);
            if (link == 0 || level + 1 >= kMaxQuadLevel) {
                if (LDS_TREE) atomicAdd(sums_of + 4 * node + c, radiance * wgt);
                else unsafeAtomicAdd(sums_of + 4 * node + c, radiance * wgt);
            } else if (ox == cs && oy == cs) {
                // The footprint covers this child's cell entirely: every leaf cell below it is owed
                // radiance * (its area / footprint area) = this term * 4^-(levels below) — a power-of-two scaling,
                // exact. Owe it to the subtree once (k_push_lazy hands it down) instead of walking the subtree.
                if (LDS_TREE) atomicAdd(lazy_of + 4 * node + c, radiance * wgt);
                else unsafeAtomicAdd(lazy_of + 4 * node + c, radiance * wgt);
            } else if (sp < (int)(kBinStackLds + kBinSpillDepth))
                push(link, Cell{2 * ix + (uint32_t)(c & 1), 2 * iy + (uint32_t)(c >> 1), level + 1});
        }
    }
}

// DTree::record with the nearest filter (dtree.rs:208-222): the whole radiance to the leaf cell that holds p.
template <bool LDS_TREE>
PT_D void dtree_record_nearest_lane(const uint2* links_of, float* sums_of, V2 p, float radiance) {
    Cell cell{0, 0, 0};
    uint32_t ni = 0;
    for (;;) {
        Cell child;
        const int c = cell_child_of(cell, p, child);
        const uint32_t link = child_link(links_of[ni], c);
        if (link == 0 || child.level >= kMaxQuadLevel) {
            if (LDS_TREE) atomicAdd(sums_of + 4 * ni + c, radiance);
            else unsafeAtomicAdd(sums_of + 4 * ni + c, radiance);
            return;
        }
        cell = child;
        ni = link;
    }
}

// DTree::record with the stochastic filter (dtree.rs:223-234): the point jittered inside a box of its leaf cell's size
// around it (two draws), clipped to the unit square, then recorded by the nearest filter.
template <bool LDS_TREE>
PT_D void dtree_record_stochastic_lane(const uint2* links_of, float* sums_of, V2 p, float radiance, float d0, float d1) {
    Cell cell{0, 0, 0};  // DTree::leaf (dtree.rs:81-94): the cell that holds p
    uint32_t ni = 0;
    for (;;) {
        Cell child;
        const int c = cell_child_of(cell, p, child);
        const uint32_t link = child_link(links_of[ni], c);
        cell = child;
        if (link == 0 || child.level >= kMaxQuadLevel) break;
        ni = link;
    }
    const Box2 jitter = box2_around(p, box2_extent(cell_box(cell)));
    const V2 q = jitter.lo + V2{d0, d1} * box2_extent(jitter);                        // Bounds2::lerp
    dtree_record_nearest_lane<LDS_TREE>(links_of, sums_of, clamp2(q, V2{0.0f, 0.0f}, V2{1.0f, 1.0f}), radiance);  // Bounds2::unit().clip
}

__global__ __launch_bounds__(kBinBlock) void k_guide_splat_bins(SdTreeDev tree, RecordBins bins, uint32_t per_block, uint2* spill_base, uint32_t filter) {
    __shared__ uint2 sm_links[kBinTreeCap];
    __shared__ float sm_sums[4 * kBinTreeCap];
    __shared__ float sm_lazy[4 * kBinTreeCap];
    __shared__ uint32_t sm_stack[2 * kBinStackLds * kBinBlock];
    const uint32_t total = bins.start[tree.n_leaves];
    const size_t spill_stride = (size_t)gridDim.x * kBinBlock;
    uint2* spill = spill_base + (size_t)blockIdx.x * kBinBlock + threadIdx.x;
    for (uint32_t begin = blockIdx.x * per_block; begin < total; begin += gridDim.x * per_block) {
        const uint32_t end = min(begin + per_block, total);
        uint32_t pos = begin;
        while (pos < end) {  // block-uniform: one run of records of the same leaf per turn
            const uint32_t leaf_id = f2u(bins.grouped[pos].w);
            const uint32_t run_end = min(end, bins.start[leaf_id + 1]);
            const uint32_t c_off = tree.leaf_info[leaf_id].y, n_nodes = tree.c_size[leaf_id];
            float* sums_hbm = reinterpret_cast<float*>(tree.c_rad + c_off);
            float* lazy_hbm = reinterpret_cast<float*>(tree.c_lazy + c_off);
            if (n_nodes <= kBinTreeCap) {
                for (uint32_t i = threadIdx.x; i < n_nodes; i += kBinBlock) sm_links[i] = tree.c_child[c_off + i];
                for (uint32_t i = threadIdx.x; i < 4 * n_nodes; i += kBinBlock) sm_sums[i] = sm_lazy[i] = 0.0f;
                __syncthreads();
                for (uint32_t j = pos + threadIdx.x; j < run_end; j += kBinBlock) {
                    const float4 r = bins.grouped[j];
                    if (filter == 0u) dtree_record_nearest_lane<true>(sm_links, sm_sums, V2{r.x, r.y}, r.z);
                    else if (filter == 2u) {
                        const float2 d = bins.grouped2[j];
                        dtree_record_stochastic_lane<true>(sm_links, sm_sums, V2{r.x, r.y}, r.z, d.x, d.y);
                    } else dtree_record_box_lane<true>(sm_links, sm_sums, sm_lazy, V2{r.x, r.y}, r.z, sm_stack + threadIdx.x, spill, spill_stride);
                }
                __syncthreads();
                for (uint32_t k = threadIdx.x; k < 4 * n_nodes; k += kBinBlock) {
                    const float v = sm_sums[k], z = sm_lazy[k];
                    if (v != 0.0f) unsafeAtomicAdd(sums_hbm + k, v);
                    if (z != 0.0f) unsafeAtomicAdd(lazy_hbm + k, z);
                }
                __syncthreads();
            } else {
                for (uint32_t j = pos + threadIdx.x; j < run_end; j += kBinBlock) {
                    const float4 r = bins.grouped[j];
                    if (filter == 0u) dtree_record_nearest_lane<false>(tree.c_child + c_off, sums_hbm, V2{r.x, r.y}, r.z);
                    else if (filter == 2u) {
                        const float2 d = bins.grouped2[j];
                        dtree_record_stochastic_lane<false>(tree.c_child + c_off, sums_hbm, V2{r.x, r.y}, r.z, d.x, d.y);
                    } else dtree_record_box_lane<false>(tree.c_child + c_off, sums_hbm, lazy_hbm, V2{r.x, r.y}, r.z, sm_stack + threadIdx.x, spill, spill_stride);
                }
            }
            if (threadIdx.x == 0) unsafeAtomicAdd(tree.c_weight + leaf_id, (float)(run_end - pos));  // DTree::record: weight += 1 per record
            pos = run_end;
        }
    }
}

// STree::record with the BOX spatial filter (stree.rs:127-185): the record goes to every spatial leaf its footprint — a box
// of the size of p's leaf around p, clamped to the tree's bounds — overlaps, weighted by the share of the footprint's
// volume; a leaf takes it TWICE, once for each half of its box along its split axis (the reference's recursion halves a
// leaf like any node before it asks whether it is one). The parser cannot select this filter, so it has no fast path: one
// lane per record walks the spatial tree (a private stack of node boxes) and splats into the collectors in HBM directly.
// No draws: the directional filter is box or nearest.
constexpr int kSpatialBoxStack = 96;
__global__ __launch_bounds__(kBinBlock) void k_guide_splat_spatial_box(SdTreeDev tree, GuidedParams gp, RecordQueue rq, uint32_t n_paths, uint2* spill_base, uint32_t* guards) {
    __shared__ uint32_t sm_stack[2 * kBinStackLds * kBinBlock];
    const size_t spill_stride = (size_t)gridDim.x * kBinBlock;
    uint2* spill = spill_base + (size_t)blockIdx.x * kBinBlock + threadIdx.x;
    const uint32_t n_chunks = (n_paths + kBlock - 1) / kBlock;
    for (uint32_t chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const uint32_t n = rq.count[chunk];
        const size_t base = (size_t)chunk * kBlock * gp.n_rows;
        for (uint32_t j = threadIdx.x; j < n; j += kBinBlock) {
            const float4 a = rq.q0[base + j], b = rq.q1[base + j];
            const V3 p = mk3(a.x, a.y, a.z);
            const float radiance = a.w;
            const V2 sq = dir_to_square(mk3(b.x, b.y, b.z));
            V3 lo, hi;
            (void)stree_leaf<true>(tree, p, lo, hi);
            const V3 half = (hi - lo) / 2.0f;                                         // Bounds3::around(p, extent)
            const V3 alo = p - half, ahi = p + half;
            const V3 flo = clamp3(tree.blo, alo, ahi), fhi = clamp3(tree.bhi, alo, ahi);  // Bounds3::clamp (bounds.rs:100-104)
            const V3 fe = fhi - flo;
            const float volume = fe.x * fe.y * fe.z;
            uint32_t st_node[kSpatialBoxStack];
            V3 st_lo[kSpatialBoxStack], st_hi[kSpatialBoxStack];
            int sp = 0;
            st_node[0] = 0u;
            st_lo[0] = tree.blo;
            st_hi[0] = tree.bhi;
            sp = 1;
            while (sp > 0) {
                --sp;
                const uint32_t ni = st_node[sp];
                const V3 nlo = st_lo[sp], nhi = st_hi[sp];
                const uint4 node = tree.snodes[ni];
                const int ax = (int)node.z;
                const float mid = (get(nlo, ax) + get(nhi, ax)) / 2.0f;               // centroid()[split_axis]
                for (int i = 0; i < 2; i++) {
                    V3 clo = nlo, chi = nhi;
                    if (i == 0) set(chi, ax, mid);
                    else set(clo, ax, mid);
                    const V3 olo = mk3(fmaxf(flo.x, clo.x), fmaxf(flo.y, clo.y), fmaxf(flo.z, clo.z));
                    const V3 ohi = mk3(fminf(fhi.x, chi.x), fminf(fhi.y, chi.y), fminf(fhi.z, chi.z));
                    const V3 oe = ohi - olo;
                    const float overlap = (oe.x <= 0.0f || oe.y <= 0.0f || oe.z <= 0.0f) ? 0.0f : oe.x * oe.y * oe.z;
                    const float w = overlap / volume;
                    if (w == 0.0f) continue;
                    if (node.x == 0u) {  // a leaf: DTree::record(p2, radiance * w, w) (dtree.rs:200-207)
                        const uint32_t leaf_id = node.w, c_off = tree.leaf_info[leaf_id].y;
                        float* sums = reinterpret_cast<float*>(tree.c_rad + c_off);
                        if (gp.directional_filter == 0u) dtree_record_nearest_lane<false>(tree.c_child + c_off, sums, sq, radiance * w);
                        else dtree_record_box_lane<false>(tree.c_child + c_off, sums, reinterpret_cast<float*>(tree.c_lazy + c_off), sq, radiance * w, sm_stack + threadIdx.x, spill, spill_stride);
                        unsafeAtomicAdd(tree.c_weight + leaf_id, w);
                    } else if (sp < kSpatialBoxStack) {
                        st_node[sp] = i == 0 ? node.x : node.y;
                        st_lo[sp] = clo;
                        st_hi[sp] = chi;
                        sp++;
                    } else
                        atomicAdd(guards + kGuardVertexStore, 1u);  // (a spatial tree deeper than the walk's stack: counted, never seen)
                }
            }
        }
    }
}

// Hands the owed radiance down (see dtree_record_box_lane): one thread per spatial leaf walks its collector quadtree
// parents first; what a node owes to a child's subtree is split in four, to that child's leaf cells directly and to
// its interior children as their debt. Runs once per wave, after the last record pass and before the statistics
// leave the device (ptgpu_guided_stats) or are refined.
constexpr uint32_t kPushTreeLds = 256;  // nodes of a leaf's collector quadtree staged in LDS (larger trees are swept in global memory)
constexpr uint32_t kPushWavePerLeafUpTo = 49152;  // leaves: beyond, one thread per leaf (LANES = 1) keeps more sweeps in flight (pt_refine.h)
template <uint32_t LANES>
__global__ __launch_bounds__(kBlock) void k_push_lazy(SdTreeDev t) {
    // LANES = 64: one wave per spatial leaf: the tree goes to LDS, one lane sweeps it there, the wave writes it back (pt_refine.h)
    struct Staged {
        float4 rad[kPushTreeLds], lazy[kPushTreeLds];
        uint2 child[kPushTreeLds];
    };
    __shared__ Staged sm[LANES == 64u ? kBlock / 64u : 1u];
    const uint32_t leaf = LANES == 64u ? blockIdx.x * (kBlock / 64u) + threadIdx.x / 64u : blockIdx.x * kBlock + threadIdx.x, lane = LANES == 64u ? __lane_id() : 0u;
    if (leaf >= t.n_leaves) return;
    Staged& st = sm[LANES == 64u ? threadIdx.x / 64u : 0u];
    const uint32_t base = t.leaf_info[leaf].y, n = t.c_size[leaf];
    const bool staged = LANES == 64u && n <= kPushTreeLds;
    // interior children have larger indices than their parent and lie above kMaxQuadLevel (pt_refine.h:
    // reshape_collector), so one sweep from the first node to the last sees every debt after it has been handed down
    // (instantiated for the LDS copy and for global memory: see k_refine_sums)
    auto sweep = [n](float4* rad, float4* lazy, const uint2* child) {
        for (uint32_t i = 0; i < n; i++) {
            const float4 z = lazy[i];
            if (z.x == 0.0f && z.y == 0.0f && z.z == 0.0f && z.w == 0.0f) continue;
            const uint2 links = child[i];
            lazy[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            const float owed[4] = {z.x, z.y, z.z, z.w};
            for (int c = 0; c < 4; c++) {
                const uint32_t link = child_link(links, c);
                if (owed[c] == 0.0f || link <= i || link >= n) continue;
                const float q = owed[c] * 0.25f;
                const uint2 below = child[link];
                float4 s = rad[link], d = lazy[link];
                if (child_link(below, 0) == 0u) s.x += q; else d.x += q;
                if (child_link(below, 1) == 0u) s.y += q; else d.y += q;
                if (child_link(below, 2) == 0u) s.z += q; else d.z += q;
                if (child_link(below, 3) == 0u) s.w += q; else d.w += q;
                rad[link] = s;
                lazy[link] = d;
            }
        }
    };
    if (staged) {
        bool any = false;
        for (uint32_t i = lane; i < n; i += LANES) {
            const float4 z = t.c_lazy[base + i];
            st.lazy[i] = z;
            st.rad[i] = t.c_rad[base + i];
            st.child[i] = t.c_child[base + i];
            any = any || z.x != 0.0f || z.y != 0.0f || z.z != 0.0f || z.w != 0.0f;
        }
        if (__ballot(any) == 0ull) return;  // nothing owed anywhere in this tree
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) sweep(st.rad, st.lazy, st.child);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (uint32_t i = lane; i < n; i += LANES) {
            t.c_rad[base + i] = st.rad[i];
            t.c_lazy[base + i] = st.lazy[i];
        }
    } else if (lane == 0)
        sweep(t.c_rad + base, t.c_lazy + base, t.c_child + base);
}

}  // namespace pt
