// pt_refine.h — STree::new_iteration on the device (stree.rs:42-101,255-260; dtree.rs:185-198,
// 291-328). Between two waves every spatial leaf's collector quadtree is (1) summed bottom-up (DTree::build), (2) copied
// into the sampling arena the next wave reads (emit_into), (3) reshaped by subdivide_and_clear. The collector sums
// already live in HBM (the record pass splatted them there, RCCL all-reduced them in place), so doing this here removes
// the round trip of the whole tree over PCIe and the host pass over it. The spatial binary tree is refined here as well
// (second half of this file): the host only reads four totals per refinement to size the next collector.
// One thread (or one wave) per leaf: the trees are small (~10^2 nodes) and there are 10^4..10^5 of them.
// The arithmetic (sum order, IEEE division, 0.25^depth) is the host's/reference's, so the decisions are the same bits;
// only the numbering of quadtree nodes inside a leaf differs (breadth of the walk here, free-list order there), which
// nothing observes.
#pragma once
#include "pt_guided.h"

namespace pt {

// What the refinement tells the host at its one synchronisation (device words; see the spatial refinement below).
enum StreeHeader : uint32_t {
    kHdrNodes = 0, kHdrLeaves = 1, kHdrArenaLo = 2, kHdrArenaHi = 3, kHdrLevels = 4, kHdrFlags = 5, kHdrCollector = 6, kHdrRefineError = 7, kHdrWords = 8
};
enum StreeFlags : uint32_t { kStreeTooDeep = 1u, kStreeNodesOver = 2u, kStreeLeavesOver = 4u, kStreeArenaOver = 8u, kStreeArenaBeyondU32 = 16u };
struct RefineArgs {
    // the generation just trained
    const uint2* old_child;     // collector topology
    float4* old_rad;            // collector sums (leaf cells as splatted; interior sums are rebuilt in place)
    const float* old_weight;    // per old leaf
    const uint32_t* old_off;    // per old leaf: first node
    const uint32_t* old_size;   // per old leaf: node count
    uint32_t n_old_leaves;
    // the next generation
    const uint32_t* src_leaf;   // per new leaf: the old leaf it descends from
    const uint32_t* halvings;   // per new leaf: how many spatial splits lie between (collector weight / 2^k)
    const uint32_t* s_off;      // per new leaf: first node in the sampling arena (prefix of old sizes, host)
    uint32_t n_new_leaves;
    const uint32_t* header;     // when set (before the host knows the counts): n_new_leaves = header[kHdrLeaves], nothing runs if header[kHdrFlags] (see the spatial refinement below)
    SNode* s_node;              // sampling arena
    uint32_t* new_size;         // per new leaf: node count of the reshaped collector
    const uint32_t* new_off;    // per new leaf: first node (exclusive scan of new_size)
    uint2* new_child;           // reshaped collector topology
    uint4* leaf_info;           // per new leaf: {s_off, new_off, sampler mean == 0, bits of the sampler's statistical weight}
    uint32_t* error_flag;       // set when a quadtree would need more than 65535 nodes
    uint32_t* guards;           // SceneDev::guards
    float threshold;            // directional threshold (0.01, mod.rs:73)
};

PT_D float sum4(float4 r) { return (((0.0f + r.x) + r.y) + r.z) + r.w; }
PT_D uint32_t refine_new_leaves(const RefineArgs& a) { return !a.header ? a.n_new_leaves : a.header[kHdrFlags] ? 0u : a.header[kHdrLeaves]; }

// The per-leaf kernels below run ONE WAVE per spatial leaf (four leaves per workgroup): the walk over a leaf's quadtree
// is serial (one lane does it), but every step of it used to be a round trip to global memory (~120 dependent steps, 0.15-0.2
// ms per kernel however small the wave: 5 ms per render, a per-rank constant of a multi-GPU job). The wave first copies
// the leaf's tree into LDS, 64 nodes at a time, the serial walk then runs on LDS, and the wave writes the result back.
// Trees larger than kLeafTreeLds nodes (the 1 % subdivision threshold keeps them near 10^2) are walked in global memory.
// That pays while the leaves are few: beyond a kernel's threshold one THREAD per leaf (LANES = 1, no staging) keeps more
// walks in flight than one wave per leaf does. Measured per wave of the headline (microseconds, wave per leaf against
// thread per leaf, at 4 418 / 6 312 / 8 814 / 11 881 / 15 833 leaves): sums 24-52 against 176-193 (a thread's walk is
// ~120 dependent round trips to memory however few the leaves); lazy push 37-75 against 36-257; sampling copy + count
// 142 / 165 / 208 / 253 against 253 / 249 / 272 / 271; reshape + write 140 / 170 / 222 / 273 against 158 / 158 / 162 / 182.
constexpr uint32_t kSumsWavePerLeafUpTo = 49152, kEmitWavePerLeafUpTo = 1u << 22, kWriteWavePerLeafUpTo = 1u << 22;  // (the reshape is wave-parallel since round 3: reshape_collector_wave)
#ifndef PT_LEAF_TREE_LDS
#define PT_LEAF_TREE_LDS 256  // (a multiple of 64; tests build a variant with 64 so that the serial fallback runs on real trees)
#endif
constexpr uint32_t kLeafTreeLds = PT_LEAF_TREE_LDS;
constexpr uint32_t kLeavesPerBlock = kBlock / 64u;
struct ReshapeItem {  // walk stack of reshape_collector
    uint32_t src_out;  // source node (0xffff = a cell split in this pass) | output index << 16
    float quarter;     // for a split cell: the value each of its four children starts with
    uint32_t depth;
};
constexpr int kReshapeStack = 3 * kMaxQuadLevel + 16;
struct LeafTreeLds {
    float4 rad[kLeafTreeLds];
    uint2 child[kLeafTreeLds];
};
struct ReshapeWaveLds {                // reshape_collector_wave's per-node results
    uint8_t depth[kLeafTreeLds];       // item depth of a node that survives (root: 1), 0 = dropped with its subtree
    uint16_t out_index[kLeafTreeLds];  // its index in the reshaped tree
    uint16_t spawn_at[kLeafTreeLds];   // where the nodes spawned by its split cells start, behind the surviving nodes
};
struct LeafWalkLds {  // k_refine_emit_count / k_refine_write, one wave per leaf: the tree and, for the serial walk, its stack
    LeafTreeLds tree;  // (private memory otherwise: 1 KB per lane of scratch, each push and pop a trip to memory)
    union {
        ReshapeItem stack[kReshapeStack];
        ReshapeWaveLds par;
    };
};
PT_D void wave_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }  // LDS traffic of one wave is in order; this keeps the compiler in line too

// DTree::build (dtree.rs:185-198): interior children take the sum of their subtree, bottom-up. One thread per OLD leaf.
// Every node's interior children have larger indices than the node (reshape_collector numbers a child when its parent
// is expanded), so one sweep from the last node to the first finds every child's sums already final — no stack.
template <uint32_t LANES>
__global__ __launch_bounds__(kBlock) void k_refine_sums(RefineArgs a) {
    __shared__ LeafTreeLds sm[LANES == 64u ? kLeavesPerBlock : 1u];
    const uint32_t leaf = LANES == 64u ? blockIdx.x * kLeavesPerBlock + threadIdx.x / 64u : blockIdx.x * kBlock + threadIdx.x, lane = LANES == 64u ? __lane_id() : 0u;
    if (leaf >= a.n_old_leaves) return;
    LeafTreeLds& t = sm[LANES == 64u ? threadIdx.x / 64u : 0u];
    const uint32_t base = a.old_off[leaf], n = a.old_size[leaf];
    const bool staged = LANES == 64u && n <= kLeafTreeLds;
    // (the sweep is instantiated once for the LDS copy and once for global memory: through one pointer that may be
    // either, every access would be a FLAT one, slower than ds_read / global_load, on the serial chain)
    auto sweep = [n](float4* rad, const uint2* child) {
        for (uint32_t i = n; i-- > 0;) {
            const uint2 links = child[i];
            if ((links.x | links.y) == 0u) continue;  // four leaf cells
            float4 r = rad[i];
            const uint32_t l0 = child_link(links, 0), l1 = child_link(links, 1), l2 = child_link(links, 2), l3 = child_link(links, 3);
            if (l0 > i && l0 < n) r.x = sum4(rad[l0]);
            if (l1 > i && l1 < n) r.y = sum4(rad[l1]);
            if (l2 > i && l2 < n) r.z = sum4(rad[l2]);
            if (l3 > i && l3 < n) r.w = sum4(rad[l3]);
            rad[i] = r;
        }
    };
    if (staged) {
        for (uint32_t i = lane; i < n; i += LANES) {
            t.rad[i] = a.old_rad[base + i];
            t.child[i] = a.old_child[base + i];
        }
        wave_sync_lds();
        // the whole wave: every node takes its interior children's sums again until nothing changes (as many passes as
        // the tree is deep; the terms and their order per node are the sweep's, so are the bits)
        for (;;) {
            bool changed = false;
            for (uint32_t i = lane; i < n; i += LANES) {
                const uint2 links = t.child[i];
                if ((links.x | links.y) == 0u) continue;
                float4 r = t.rad[i];
                const float4 was = r;
                const uint32_t l0 = child_link(links, 0), l1 = child_link(links, 1), l2 = child_link(links, 2), l3 = child_link(links, 3);
                if (l0 > i && l0 < n) r.x = sum4(t.rad[l0]);
                if (l1 > i && l1 < n) r.y = sum4(t.rad[l1]);
                if (l2 > i && l2 < n) r.z = sum4(t.rad[l2]);
                if (l3 > i && l3 < n) r.w = sum4(t.rad[l3]);
                if (f2u(r.x) != f2u(was.x) || f2u(r.y) != f2u(was.y) || f2u(r.z) != f2u(was.z) || f2u(r.w) != f2u(was.w)) {
                    t.rad[i] = r;
                    changed = true;
                }
            }
            wave_sync_lds();
            if (!__any(changed)) break;
        }
        for (uint32_t i = lane; i < n; i += LANES) a.old_rad[base + i] = t.rad[i];
    } else if (lane == 0)
        sweep(a.old_rad + base, a.old_child + base);
}

// DTree::subdivide_and_clear (dtree.rs:291-324) as a pure function of the summed tree: drop subtrees that carry less
// than `threshold` of the total, split leaf cells that carry more (their four children start with a quarter each and
// are examined in turn). WRITE = false counts the nodes of the result, WRITE = true emits its topology (sums start at
// zero: the caller clears the new statistics buffer). Output numbering: a node's interior children get consecutive
// indices when the node is expanded.
template <bool WRITE>
PT_D uint32_t reshape_collector(const RefineArgs& a, const float4* src_rad, const uint2* src_child, uint32_t new_base, ReshapeItem* stack) {
    using Item = ReshapeItem;
    int sp = 0;
    const float total = sum4(src_rad[0]);
    uint32_t counter = 1;
    stack[sp++] = Item{0u, 0.0f, 1u};
    for (uint32_t step = 0; sp > 0 && step < 70000u; step++) {
        const Item it = stack[--sp];
        const uint32_t src = it.src_out & 0xffffu, out = it.src_out >> 16;
        float r[4];
        uint2 links = make_uint2(0u, 0u);
        if (src != 0xffffu) {
            const float4 v = src_rad[src];
            r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.w;
            links = src_child[src];
        } else {
            r[0] = r[1] = r[2] = r[3] = it.quarter;
        }
        uint32_t out_link[4] = {0u, 0u, 0u, 0u};
        for (int c = 0; c < 4; c++) {
            float frac;
            if (total > 0.0f) frac = r[c] / total;
            else {
                frac = 1.0f;
                for (uint32_t k = 0; k < it.depth; k++) frac *= 0.25f;  // 0.25f32.powi(depth), exact
            }
            if (frac < a.threshold) continue;  // stays (or becomes) a leaf cell
            if (it.depth >= kMaxQuadLevel) {  // a node there would be out of every walk's reach (they stop at kMaxQuadLevel)
                if (!WRITE) atomicAdd(a.guards + kGuardQuadLevel, 1u);
                continue;
            }
            if (counter >= 65535u) {
                *a.error_flag = 1u;
                continue;
            }
            const uint32_t child_out = counter++;
            out_link[c] = child_out;
            const uint32_t link = src != 0xffffu ? child_link(links, c) : 0u;
            if (sp < kReshapeStack)
                stack[sp++] = link == 0u ? Item{0xffffu | (child_out << 16), r[c] / 4.0f, it.depth + 1u} : Item{link | (child_out << 16), 0.0f, it.depth + 1u};
            else
                *a.error_flag = 2u;
        }
        if (WRITE) a.new_child[new_base + out] = make_uint2(out_link[0] | (out_link[1] << 16), out_link[2] | (out_link[3] << 16));
    }
    return counter;
}

// The same for a tree staged in LDS, by the WHOLE wave. The serial walk above is ~100 steps of ~150 dependent instructions
// on one lane — 35 us per leaf however small the tree, 0.2-0.3 ms per kernel and wave of the render, a constant every rank
// of a multi-GPU job pays. Nothing in it is sequential but the numbering: whether a node survives depends on its ancestors'
// decisions only (propagated top-down, one level per pass: children have larger indices than parents, each node one
// parent), what a split cell spawns is a closed form (a cell of fraction f examined at depth d becomes a complete 4-ary
// subtree with a level j wherever f / 4^j >= threshold and d + j < kMaxQuadLevel — exact: dividing by four commutes with
// the rounding of f / total), and any numbering that keeps children behind their parents will do (surviving nodes in
// their old order, spawned subtrees behind them, level by level). Same decisions, hence the same tree up to numbering.
PT_D uint32_t spawned_by_cell(float value, float total, float threshold, uint32_t depth, uint32_t& levels, uint32_t& too_deep) {
    uint32_t nodes = 0, width = 1;
    levels = 0;
    for (uint32_t d = depth;; d++) {
        float frac;
        if (total > 0.0f) frac = value / total;
        else {
            frac = 1.0f;
            for (uint32_t k = 0; k < d; k++) frac *= 0.25f;
        }
        if (frac < threshold) break;
        if (d >= kMaxQuadLevel) {
            too_deep += width;
            break;
        }
        nodes += width;
        levels++;
        if (nodes > 70000u) break;  // (a NaN fraction is "not below the threshold" all the way down: beyond the u16 links, an error either way)
        width *= 4u;
        value = value / 4.0f;
    }
    return nodes;
}
template <bool WRITE>
PT_D uint32_t reshape_collector_wave(const RefineArgs& a, const LeafTreeLds& t, ReshapeWaveLds& w, uint32_t n, uint32_t new_base) {
    const uint32_t lane = __lane_id();
    const float total = sum4(t.rad[0]);
    const auto frac_of = [&](float v, uint32_t depth) {
        if (total > 0.0f) return v / total;
        float f = 1.0f;
        for (uint32_t k = 0; k < depth; k++) f *= 0.25f;
        return f;
    };
    // (1) who survives, and at which depth
    for (uint32_t i = lane; i < n; i += 64u) w.depth[i] = i == 0u ? 1u : 0u;
    wave_sync_lds();
    uint32_t expanded = 0, too_deep = 0;
    for (;;) {
        bool any = false;
        for (uint32_t k = 0; k < kLeafTreeLds / 64u; k++) {
            const uint32_t i = k * 64u + lane;
            if (i >= n || ((expanded >> k) & 1u)) continue;
            const uint32_t d = w.depth[i];
            if (d == 0u) continue;
            expanded |= 1u << k;
            any = true;
            const float4 v = t.rad[i];
            const uint2 links = t.child[i];
            const float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t link = child_link(links, c);
                if (link == 0u || link >= n) continue;
                if (frac_of(r[c], d) < a.threshold) continue;
                if (d >= kMaxQuadLevel) too_deep++;
                else w.depth[link] = (uint8_t)(d + 1u);
            }
        }
        wave_sync_lds();
        if (!__any(any)) break;
    }
    // (2) numbering: surviving nodes in their old order, then what their split cells spawn
    uint32_t n_alive = 0, n_spawned = 0;
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        const uint32_t d = i < n ? w.depth[i] : 0u;
        uint32_t mine = 0;
        if (d != 0u) {
            const float4 v = t.rad[i];
            const uint2 links = t.child[i];
            const float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) {
                uint32_t levels;
                if (child_link(links, c) == 0u) mine += spawned_by_cell(r[c], total, a.threshold, d, levels, too_deep);
            }
        }
        const unsigned long long alive = __ballot(d != 0u);
        uint32_t x = mine;
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, off);
            if (lane >= off) x += y;
        }
        if (d != 0u) {
            w.out_index[i] = (uint16_t)(n_alive + (uint32_t)__popcll(alive & ((1ull << lane) - 1ull)));
            w.spawn_at[i] = (uint16_t)(n_spawned + x - mine);
        }
        n_alive += (uint32_t)__popcll(alive);
        const uint32_t block_total = (uint32_t)__shfl((int)x, 63);
        n_spawned = n_spawned + block_total < n_spawned ? 0xffffffffu : n_spawned + block_total;  // (saturating)
    }
    wave_sync_lds();
    const uint32_t size = n_spawned > 0xffffu ? 0x10000u : n_alive + n_spawned;
    if (!WRITE) {
        if (too_deep) atomicAdd(a.guards + kGuardQuadLevel, too_deep);
        if (size > 65535u && lane == 0u) *a.error_flag = 1u;
        return size;
    }
    if (size > 65535u) return size;  // (the count pass has flagged it; the host fails the wave)
    // (3) topology
    for (uint32_t k = 0; k * 64u < n; k++) {
        const uint32_t i = k * 64u + lane;
        const uint32_t d = i < n ? w.depth[i] : 0u;
        if (d == 0u) continue;
        const float4 v = t.rad[i];
        const uint2 links = t.child[i];
        const float r[4] = {v.x, v.y, v.z, v.w};
        uint32_t cursor = n_alive + w.spawn_at[i], out_link[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t link = child_link(links, c);
            if (link != 0u) {
                if (link < n && w.depth[link] != 0u) out_link[c] = w.out_index[link];
                continue;
            }
            uint32_t levels, unused = 0;
            const uint32_t nodes = spawned_by_cell(r[c], total, a.threshold, d, levels, unused);
            if (nodes == 0u) continue;
            out_link[c] = cursor;
            uint32_t level_at = cursor, width = 1;  // the subtree level by level; node q of a level has children 4q .. 4q + 3 of the next
            for (uint32_t j = 0; j < levels; j++) {
                const uint32_t next_at = level_at + width;
                for (uint32_t q = 0; q < width; q++) {
                    const uint32_t f = next_at + 4u * q;
                    a.new_child[new_base + level_at + q] = j + 1u < levels ? make_uint2(f | ((f + 1u) << 16), (f + 2u) | ((f + 3u) << 16)) : make_uint2(0u, 0u);
                }
                level_at = next_at;
                width *= 4u;
            }
            cursor += nodes;
        }
        a.new_child[new_base + w.out_index[i]] = make_uint2(out_link[0] | (out_link[1] << 16), out_link[2] | (out_link[3] << 16));
    }
    return size;
}

// Per NEW leaf: the sampling copy (emit_into), whether that copy's mean is zero, and the size of the reshaped collector.
template <uint32_t LANES>
__global__ __launch_bounds__(kBlock) void k_refine_emit_count(RefineArgs a) {
    __shared__ LeafWalkLds sm[LANES == 64u ? kLeavesPerBlock : 1u];
    const uint32_t leaf = LANES == 64u ? blockIdx.x * kLeavesPerBlock + threadIdx.x / 64u : blockIdx.x * kBlock + threadIdx.x, lane = LANES == 64u ? __lane_id() : 0u;
    if (leaf >= refine_new_leaves(a)) return;
    LeafTreeLds& t = sm[LANES == 64u ? threadIdx.x / 64u : 0u].tree;
    ReshapeItem* lds_stack = sm[LANES == 64u ? threadIdx.x / 64u : 0u].stack;
    const uint32_t src = a.src_leaf[leaf];
    const uint32_t ob = a.old_off[src], n = a.old_size[src], sb = a.s_off[leaf];
    const bool staged = LANES == 64u && n <= kLeafTreeLds;
    for (uint32_t k = lane; k < n; k += LANES) {
        const float4 r = a.old_rad[ob + k];
        const uint2 c = a.old_child[ob + k];
        a.s_node[sb + k] = SNode{r, c, make_uint2(0u, 0u)};
        if (staged) {
            t.rad[k] = r;
            t.child[k] = c;
        }
    }
    wave_sync_lds();
    if (staged) {  // the whole wave (reshape_collector_wave)
        const uint32_t size = reshape_collector_wave<false>(a, t, sm[LANES == 64u ? threadIdx.x / 64u : 0u].par, n, 0u);
        if (lane == 0) a.new_size[leaf] = size;
    }
    if (lane != 0) return;
    float weight = a.old_weight[src];
    for (uint32_t k = 0; k < a.halvings[leaf]; k++) weight /= 2.0f;  // STree split: collector weight halves (stree.rs:55-101)
    // DTree::mean() == 0 (dtree.rs:71-77) decides between uniform and tree sampling
    const float mean = weight == 0.0f ? 0.0f : sum4(a.old_rad[ob]) / (kFourPi * weight);
    if (staged) {
    } else if (LANES == 64u) a.new_size[leaf] = reshape_collector<false>(a, a.old_rad + ob, a.old_child + ob, 0u, lds_stack);  // (separate instances: see k_refine_sums)
    else {
        ReshapeItem stack[kReshapeStack];
        a.new_size[leaf] = reshape_collector<false>(a, a.old_rad + ob, a.old_child + ob, 0u, stack);
    }
    a.leaf_info[leaf] = make_uint4(sb, 0u, mean == 0.0f ? 1u : 0u, f2u(weight));  // w: the sampling copy's statistical weight (parity hooks)
}

template <uint32_t LANES>
__global__ __launch_bounds__(kBlock) void k_refine_write(RefineArgs a) {
    __shared__ LeafWalkLds sm[LANES == 64u ? kLeavesPerBlock : 1u];
    const uint32_t leaf = LANES == 64u ? blockIdx.x * kLeavesPerBlock + threadIdx.x / 64u : blockIdx.x * kBlock + threadIdx.x, lane = LANES == 64u ? __lane_id() : 0u;
    if (leaf >= refine_new_leaves(a)) return;
    LeafTreeLds& t = sm[LANES == 64u ? threadIdx.x / 64u : 0u].tree;
    ReshapeItem* lds_stack = sm[LANES == 64u ? threadIdx.x / 64u : 0u].stack;
    const uint32_t src = a.src_leaf[leaf];
    const uint32_t ob = a.old_off[src], n = a.old_size[src];
    const bool staged = LANES == 64u && n <= kLeafTreeLds;
    if (staged) {
        for (uint32_t k = lane; k < n; k += LANES) {
            t.rad[k] = a.old_rad[ob + k];
            t.child[k] = a.old_child[ob + k];
        }
        wave_sync_lds();
    }
    const uint32_t nb = a.new_off[leaf];
    if (staged) (void)reshape_collector_wave<true>(a, t, sm[LANES == 64u ? threadIdx.x / 64u : 0u].par, n, nb);  // the whole wave
    if (lane != 0) return;
    if (staged) {
    } else if (LANES == 64u) (void)reshape_collector<true>(a, a.old_rad + ob, a.old_child + ob, nb, lds_stack);
    else {
        ReshapeItem stack[kReshapeStack];
        (void)reshape_collector<true>(a, a.old_rad + ob, a.old_child + ob, nb, stack);
    }
    a.leaf_info[leaf].y = nb;
}

// Exclusive scan of `n` counts by one workgroup (n: spatial leaves or nodes, 1..10^6); out[n] = total. Tiles of 4096
// consecutive elements, four per thread in one 16-byte load (in and out are buffer bases: 16-byte aligned), the next tile
// requested before this one is scanned: a tile costs six shuffles, one barrier and the sixteen wave totals. (The first
// version gave every thread n / 1024 consecutive elements — 64 different lines per load instruction: 57 us for 42 k elements.)
PT_D void block_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* total_out) {
    __shared__ uint32_t sm_totals[2][16];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const auto fetch = [&](uint32_t base) {
        const uint32_t i = base + 4u * t;
        if (i + 3u < n) return *reinterpret_cast<const uint4*>(in + i);
        return make_uint4(i < n ? in[i] : 0u, i + 1u < n ? in[i + 1u] : 0u, i + 2u < n ? in[i + 2u] : 0u, 0u);
    };
    uint4 cur = fetch(0u);
    uint32_t carry = 0, flip = 0;
    for (uint32_t base = 0; base < n; base += 4096u, flip ^= 1u) {
        uint4 nxt = make_uint4(0u, 0u, 0u, 0u);
        if (base + 4096u < n) nxt = fetch(base + 4096u);
        const uint32_t mine = cur.x + cur.y + cur.z + cur.w;
        uint32_t x = mine;
        for (uint32_t off = 1; off < 64u; off <<= 1) {
            const uint32_t y = (uint32_t)__shfl_up((int)x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63u) sm_totals[flip][wave] = x;
        __syncthreads();  // (one per tile: the other buffer is written next, and nobody passes THAT tile's barrier while a wave still reads this one)
        uint32_t before = 0, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; w++) {
            const uint32_t v = sm_totals[flip][w];
            before += w < wave ? v : 0u;
            all += v;
        }
        const uint32_t first = carry + before + x - mine, i = base + 4u * t;
        const uint4 o = make_uint4(first, first + cur.x, first + cur.x + cur.y, first + cur.x + cur.y + cur.z);
        if (i + 3u < n) *reinterpret_cast<uint4*>(out + i) = o;
        else {
            if (i < n) out[i] = o.x;
            if (i + 1u < n) out[i + 1u] = o.y;
            if (i + 2u < n) out[i + 2u] = o.z;
        }
        carry += all;
        cur = nxt;
    }
    if (t == 0) {
        out[n] = carry;
        if (total_out) *total_out = carry;
    }
}
__global__ __launch_bounds__(1024) void k_exclusive_scan(const uint32_t* in, uint32_t* out, uint32_t n) { block_exclusive_scan(in, out, n, nullptr); }

// ---------------------------------------------------------------------------------------------------------------
// The spatial half of STree::new_iteration on the device: STree::subdivide_nodes (stree.rs:55-101), the numbering of
// the new leaves and the layout of the next sampling arena. The reference sweeps the node array once, front to back;
// a leaf heavier than the threshold gets two children APPENDED (copies of it with half its weight), and the sweep
// reaches them later and may split them again. In closed form: an old leaf of weight w ends up as a complete subtree of
// depth h = the number of halvings that bring w under the threshold (all nodes of one level of it weigh the same, so
// they split together), and the append order is level by level — first the level-1 children of every splitting old
// leaf in old node order, then all level-2 nodes in the order of their parents, and so on. So the first node of leaf
// i's level-L block sits at n_old + the exclusive scan, over (L, i) in that order, of the block sizes 2^L.
//   k_stree_plan      per old node: h, its block sizes per level, and the totals (nodes, leaves, sampling-arena nodes)
//   k_exclusive_scan_levels over the block sizes of the levels in use
//   k_stree_append    writes the new node array (old nodes, now interior where they split, then the blocks) and, per
//                     node, whether it is a leaf and the size of the sampling copy it will need
//   k_exclusive_scan_dev (twice) -> leaf ids in node order (the reference's numbering) and arena offsets
//   k_stree_finalize  leaf words of the node array, src_leaf / halvings / s_off per new leaf
// The host learns the four totals at the one synchronisation of the refinement; until then kernels read them from the
// header and buffers are sized by capacity (exceeding one sets a flag, nothing is written, the host grows and repeats).
constexpr uint32_t kMaxSpatialHalvings = 20;  // a leaf may split into 2^20 in one refinement; beyond is an error (the reference would sweep as long)
struct StreeArgs {
    const uint4* old_nodes;     // {left, right, axis, -} / leaves {0, sampler word, axis, leaf id}
    uint32_t n_old_nodes;
    const float* old_weight;    // per old leaf: the collector's statistical weight
    const uint32_t* old_size;   // per old leaf: collector node count (= the size of a sampling copy of it)
    float threshold;            // spatial threshold * sqrt(2^iteration)
    uint32_t* halvings_of;      // per old node
    uint32_t* blocks;           // [kMaxSpatialHalvings][n_old_nodes]: nodes of level L + 1 under old node i
    uint32_t* block_start;      // exclusive scan of the above over the levels in use
    uint4* new_nodes;
    uint2* origin;              // per new node: {old leaf, halvings} (leaves)
    uint32_t* is_leaf;          // per new node (+ 1: scan total)
    uint32_t* copy_size;        // per new node (+ 1)
    uint32_t* leaf_rank;        // exclusive scans of the two above
    uint32_t* copy_start;
    uint32_t* src_leaf;         // per new leaf (RefineArgs)
    uint32_t* halvings;
    uint32_t* s_off;
    uint32_t* header;           // StreeHeader
    uint32_t cap_nodes, cap_leaves;
    uint64_t cap_arena;
};

__global__ __launch_bounds__(kBlock) void k_stree_plan(StreeArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t h = 0, nodes = 0, leaves = 0;
    unsigned long long arena = 0;
    if (i < a.n_old_nodes) {
        const uint4 n = a.old_nodes[i];
        nodes = 1;
        if (n.x == 0u) {
            float w = a.old_weight[n.w];
            while (w > a.threshold && h <= kMaxSpatialHalvings) {  // subdivide_nodes: split while heavier than the threshold, children weigh half
                w /= 2.0f;
                h++;
            }
            if (h > kMaxSpatialHalvings) {
                atomicOr(a.header + kHdrFlags, (uint32_t)kStreeTooDeep);
                h = kMaxSpatialHalvings;
            }
            nodes += (2u << h) - 2u;
            leaves = 1u << h;
            arena = (unsigned long long)leaves * a.old_size[n.w];
        }
        a.halvings_of[i] = h;
        for (uint32_t l = 0; l < kMaxSpatialHalvings; l++) a.blocks[(size_t)l * a.n_old_nodes + i] = h > l ? 2u << l : 0u;
    }
    // totals: one atomic per wave and word
    uint32_t hmax = h;
    for (int off = 32; off > 0; off >>= 1) {
        nodes += (uint32_t)__shfl_xor((int)nodes, off);
        leaves += (uint32_t)__shfl_xor((int)leaves, off);
        arena += ((unsigned long long)(uint32_t)__shfl_xor((int)(arena >> 32), off) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)arena, off);
        hmax = max(hmax, (uint32_t)__shfl_xor((int)hmax, off));
    }
    if (__lane_id() == 0 && nodes != 0u) {
        atomicAdd(a.header + kHdrNodes, nodes);
        atomicAdd(a.header + kHdrLeaves, leaves);
        atomicAdd(reinterpret_cast<unsigned long long*>(a.header + kHdrArenaLo), arena);
        atomicMax(a.header + kHdrLevels, hmax);
    }
}

// The scans of the spatial refinement: as k_exclusive_scan, with the element count read from the header (the host does
// not know it yet). n = min(header[word] * per_unit, cap); the total also goes to *total_out when given.
__global__ __launch_bounds__(1024) void k_exclusive_scan_dev(const uint32_t* in, uint32_t* out, const uint32_t* header, uint32_t word, uint32_t per_unit, uint32_t cap, uint32_t* total_out) {
    const unsigned long long want = (unsigned long long)header[word] * per_unit;
    block_exclusive_scan(in, out, want < cap ? (uint32_t)want : cap, total_out);
}

PT_D bool stree_fits(const StreeArgs& a) {
    const unsigned long long arena = ((unsigned long long)a.header[kHdrArenaHi] << 32) | a.header[kHdrArenaLo];
    return (a.header[kHdrFlags] & kStreeTooDeep) == 0u && a.header[kHdrNodes] <= a.cap_nodes && a.header[kHdrLeaves] <= a.cap_leaves && arena <= a.cap_arena &&
           arena <= 0xfffffff0ull;
}

// One thread per (level, old node): level 0 is the old node itself, level L >= 1 its block of 2^L appended nodes.
__global__ __launch_bounds__(kBlock) void k_stree_append(StreeArgs a) {
    const uint32_t levels = a.header[kHdrLevels];
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {  // the capacity verdict, once
        const unsigned long long arena = ((unsigned long long)a.header[kHdrArenaHi] << 32) | a.header[kHdrArenaLo];
        uint32_t f = 0;
        if (a.header[kHdrNodes] > a.cap_nodes) f |= kStreeNodesOver;
        if (a.header[kHdrLeaves] > a.cap_leaves) f |= kStreeLeavesOver;
        if (arena > a.cap_arena) f |= kStreeArenaOver;
        if (arena > 0xfffffff0ull) f |= kStreeArenaBeyondU32;
        if (f) atomicOr(a.header + kHdrFlags, f);
    }
    if (!stree_fits(a)) return;
    const unsigned long long item = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t level = (uint32_t)(item / a.n_old_nodes), i = (uint32_t)(item % a.n_old_nodes);
    if (level > levels) return;
    const uint4 n = a.old_nodes[i];
    const uint32_t h = a.halvings_of[i];
    const auto start_of = [&](uint32_t l) { return a.n_old_nodes + a.block_start[(size_t)(l - 1u) * a.n_old_nodes + i]; };  // first node of leaf i's level-l block
    if (level == 0u) {
        if (blockIdx.y != 0u) return;
        if (n.x != 0u) {
            a.new_nodes[i] = make_uint4(n.x, n.y, n.z, 0xffffffffu);
            a.is_leaf[i] = 0u;
            a.copy_size[i] = 0u;
        } else if (h == 0u) {
            a.new_nodes[i] = make_uint4(0u, 0u, n.z, 0u);
            a.origin[i] = make_uint2(n.w, 0u);
            a.is_leaf[i] = 1u;
            a.copy_size[i] = a.old_size[n.w];
        } else {
            const uint32_t c = start_of(1u);
            a.new_nodes[i] = make_uint4(c, c + 1u, n.z, 0xffffffffu);
            a.is_leaf[i] = 0u;
            a.copy_size[i] = 0u;
        }
        return;
    }
    if (n.x != 0u || h < level) return;
    const uint32_t base = start_of(level), axis = (n.z + level) % 3u, size = a.old_size[n.w];
    const uint32_t below = h > level ? start_of(level + 1u) : 0u;
    for (uint32_t t = blockIdx.y; t < (1u << level); t += gridDim.y) {  // (gridDim.y > 1 while the tree is young: one leaf becomes a thousand)
        const uint32_t idx = base + t;
        if (h > level) {
            a.new_nodes[idx] = make_uint4(below + 2u * t, below + 2u * t + 1u, axis, 0xffffffffu);
            a.is_leaf[idx] = 0u;
            a.copy_size[idx] = 0u;
        } else {
            a.new_nodes[idx] = make_uint4(0u, 0u, axis, 0u);
            a.origin[idx] = make_uint2(n.w, level);
            a.is_leaf[idx] = 1u;
            a.copy_size[idx] = size;
        }
    }
}

__global__ __launch_bounds__(kBlock) void k_stree_finalize(StreeArgs a) {
    if (a.header[kHdrFlags]) return;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.header[kHdrNodes] || !a.is_leaf[i]) return;
    const uint32_t id = a.leaf_rank[i], off = a.copy_start[i];
    const uint2 o = a.origin[i];
    a.new_nodes[i].y = off;  // (the zero-mean bit comes from k_refine_patch_words)
    a.new_nodes[i].w = id;
    a.src_leaf[id] = o.x;
    a.halvings[id] = o.y;
    a.s_off[id] = off;
}

// The descent hands the leaf's sampler word (sampling-arena offset | mean_is_zero << 31) out of the spatial nodes and
// the jump table; the host filled in the offsets, the zero-mean bits are known only here.
__global__ __launch_bounds__(kBlock) void k_refine_patch_words(const uint4* leaf_info, uint4* snodes, uint32_t n_snodes, uint2* top, uint32_t n_top) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_snodes) {
        const uint4 n = snodes[i];
        if (n.x == 0u && leaf_info[n.w].z) snodes[i].y = n.y | 0x80000000u;
    }
    if (i < n_top) {
        const uint2 e = top[i];
        if ((e.x & 0x80000000u) && leaf_info[(e.x >> 5) & 0x3ffffffu].z) top[i].y = e.y | 0x80000000u;
    }
}

// Jump table over the first kStreeTopLevels decisions of the spatial descent (stree_leaf, pt_guided.h): entry `code`
// (one bit per decision, first decision in the top bit) names where that prefix of decisions ends: a leaf reached
// after `depth` <= kStreeTopLevels steps — {1 << 31 | leaf << 5 | depth, the leaf's sampler word} — or the interior node
// standing at depth kStreeTopLevels — {node << 5 | depth, 0}. One thread per entry follows the node links.
__global__ __launch_bounds__(kBlock) void k_build_top(const uint4* snodes, uint2* top, const uint32_t* header) {
    const uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;
    if (code >= (1u << kStreeTopLevels)) return;
    if (header && header[kHdrFlags]) return;  // the node array was not written (it did not fit: the host grows it and repeats)
    uint32_t ni = 0, depth = 0;
    for (;;) {
        const uint4 n = snodes[ni];
        if (n.x == 0u) {
            top[code] = make_uint2(0x80000000u | (n.w << 5) | depth, n.y);
            return;
        }
        if (depth == kStreeTopLevels) {
            top[code] = make_uint2((ni << 5) | depth, 0u);
            return;
        }
        ni = ((code >> (kStreeTopLevels - 1u - depth)) & 1u) ? n.y : n.x;
        depth++;
    }
}

}  // namespace pt
