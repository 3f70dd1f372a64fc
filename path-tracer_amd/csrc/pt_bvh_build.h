// pt_bvh_build.h — Bvh::new (pt/src/bvh.rs:48-85,148-304) on the device, producing the SAME tree as
// bvh_build.cpp word for word: same 12-bucket SAH decisions, same itertools::partition permutation, same depth-first
// node numbering. What makes that possible in parallel:
//   * bounds, centroid bounds and bucket bounds are min/max reductions and the bucket counts are sums of 1.0f below
//     2^24: exact, so their order does not matter (inputs with a negative zero — where fminf's pick between -0 and +0
//     would depend on the order — and inputs that are not finite go to the host builder);
//   * the SAH sweep over the twelve buckets is run by one thread in the reference's order (its NaNs included: an empty
//     prefix has area inf * 0);
//   * itertools::partition, a sequential two-pointer swap loop, has a closed form: with `mid` = number of elements that
//     pass, the k-th failing element (from the front) among the first `mid` positions changes places with the k-th
//     passing element (from the back) among the rest, everything else stays;
//   * the tree is grown level by level (one workgroup per node of the level, children allocated in pairs), then
//     numbered: subtree sizes bottom-up, pre-order positions top-down (left child = parent + 1, right child = parent +
//     1 + size of the left subtree); a leaf's first triangle is its range's start, because leaves come out in array order.
#pragma once
#include "pt_math.h"
#include "pt_types.h"

namespace pt {

constexpr int kSahBuckets = 12;  // N_SAH_BUCKETS (bvh.rs:148)
constexpr uint32_t kBuildBlock = 256;

struct BuildNode {  // level order; the children of a node are a pair (left, left + 1)
    float lo[3], hi[3];
    uint32_t begin, end;  // triangle range
    uint32_t left;        // 0 = leaf
    uint32_t axis;
    uint32_t count;       // nodes of the subtree
    uint32_t pre;         // depth-first position
};

PT_HD int sah_bucket_of(const Box3& cb, V3 centroid, int axis) {  // bvh.rs:154-156 (`as usize` saturates, NaN -> 0)
    const float rel = get((centroid - cb.lo) / box_extent(cb), axis);
    const float s = (float)kSahBuckets * rel;
    const long b = (s > 0.0f) ? (s >= 9.0e18f ? (long)kSahBuckets : (long)s) : 0;
    return (int)(b < (long)(kSahBuckets - 1) ? b : (long)(kSahBuckets - 1));
}

// float <-> unsigned key with the same order (finite values and infinities, no negative zero among the inputs)
PT_HD uint32_t order_key(float f) {
    const uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
PT_HD float order_unkey(uint32_t k) { return u2f((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

struct BuildArgs {
    float4* tri_lo;  // bounds.lo | input id
    float4* tri_hi;  // bounds.hi | -
    BuildNode* nodes;
    uint32_t* n_nodes;     // allocation counter
    uint32_t max_nodes;
    uint32_t* list_a;      // per triangle position: scratch of the partition (failing positions of the left part)
    uint32_t* list_b;      //                                                 (passing positions of the right part)
    uint32_t* error_flag;  // 1 = node array full, 2 = partition lists disagree (cannot happen)
};

PT_D Box3 tri_box(const BuildArgs& a, uint32_t i) {
    const float4 l = a.tri_lo[i], h = a.tri_hi[i];
    return Box3{mk3(l.x, l.y, l.z), mk3(h.x, h.y, h.z)};
}

// Exclusive rank of the threads with `flag` inside the workgroup + the workgroup's total (every thread calls it).
PT_D uint32_t block_rank(uint32_t* sm_wave, bool flag, uint32_t& total) {
    const uint32_t lane = __lane_id(), wave = threadIdx.x / 64u;
    const unsigned long long m = __ballot(flag);
    __syncthreads();  // sm_wave may still be read from the previous call
    if (lane == 0) sm_wave[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0;
    total = 0;
    for (uint32_t w = 0; w < kBuildBlock / 64u; w++) {
        const uint32_t c = sm_wave[w];
        if (w < wave) before += c;
        total += c;
    }
    return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// One workgroup per node of the current level: leaf or split (bvh.rs:158-304).
__global__ __launch_bounds__(kBuildBlock) void k_bvh_split(BuildArgs a, uint32_t level_first) {
    __shared__ uint32_t sm_box[6];
    __shared__ uint32_t sm_cb[6];
    __shared__ uint32_t sm_bucket[kSahBuckets][6];
    __shared__ uint32_t sm_count[kSahBuckets];
    __shared__ uint32_t sm_wave[kBuildBlock / 64u];
    __shared__ int sm_best;    // -1 = leaf
    __shared__ uint32_t sm_mid;
    BuildNode& nd = a.nodes[level_first + blockIdx.x];
    const uint32_t begin = nd.begin, end = nd.end, n = end - begin, tid = threadIdx.x;
    if (tid < 3) {
        sm_box[tid] = sm_cb[tid] = order_key(f_inf());
        sm_box[3 + tid] = sm_cb[3 + tid] = order_key(-f_inf());
    }
    for (uint32_t k = tid; k < kSahBuckets * 6u; k += kBuildBlock) sm_bucket[k / 6u][k % 6u] = order_key((k % 6u) < 3u ? f_inf() : -f_inf());
    if (tid < (uint32_t)kSahBuckets) sm_count[tid] = 0;
    if (tid == 0) sm_mid = 0;
    __syncthreads();
    {   // bounds of the triangles, bounds of their centroids
        Box3 b = box_empty(), cb = box_empty();
        for (uint32_t i = begin + tid; i < end; i += kBuildBlock) {
            const Box3 t = tri_box(a, i);
            b = box_join(b, t);
            cb = box_join(cb, box_centroid(t));
        }
        if (begin + tid < end) {
            atomicMin(&sm_box[0], order_key(b.lo.x)); atomicMin(&sm_box[1], order_key(b.lo.y)); atomicMin(&sm_box[2], order_key(b.lo.z));
            atomicMax(&sm_box[3], order_key(b.hi.x)); atomicMax(&sm_box[4], order_key(b.hi.y)); atomicMax(&sm_box[5], order_key(b.hi.z));
            atomicMin(&sm_cb[0], order_key(cb.lo.x)); atomicMin(&sm_cb[1], order_key(cb.lo.y)); atomicMin(&sm_cb[2], order_key(cb.lo.z));
            atomicMax(&sm_cb[3], order_key(cb.hi.x)); atomicMax(&sm_cb[4], order_key(cb.hi.y)); atomicMax(&sm_cb[5], order_key(cb.hi.z));
        }
    }
    __syncthreads();
    const Box3 bounds{mk3(order_unkey(sm_box[0]), order_unkey(sm_box[1]), order_unkey(sm_box[2])), mk3(order_unkey(sm_box[3]), order_unkey(sm_box[4]), order_unkey(sm_box[5]))};
    const Box3 cb{mk3(order_unkey(sm_cb[0]), order_unkey(sm_cb[1]), order_unkey(sm_cb[2])), mk3(order_unkey(sm_cb[3]), order_unkey(sm_cb[4]), order_unkey(sm_cb[5]))};
    if (tid == 0) {
        nd.lo[0] = bounds.lo.x; nd.lo[1] = bounds.lo.y; nd.lo[2] = bounds.lo.z;
        nd.hi[0] = bounds.hi.x; nd.hi[1] = bounds.hi.y; nd.hi[2] = bounds.hi.z;
        nd.left = 0;
        nd.axis = 0;
    }
    if (n == 1 || box_area(bounds) == 0.0f) return;  // leaf (workgroup-uniform)
    const int axis = max_axis(box_extent(cb));
    for (uint32_t i = begin + tid; i < end; i += kBuildBlock) {
        const Box3 t = tri_box(a, i);
        const int b = sah_bucket_of(cb, box_centroid(t), axis);
        atomicMin(&sm_bucket[b][0], order_key(t.lo.x)); atomicMin(&sm_bucket[b][1], order_key(t.lo.y)); atomicMin(&sm_bucket[b][2], order_key(t.lo.z));
        atomicMax(&sm_bucket[b][3], order_key(t.hi.x)); atomicMax(&sm_bucket[b][4], order_key(t.hi.y)); atomicMax(&sm_bucket[b][5], order_key(t.hi.z));
        atomicAdd(&sm_count[b], 1u);
    }
    __syncthreads();
    if (tid == 0) {  // the sweep, in the reference's order (bvh.rs:218-252)
        Box3 bb[kSahBuckets];
        float cnt[kSahBuckets];
        for (int b = 0; b < kSahBuckets; b++) {
            bb[b] = Box3{mk3(order_unkey(sm_bucket[b][0]), order_unkey(sm_bucket[b][1]), order_unkey(sm_bucket[b][2])),
                         mk3(order_unkey(sm_bucket[b][3]), order_unkey(sm_bucket[b][4]), order_unkey(sm_bucket[b][5]))};
            cnt[b] = (float)sm_count[b];
        }
        float cost[kSahBuckets - 1];
        {
            Box3 acc = box_empty();
            float run = 0.0f;
            for (int i = 0; i < kSahBuckets - 1; i++) {
                acc = box_join(acc, bb[i]);
                run += cnt[i] * box_area(acc);
                cost[i] = 0.0f + run;
            }
        }
        {
            Box3 acc = box_empty();
            float run = 0.0f;
            for (int i = kSahBuckets - 1; i >= 1; i--) {
                acc = box_join(acc, bb[i]);
                run += cnt[i] * box_area(acc);
                cost[i - 1] += run;
            }
        }
        int best = 0;
        float best_cost = f_inf();
        for (int i = 0; i < kSahBuckets - 1; i++)
            if (cost[i] < best_cost) {
                best = i;
                best_cost = cost[i];
            }
        const float leaf_cost = (float)n;
        const float split_cost = 0.5f + best_cost / box_area(bounds);
        sm_best = leaf_cost < split_cost ? -1 : best;
    }
    __syncthreads();
    const int best = sm_best;
    if (best < 0) return;  // leaf
    // ---- itertools::partition in closed form
    uint32_t mine = 0;
    for (uint32_t i = begin + tid; i < end; i += kBuildBlock) mine += sah_bucket_of(cb, box_centroid(tri_box(a, i)), axis) <= best ? 1u : 0u;
    if (mine) atomicAdd(&sm_mid, mine);
    __syncthreads();
    const uint32_t mid = sm_mid;
    uint32_t n_a = 0, n_b = 0;
    for (uint32_t base = 0; base < mid; base += kBuildBlock) {  // failing elements of the left part, front to back
        const uint32_t i = begin + base + tid;
        const bool fails = base + tid < mid && !(sah_bucket_of(cb, box_centroid(tri_box(a, i)), axis) <= best);
        uint32_t total;
        const uint32_t r = block_rank(sm_wave, fails, total);
        if (fails) a.list_a[begin + n_a + r] = i;
        n_a += total;
    }
    for (uint32_t base = 0; base < n - mid; base += kBuildBlock) {  // passing elements of the right part, back to front
        const bool in = base + tid < n - mid;
        const uint32_t i = end - 1u - (base + tid);
        const bool passes = in && sah_bucket_of(cb, box_centroid(tri_box(a, i)), axis) <= best;
        uint32_t total;
        const uint32_t r = block_rank(sm_wave, passes, total);
        if (passes) a.list_b[begin + n_b + r] = i;
        n_b += total;
    }
    if (n_a != n_b && tid == 0) atomicMax(a.error_flag, 2u);
    __syncthreads();  // the lists are in memory for every thread of the workgroup
    for (uint32_t k = tid; k < n_a && k < n_b; k += kBuildBlock) {
        const uint32_t x = a.list_a[begin + k], y = a.list_b[begin + k];
        const float4 xl = a.tri_lo[x], xh = a.tri_hi[x];
        a.tri_lo[x] = a.tri_lo[y];
        a.tri_hi[x] = a.tri_hi[y];
        a.tri_lo[y] = xl;
        a.tri_hi[y] = xh;
    }
    if (mid == 0 || mid == n - 1) return;  // leaf all the same (sic, bvh.rs:257), in the permuted order
    if (tid == 0) {
        const uint32_t first = atomicAdd(a.n_nodes, 2u);
        if (first + 2u > a.max_nodes) {
            atomicMax(a.error_flag, 1u);
        } else {
            BuildNode l{}, r{};
            l.begin = begin; l.end = begin + mid;
            r.begin = begin + mid; r.end = end;
            a.nodes[first] = l;
            a.nodes[first + 1] = r;
            nd.left = first;
            nd.axis = (uint32_t)axis;
        }
    }
}

// Subtree sizes, deepest level first; depth-first positions, root first (one thread per node of the level).
__global__ void k_bvh_count(BuildNode* nodes, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    BuildNode& nd = nodes[first + i];
    nd.count = nd.left ? 1u + nodes[nd.left].count + nodes[nd.left + 1].count : 1u;
}
__global__ void k_bvh_number(BuildNode* nodes, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BuildNode& nd = nodes[first + i];
    if (!nd.left) return;
    nodes[nd.left].pre = nd.pre + 1u;
    nodes[nd.left + 1].pre = nd.pre + 1u + nodes[nd.left].count;
}
// The flattened tree in the layout ptgpu_scene_create uploads: two float4 per node {lo | offset, hi | n_tris | axis << 16}.
__global__ void k_bvh_emit(const BuildNode* nodes, uint32_t n_nodes, float4* out, uint32_t* error_flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const BuildNode& nd = nodes[i];
    const uint32_t n_tris = nd.left ? 0u : nd.end - nd.begin;
    if (n_tris > 65535u) atomicMax(error_flag, 3u);  // n_tris is u16 in the reference
    const uint32_t offset = nd.left ? nodes[nd.left + 1].pre : nd.begin;
    out[2 * (size_t)nd.pre] = make_float4(nd.lo[0], nd.lo[1], nd.lo[2], u2f(offset));
    out[2 * (size_t)nd.pre + 1] = make_float4(nd.hi[0], nd.hi[1], nd.hi[2], u2f((n_tris & 0xffffu) | (nd.left ? nd.axis << 16 : 0u)));
}

}  // namespace pt
