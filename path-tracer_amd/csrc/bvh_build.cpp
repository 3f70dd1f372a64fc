// bvh_build.cpp — see bvh_build.h. Host code only (compiled by hipcc for the shared math header).
#include "bvh_build.h"

#include <algorithm>

namespace pt {
namespace {

constexpr int kBuckets = 12;  // N_SAH_BUCKETS (bvh.rs:148)

inline int bucket_of(const Box3& cb, V3 centroid, int axis) {  // bvh.rs:154-156 (`as usize` saturates, NaN -> 0)
    const float rel = get((centroid - cb.lo) / box_extent(cb), axis);
    const float s = (float)kBuckets * rel;
    long b = (s > 0.0f) ? (s >= 9.0e18f ? (long)kBuckets : (long)s) : 0;
    return (int)std::min<long>(kBuckets - 1, b);
}

struct Builder {
    std::vector<BuildTri>& tris;
    FlatBvh& out;
    bool ok = true;

    uint32_t emit_leaf(const Box3& bounds, size_t begin, size_t end, uint32_t index) {
        FlatNode& n = out.nodes[index];
        n.bounds = bounds;
        n.offset = (uint32_t)out.leaf_order.size();
        if (end - begin > 65535) ok = false;
        n.n_tris = (uint16_t)(end - begin);
        n.split_axis = 0;
        for (size_t i = begin; i < end; i++) out.leaf_order.push_back(tris[i].id);
        return index;
    }

    // Builds the subtree over tris[begin, end) and returns its node index (pre-order position).
    uint32_t build(size_t begin, size_t end) {
        const uint32_t index = (uint32_t)out.nodes.size();
        out.nodes.push_back(FlatNode{box_empty(), 0, 0, 0});
        const size_t n = end - begin;
        Box3 bounds = box_empty();
        for (size_t i = begin; i < end; i++) bounds = box_join(bounds, tris[i].bounds);
        if (n == 1 || box_area(bounds) == 0.0f) return emit_leaf(bounds, begin, end, index);

        Box3 cb = box_empty();
        for (size_t i = begin; i < end; i++) cb = box_join(cb, box_centroid(tris[i].bounds));
        const int axis = max_axis(box_extent(cb));

        Box3 bucket_bounds[kBuckets];
        float bucket_count[kBuckets];
        for (int b = 0; b < kBuckets; b++) {
            bucket_bounds[b] = box_empty();
            bucket_count[b] = 0.0f;
        }
        for (size_t i = begin; i < end; i++) {
            const int b = bucket_of(cb, box_centroid(tris[i].bounds), axis);
            bucket_bounds[b] = box_join(bucket_bounds[b], tris[i].bounds);
            bucket_count[b] += 1.0f;
        }
        // cost of splitting after bucket i = sum_below(count * area so far) + sum_above(...) (bvh.rs:218-236)
        float cost[kBuckets - 1];
        {
            Box3 acc = box_empty();
            float run = 0.0f;
            for (int i = 0; i < kBuckets - 1; i++) {
                acc = box_join(acc, bucket_bounds[i]);
                run += bucket_count[i] * box_area(acc);
                cost[i] = 0.0f + run;
            }
        }
        {
            Box3 acc = box_empty();
            float run = 0.0f;
            for (int i = kBuckets - 1; i >= 1; i--) {
                acc = box_join(acc, bucket_bounds[i]);
                run += bucket_count[i] * box_area(acc);
                cost[i - 1] += run;
            }
        }
        int best = 0;
        float best_cost = f_inf();
        for (int i = 0; i < kBuckets - 1; i++)
            if (cost[i] < best_cost) {
                best = i;
                best_cost = cost[i];
            }
        const float leaf_cost = (float)n;
        const float split_cost = 0.5f + best_cost / box_area(bounds);
        if (leaf_cost < split_cost) return emit_leaf(bounds, begin, end, index);

        // itertools::partition: walk from the front; a failing element is swapped with the last passing
        // element found scanning from the back; stop when the back scan meets the front.
        auto goes_left = [&](const BuildTri& t) { return bucket_of(cb, box_centroid(t.bounds), axis) <= best; };
        size_t front = begin, back = end, mid = 0;
        while (front < back) {
            BuildTri& f = tris[front++];
            if (!goes_left(f)) {
                bool swapped = false;
                while (back > front) {
                    --back;
                    if (goes_left(tris[back])) {
                        std::swap(f, tris[back]);
                        swapped = true;
                        break;
                    }
                }
                if (!swapped) break;
            }
            mid++;
        }
        if (mid == 0 || mid == n - 1) return emit_leaf(bounds, begin, end, index);  // sic (bvh.rs:257)

        build(begin, begin + mid);                     // left child = index + 1
        const uint32_t right = build(begin + mid, end);
        FlatNode& node = out.nodes[index];
        node.bounds = bounds;
        node.offset = right;
        node.n_tris = 0;
        node.split_axis = (uint8_t)axis;
        return index;
    }
};

}  // namespace

namespace {
struct WideBuilder {
    const FlatBvh& b;
    std::vector<WideNode>& out;
    bool ok = true;
    uint32_t depth = 0;

    bool leaf_ref(uint32_t node, uint32_t& ref) const {
        const FlatNode& n = b.nodes[node];
        if (n.n_tris == 0 || n.n_tris > 128 || n.offset >= (1u << 24)) return false;
        ref = 0x80000000u | ((uint32_t)(n.n_tris - 1) << 24) | n.offset;
        return true;
    }
    // visit order of the frontier under `node` for one octant, as the binary walk would produce it
    void order_of(uint32_t node, uint32_t octant, const std::vector<uint32_t>& frontier, std::vector<uint32_t>& seq) const {
        for (size_t s = 0; s < frontier.size(); s++)
            if (frontier[s] == node) {
                seq.push_back((uint32_t)s);
                return;
            }
        const FlatNode& n = b.nodes[node];
        const bool neg = ((octant >> n.split_axis) & 1u) != 0u;
        order_of(neg ? n.offset : node + 1, octant, frontier, seq);
        order_of(neg ? node + 1 : n.offset, octant, frontier, seq);
    }
    uint32_t collapse(uint32_t node, uint32_t level) {
        const uint32_t index = (uint32_t)out.size();
        out.push_back(WideNode{});
        depth = std::max(depth, level);
        std::vector<uint32_t> frontier;
        if (b.nodes[node].n_tris != 0) frontier = {node};  // a tree that is one leaf
        else {
            frontier = {node + 1, b.nodes[node].offset};
            while (frontier.size() < 4) {
                int pick = -1;
                float best = -1.0f;
                for (size_t s = 0; s < frontier.size(); s++) {
                    const FlatNode& n = b.nodes[frontier[s]];
                    if (n.n_tris != 0) continue;
                    const float a = box_area(n.bounds);
                    if (a > best) {
                        best = a;
                        pick = (int)s;
                    }
                }
                if (pick < 0) break;
                const uint32_t x = frontier[pick];
                frontier[pick] = x + 1;
                frontier.insert(frontier.begin() + pick + 1, b.nodes[x].offset);
            }
        }
        WideNode w{};
        w.n_children = (uint32_t)frontier.size();
        for (int c = 0; c < 4; c++)
            for (int a = 0; a < 3; a++) {
                w.lo[a][c] = f_inf();
                w.hi[a][c] = -f_inf();
            }
        for (uint32_t octant = 0; octant < 8; octant++) {
            std::vector<uint32_t> seq;
            order_of(node, octant, frontier, seq);
            uint32_t byte = 0;
            for (size_t k = 0; k < seq.size(); k++) byte |= seq[k] << (2 * k);
            w.order[octant / 4] |= byte << (8 * (octant % 4));
        }
        for (size_t s = 0; s < frontier.size(); s++) {
            const FlatNode& n = b.nodes[frontier[s]];
            for (int a = 0; a < 3; a++) {
                w.lo[a][s] = get(n.bounds.lo, a);
                w.hi[a][s] = get(n.bounds.hi, a);
            }
            if (n.n_tris != 0) {
                if (!leaf_ref(frontier[s], w.ref[s])) ok = false;
            } else
                w.ref[s] = collapse(frontier[s], level + 1);
        }
        out[index] = w;
        return index;
    }
};
}  // namespace

bool build_wide(const FlatBvh& bvh, std::vector<WideNode>& out, uint32_t& depth_out) {
    out.clear();
    if (bvh.nodes.empty()) return false;
    WideBuilder wb{bvh, out};
    wb.collapse(0, 1);
    depth_out = wb.depth;
    return wb.ok && out.size() < (1u << 31);
}

bool build_bvh(std::vector<BuildTri>& tris, FlatBvh& out) {
    out.nodes.clear();
    out.leaf_order.clear();
    out.leaf_order.reserve(tris.size());
    if (tris.empty()) return false;
    Builder b{tris, out};
    b.build(0, tris.size());
    return b.ok;
}

}  // namespace pt
