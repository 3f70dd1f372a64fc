// sdtree_host.h — host side of the guided integrator's SD-tree (Müller et al.): the structure lives here between
// waves, is refined here ("between training iterations the tree is refined on the host") and is flattened into
// two arenas for the device: a read-only sampling copy and a collector that the record kernel splats into.
// Restates pt/src/integrator/path_guiding/stree.rs:21-101,255-260 and dtree.rs:185-198,291-328,378-418.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include <new>

#include "pt_types.h"

namespace pt {

// Page-locked host memory for everything that crosses PCIe between waves (the flattened tree going up, the collector
// statistics coming down): pageable copies are staged by the runtime at a few GB/s, pinned ones run at link speed and
// are truly asynchronous. Default construction is a no-op so that resize() does not zero-fill what flatten overwrites.
template <class T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <class U>
    PinnedAlloc(const PinnedAlloc<U>&) {}
    T* allocate(size_t n) {
        void* p = nullptr;
        if (hipHostMalloc(&p, n * sizeof(T), hipHostMallocDefault) != hipSuccess) throw std::bad_alloc();
        return static_cast<T*>(p);
    }
    void deallocate(T* p, size_t) { (void)hipHostFree(p); }
    template <class U>
    void construct(U*) noexcept {}
    template <class U, class A, class... Rest>
    void construct(U* p, A&& a, Rest&&... rest) {
        ::new ((void*)p) U(static_cast<A&&>(a), static_cast<Rest&&>(rest)...);
    }
    template <class U>
    bool operator==(const PinnedAlloc<U>&) const { return true; }
    template <class U>
    bool operator!=(const PinnedAlloc<U>&) const { return false; }
};
template <class T>
using PinnedVec = std::vector<T, PinnedAlloc<T>>;

struct HostDNode {  // DTreeNode (dtree.rs:28-32): 4 child sums + 4 child links (0 = leaf child)
    float rad[4] = {0, 0, 0, 0};
    uint16_t child[4] = {0, 0, 0, 0};
    float sum() const { return ((0.0f + rad[0]) + rad[1] + rad[2]) + rad[3]; }
};
struct HostDTree {
    float weight = 0.0f;  // statistical_weight
    std::vector<HostDNode> nodes;
    std::vector<uint16_t> free_slots;
    void reset_root() {
        weight = 0.0f;
        nodes.assign(1, HostDNode{});
        free_slots.clear();
    }
};
struct HostSNode {  // STreeNode (stree.rs:12-19); the sampling DTree lives only in the flattened arena (SdTreeFlat)
    HostDTree collector;
    float sampler_weight = 0.0f;  // statistical_weight of the sampling DTree
    uint32_t left = 0, right = 0;  // 0 = leaf
    uint32_t axis = 0, depth = 0;
    int64_t tree_source = -1;  // during new_iteration: node whose quadtrees a freshly split leaf still has to copy
};

// Flat device image of the tree (see pt_guided.h for how the kernels read it).
struct SdTreeFlat {
    PinnedVec<uint4> snodes;            // {left, right, axis, leaf_id}; a leaf keeps its sampler word (below) in .y
    PinnedVec<uint2> top;               // kStreeTopLevels-bit descent code -> packed node / leaf (pt_guided.h: stree_leaf)
    PinnedVec<uint4> leaf_info;         // per leaf: {sampler node offset, collector node offset, mean_is_zero, 0}
    PinnedVec<float4> s_rad;            // sampler arena
    PinnedVec<uint2> s_child;
    PinnedVec<uint2> c_child;           // collector arena topology (sums start at zero on the device)
    std::vector<uint32_t> leaf_snode;   // leaf id -> STree node index
    uint32_t n_collector_nodes = 0;
};

// The spatial tree always halves its box at the midpoint and cycles the axes with the depth (stree.rs:55-101), so
// the first kStreeTopLevels left/right decisions of a descent depend on the point and the root box only. The device
// computes them in registers and jumps to the node they lead to through this table: entry = depth | payload << 5 |
// is_leaf << 31, payload = leaf id for a leaf reached at depth <= kStreeTopLevels, else the node index at that depth.
// (kStreeTopLevels lives in pt_types.h: host and device share it)

struct HostSdTree {
    std::vector<HostSNode> nodes;
    Box3 bounds;
    // STree::new (stree.rs:21-34): cube around the scene bounds' centroid with side = largest extent.
    void init(V3 scene_lo, V3 scene_hi);
    // STree::new_iteration (stree.rs:42-54: subdivide_nodes, then per leaf build -> emit_into(sampler) ->
    // subdivide_and_clear) when `refine` is set, and the flat device image of the resulting tree in any case.
    // Returns false (and sets err) if a DTree would need more than 65535 nodes (u16 links, dtree.rs:18,395).
    bool refine_and_flatten(SdTreeFlat& out, bool refine, float directional_threshold, float spatial_threshold, uint32_t iteration, std::string* err);
    // Collector statistics as the device accumulated them: stats = [rad of every collector node (4 floats each) ...,
    // weight of every leaf ...] in the layout flatten() produced.
    void load_collector_stats(const SdTreeFlat& layout, const float* stats);
    uint64_t total_sampler_nodes() const;
    uint64_t sampler_nodes_total = 0;
};

}  // namespace pt

namespace pt {

// ---------------------------------------------------------------------------------------------------------------
// Host side when the directional quadtrees are refined on the device (pt_refine.h): only the spatial binary tree lives
// here. Between waves the host learns each leaf's statistical weight and collector size, runs STree::subdivide_nodes
// (stree.rs:55-101) on the weights, renumbers the leaves and tells the device which old leaf each new leaf copies.
struct SpatialNode {
    uint32_t left = 0, right = 0;  // 0 = leaf
    uint32_t axis = 0;
    float weight = 0.0f;           // collector statistical_weight (leaves)
    uint32_t old_leaf = 0;         // leaf id in the generation just trained (leaves)
    uint32_t halvings = 0;         // spatial splits since then
    uint32_t leaf_id = 0;          // leaf id in the new generation (leaves)
};
struct SpatialFlat {               // what goes up to the device (page-locked)
    PinnedVec<uint4> snodes;       // {left, right, axis, leaf_id}; leaves: .y = sampling-arena offset (zero-mean bit patched on the device)
    PinnedVec<uint2> top;          // jump table: {entry, sampling-arena offset}
    PinnedVec<uint32_t> src_leaf, halvings, s_off;
    uint64_t sampler_nodes = 0;
};
struct HostSpatialTree {
    std::vector<SpatialNode> nodes;
    Box3 bounds;
    uint32_t n_leaves = 1;
    void init(V3 scene_lo, V3 scene_hi);
    // weights / sizes: per leaf of the generation just trained (null for the very first image: one untrained leaf of one node)
    void next_generation(const float* weights, const uint32_t* sizes, bool refine, float spatial_threshold, uint32_t iteration, SpatialFlat& out);
};

}  // namespace pt
