// sdtree_host.h — host side of the guided integrator's SD-tree (Müller et al.). The whole tree lives and is refined on
// the device (pt_refine.h: the spatial sweep in closed form, the directional quadtrees per leaf); the host keeps the
// tree's bounds, builds the one-node tree of wave 0, and holds the reference's spatial sweep AS WRITTEN —
// STree::subdivide_nodes (pt/src/integrator/path_guiding/stree.rs:55-101), a loop that appends children and meets them
// again — as the yardstick the device's result is compared with word by word under PTGPU_CHECK_STREE (a test switch:
// the weights come down for it alone).
#pragma once
#include <cstdint>
#include <vector>

#include "pt_pinned.h"
#include "pt_types.h"

namespace pt {

struct SpatialNode {  // STreeNode (stree.rs:12-19) without its DTrees
    uint32_t left = 0, right = 0;  // 0 = leaf
    uint32_t axis = 0;
    float weight = 0.0f;           // collector statistical_weight (leaves)
    uint32_t old_leaf = 0;         // leaf id in the generation just trained (leaves)
    uint32_t halvings = 0;         // spatial splits since then
    uint32_t leaf_id = 0;          // leaf id in the new generation (leaves)
};
struct SpatialFlat {               // what goes up to the device (page-locked)
    PinnedVec<uint4> snodes;       // {left, right, axis, leaf_id}; leaves: .y = sampling-arena offset (zero-mean bit patched on the device)
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
