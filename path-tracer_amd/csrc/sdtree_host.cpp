// sdtree_host.cpp — see sdtree_host.h. Host code only.
#include "sdtree_host.h"

#include <cmath>

namespace pt {

// STree::new (stree.rs:21-34): cube around the scene bounds' centroid with side = largest extent.
void HostSpatialTree::init(V3 lo, V3 hi) {
    nodes.assign(1, SpatialNode{});
    n_leaves = 1;
    const Box3 scene{lo, hi};
    bounds = box_around(box_centroid(scene), splat(max_comp(box_extent(scene))));
}

void HostSpatialTree::next_generation(const float* weights, const uint32_t* sizes, bool refine, float spatial_threshold, uint32_t iteration, SpatialFlat& out) {
    // leaves of the trained generation: take their weights, forget earlier split bookkeeping
    for (SpatialNode& n : nodes)
        if (n.left == 0) {
            n.old_leaf = n.leaf_id;
            n.halvings = 0;
            n.weight = weights ? weights[n.leaf_id] : 0.0f;
        }
    if (refine) {
        // subdivide_nodes (stree.rs:55-101): threshold = base * sqrt(2^iteration); freshly appended children are visited
        // by the same sweep and may split again; both children are copies of the parent with half its weight
        float pow2 = 1.0f;
        for (uint32_t k = 0; k < iteration; k++) pow2 *= 2.0f;
        const float threshold = spatial_threshold * sqrtf(pow2);
        for (size_t i = 0; i < nodes.size(); i++) {
            if (nodes[i].left != 0 || !(nodes[i].weight > threshold)) continue;
            SpatialNode child;
            child.axis = (nodes[i].axis + 1) % 3;
            child.weight = nodes[i].weight / 2.0f;
            child.old_leaf = nodes[i].old_leaf;
            child.halvings = nodes[i].halvings + 1;
            nodes[i].left = (uint32_t)nodes.size();
            nodes[i].right = nodes[i].left + 1;
            nodes.push_back(child);
            nodes.push_back(child);
        }
    }
    // numbering, sampling-arena offsets (the sampling copy of a leaf is its source's collector as trained)
    out.snodes.resize(nodes.size());
    out.src_leaf.clear();
    out.halvings.clear();
    out.s_off.clear();
    uint64_t s_total = 0;
    for (size_t i = 0; i < nodes.size(); i++) {
        SpatialNode& n = nodes[i];
        if (n.left != 0) {
            out.snodes[i] = make_uint4(n.left, n.right, n.axis, 0xffffffffu);
            continue;
        }
        n.leaf_id = (uint32_t)out.src_leaf.size();
        out.snodes[i] = make_uint4(0, (uint32_t)s_total, n.axis, n.leaf_id);
        out.src_leaf.push_back(n.old_leaf);
        out.halvings.push_back(n.halvings);
        out.s_off.push_back((uint32_t)s_total);
        s_total += sizes ? sizes[n.old_leaf] : 1u;
    }
    n_leaves = (uint32_t)out.src_leaf.size();
    out.sampler_nodes = s_total;
}

}  // namespace pt
