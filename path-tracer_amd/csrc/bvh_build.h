// bvh_build.h — host-side BVH construction for libptgpu.
// Same algorithm and, deliberately, the same OUTPUT ORDER as Bvh::new (pt/src/bvh.rs:48-85,148-304): 12-bucket
// SAH on the axis of largest centroid extent, itertools::partition semantics for the in-place split, leaves that
// may hold many triangles, depth-first layout with the left child at index+1. Traversal order — and therefore
// which triangle wins an equal-t tie — depends on it, so the hit-parity tests compare this layout word for
// word with the CPU restatement's.
#pragma once
#include <vector>

#include "pt_types.h"

namespace pt {

struct BuildTri {
    Box3 bounds;
    uint32_t id;  // position in the mesh-major input order
};
struct FlatNode {
    Box3 bounds;
    uint32_t offset;      // leaf: first triangle (leaf order); interior: right child
    uint16_t n_tris;      // 0 = interior
    uint8_t split_axis;
};
struct FlatBvh {
    std::vector<FlatNode> nodes;
    std::vector<uint32_t> leaf_order;  // leaf position -> input triangle id
};
// tris is permuted in place. Returns false if a leaf would exceed 65535 triangles (n_tris is u16).
bool build_bvh(std::vector<BuildTri>& tris, FlatBvh& out);

}  // namespace pt
