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
// Four-wide image of the SAME tree (SURVEY 8f.3 "wider BVH", parity-exact): every wide node stands for an interior
// binary node and holds, as its children, a frontier of up to four of that node's descendants (the two children, then
// the interior child of largest area replaced by its own two children, while fewer than four). Nothing else changes:
// the leaves, their triangles and their order are the binary tree's. The reference's traversal order over a frontier is
// a function of the ray's three direction signs only (near child first by dir_is_neg[split_axis], bvh.rs:127-133), so it
// is tabulated per octant. Boxes of skipped inner nodes need no test: a child's box lies inside them and the slab
// test is monotone (pt_device.h: bvh_intersect), so the set and order of tested leaves is the binary walk's.
struct WideNode {      // 128 bytes = eight float4 on the device
    float lo[3][4];    // lo[axis][child]
    float hi[3][4];
    uint32_t ref[4];   // interior: wide node index; leaf: 1 << 31 | (n_tris - 1) << 24 | first triangle (leaf order)
    uint32_t order[2]; // octant o (bit0 = x negative, bit1 = y, bit2 = z): byte o = four 2-bit child slots in visit order
    uint32_t n_children, pad;
};
static_assert(sizeof(WideNode) == 128, "wide node layout");
// Returns false when a leaf cannot be encoded in a child reference (more than 128 triangles, or beyond 2^24): the
// caller then keeps to the binary walk. depth_out = levels of the wide tree.
bool build_wide(const FlatBvh& bvh, std::vector<WideNode>& out, uint32_t& depth_out);

// tris is permuted in place. Returns false if a leaf would exceed 65535 triangles (n_tris is u16).
bool build_bvh(std::vector<BuildTri>& tris, FlatBvh& out);

}  // namespace pt
