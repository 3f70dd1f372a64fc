// pt_types.h — plain-data layouts shared by the host side of libptgpu and its kernels.
#pragma once
#include "pt_math.h"

namespace pt {

constexpr int NSW = 8;          // N_SAMPLED_WAVELENGTHS (pt/src/spectrum/mod.rs:6)
constexpr int CIE_N = 471;      // 360..830 nm, 1 nm
constexpr uint32_t kNoHit = 0xffffffffu;
constexpr int kBlock = 256;     // threads per workgroup of every kernel
#ifndef PT_LEAF_QUEUE
#define PT_LEAF_QUEUE 4
#endif
constexpr uint32_t kLdsNodeBudget = 4096;  // bytes: node arrays up to this size are staged in LDS (pt_device.h: bvh_stage_nodes)
constexpr uint32_t kLeafQueue = PT_LEAF_QUEUE;  // leaves a ray collects before testing their triangles (pt_device.h)
#ifndef PT_STREE_TOP_LEVELS
#define PT_STREE_TOP_LEVELS 21
#endif
constexpr uint32_t kStreeTopLevels = PT_STREE_TOP_LEVELS;  // guided: levels of the spatial tree resolved through the jump table
constexpr int kBvhStack = 64;   // ArrayVec<usize, 64> (pt/src/bvh.rs:94)

enum { MAT_LAMBERTIAN = 0, MAT_DIELECTRIC = 1, MAT_CONDUCTOR = 2 };
// A triangle's tag (third w word of its record, copied into the hit record): material + 1 in the low 28 bits (0 = a
// pass-through surface), the material's kind in bits 28-29 (so that a shade kernel can group its lanes by BSDF code
// before it has loaded anything else), has_normals in bit 31.
constexpr uint32_t kTagMaterialMask = 0x0fffffffu, kTagKindShift = 28, kTagNormalsBit = 0x80000000u;
PT_HD int tag_material(uint32_t tag) { return (int)(tag & kTagMaterialMask) - 1; }
PT_HD uint32_t tag_kind(uint32_t tag) { return (tag >> kTagKindShift) & 3u; }
enum { MODE_PT = 0, MODE_NEE = 1, MODE_MIS = 2 };

struct SpectrumDev {
    float c0, c1, c2, scale;
    int illuminant;
};
struct MaterialDev {
    int kind;
    float roughness;
    SpectrumDev s0, s1;
};
// Every spectrum of the scene evaluated at the eight wavelengths of every wavelength-table row (a path's hero sample
// is one of a few hundred to a few thousand table rows): entry (spectrum, row) = two float4 at 2 * (spectrum * n_rows
// + row). Spectrum ids: material m -> 2m (albedo | eta) and 2m + 1 (conductor k); emissive mesh e -> emission_base + e.
// Built on the device by the same spectrum_eval the kernels would otherwise call per segment (same bits).
struct SpecTab {
    const float4* tab;
    uint32_t n_rows;
    uint32_t emission_base;
};
struct MeshInfo {
    int material;     // -1 = None (pass-through surface)
    int has_normals;
};

// Flattened scene in HBM. Triangles are stored in BVH leaf order (Bvh::compact_triangles order):
//   tri_v[3k+0] = {p0.xyz, bits(mesh_index)}, tri_v[3k+1] = {p1.xyz, bits(triangle_index)}, tri_v[3k+2] = {p2.xyz, 0}
//   tri_n[3k+i] = vertex normal i (only read when mesh_info[mesh].has_normals)
struct SceneDev {
    const float4* nodes;
    const float4* tri_v;
    const float4* tri_n;
    const MeshInfo* mesh_info;
    const float4* light_v;             // the emissive triangles' tri_v records again, in emissive_tris order (one hop less for NEE)
    const MaterialDev* materials;
    const SpectrumDev* emissions;      // per emissive mesh (mesh_index < n_emissive_meshes <=> emitter)
    const uint32_t* emissive_tris;     // leaf-order index of every emissive triangle, in Scene::emissive_triangles order
    const float* d65;                  // CIE_D65_F32[471]
    const double* cie_xyz;             // CIE_XYZ_F64[471][3]
    uint32_t n_nodes, n_tris, n_emissive_tris, n_emissive_meshes;
    uint32_t n_materials;
    uint32_t lds_nodes;                // != 0: the kernels copy the node array (n_nodes * 32 B, small scenes) into LDS behind the stacks (bvh_stage_nodes)
    uint32_t bvh_stack;                // LDS stack entries per ray = tree depth + 1 (kernels get 256 * bvh_stack * 4 B of dynamic LDS)
    // four-wide image of the same tree (bvh_build.h: WideNode; eight float4 per node), walked by the WIDE kernel variants
    const float4* wnodes;
    uint32_t n_wnodes;
    uint32_t wide_stack;               // LDS stack entries per ray of the wide walk = 3 * wide depth + 1
    uint32_t lds_wnodes;               // != 0: the wide node array is staged in LDS
    V3 bounds_lo, bounds_hi;
    uint32_t* guards;                  // kGuard* counters: how often a documented safety limit cut something short (ptgpu_guard_counts)
};
// Limits the reference does not have; each is counted when it actually changes a result, so tests can assert "never".
enum { kGuardPassThrough = 0,   // a path ended after kMaxPassThrough consecutive pass-through crossings (pt_state.h)
       kGuardVertexStore = 1,   // a guided path vertex beyond the vertex store's rows was dropped (cannot happen: rows = max_depth + 1), or a
                                // subtree beyond the walk stack of the spatial box filter (pt_guided.h: k_guide_splat_spatial_box; 96 levels)
       kGuardQuadLevel = 2,     // the refinement declined to split a quadtree cell at level kMaxQuadLevel (pt_guided.h)
       kGuardNonFinite = 3,     // a sample whose radiance is not finite was dropped instead of added to the film (pt_kernels.h: film_add_rgb_wave)
       kNumGuards = 4 };

struct CameraDev {
    Mat4 c2w, w2c;
    V3 p, n;  // Camera::p(), Camera::n()
    float pixel_size;
    uint32_t resx, resy;
};

}  // namespace pt
