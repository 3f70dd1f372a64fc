"""Synthetic scenes and ray batches shared by the CPU and GPU tests."""
from __future__ import annotations

import numpy as np
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent

import path_tracer_amd as pt

GREY = (0.0, 0.0, 0.0, 1.0)  # sigmoid(0) * 1.0 = 0.5 reflectance
IDENTITY = np.eye(4, dtype=np.float32)


def icosphere(level: int):
    """Subdivided icosahedron on the unit sphere: (points (n,3) f32, triangles (m,3) u32), outward winding."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8),
         (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = [tuple(p) for p in v]
    for _ in range(level):
        cache = {}
        nf = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = (np.array(verts[a]) + np.array(verts[b])) / 2.0
                m /= np.linalg.norm(m)
                verts.append(tuple(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(verts, dtype=np.float32), np.array(f, dtype=np.uint32)


def random_soup(rng: np.random.Generator, n_tris: int, n_meshes: int = 3, extent: float = 4.0, size: float = 0.6, emissive: bool = True):
    """Random triangle soup split over several meshes (one emissive), with shared vertices inside a mesh so that
    rays through shared edges/vertices exercise the watertight test and its f64 fallback."""
    meshes = []
    per = max(1, n_tris // n_meshes)
    for m in range(n_meshes):
        k = per if m < n_meshes - 1 else n_tris - per * (n_meshes - 1)
        centers = rng.uniform(-extent, extent, size=(k, 1, 3))
        pts = (centers + rng.normal(0, size, size=(k, 3, 3))).reshape(-1, 3).astype(np.float32)
        tris = np.arange(3 * k, dtype=np.uint32).reshape(k, 3)
        # stitch a strip: consecutive triangles share an edge
        for i in range(1, k, 2):
            tris[i, 0] = tris[i - 1, 1]
            tris[i, 1] = tris[i - 1, 2]
        mesh = {"points": pts, "triangles": tris, "material_index": 0 if m % 2 == 0 else -1}
        if m == 0 and emissive:
            mesh["emission"] = GREY
        meshes.append(mesh)
    return pt.SceneDescription(meshes, [(pt.MAT_LAMBERTIAN, 0.0, GREY, (0, 0, 0, 0))])


def grid_scene(n: int = 8):
    """Axis-aligned grid of coplanar, edge-sharing triangles (equal-t ties along shared edges, degenerate boxes)."""
    xs = np.arange(n + 1, dtype=np.float32)
    px, py = np.meshgrid(xs, xs, indexing="xy")
    pts = np.stack([px.ravel(), py.ravel(), np.zeros((n + 1) ** 2, dtype=np.float32)], axis=1)
    tris = []
    for y in range(n):
        for x in range(n):
            a = y * (n + 1) + x
            tris += [(a, a + 1, a + n + 2), (a, a + n + 2, a + n + 1)]
    # a second copy of the same plane: exact duplicates, every hit is an equal-t tie between meshes
    meshes = [{"points": pts, "triangles": np.array(tris, dtype=np.uint32), "material_index": 0},
              {"points": pts.copy(), "triangles": np.array(tris, dtype=np.uint32), "material_index": 0, "emission": GREY}]
    return pt.SceneDescription(meshes, [(pt.MAT_LAMBERTIAN, 0.0, GREY, (0, 0, 0, 0))])


def near_tie_scene(rng: np.random.Generator, n_sheets: int = 48, n: int = 4):
    """Stacks of nearly coincident, slightly tilted sheets (separations from 0 to 1e-3, crossing one another): hits of
    different triangles a few ulps apart in t, the case a traversal that skips boxes by the current hit must get right."""
    xs = np.linspace(-2.0, 2.0, n + 1)
    px, py = np.meshgrid(xs, xs, indexing="xy")
    tris = []
    for y in range(n):
        for x in range(n):
            a = y * (n + 1) + x
            tris += [(a, a + 1, a + n + 2), (a, a + n + 2, a + n + 1)]
    tris = np.array(tris, dtype=np.uint32)
    meshes = []
    gaps = [0.0, 1.2e-7, 5e-7, 1e-6, 1e-5, 1e-4, 1e-3]
    z = 0.0
    for k in range(n_sheets):
        z += gaps[k % len(gaps)]
        tilt = rng.uniform(-1, 1, size=2) * gaps[(k * 3 + 1) % len(gaps)]
        pz = z + tilt[0] * px + tilt[1] * py
        pts = np.stack([px.ravel(), py.ravel(), pz.ravel()], axis=1).astype(np.float32)
        mesh = {"points": pts, "triangles": tris, "material_index": 0}
        if k == 0:
            mesh["emission"] = GREY
        meshes.append(mesh)
    return pt.SceneDescription(meshes, [(pt.MAT_LAMBERTIAN, 0.0, GREY, (0, 0, 0, 0))])


def near_tie_rays(rng: np.random.Generator, n: int):
    """Rays onto near_tie_scene from both sides: normal, oblique and grazing (down to 1e-4 rad), some starting between the sheets."""
    target = np.stack([rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n), np.zeros(n)], axis=1)
    elev = 10.0 ** rng.uniform(-4, 0.19, n)                      # angle to the sheets' plane
    az = rng.uniform(0, 2 * np.pi, n)
    side = np.where(rng.uniform(size=n) < 0.5, 1.0, -1.0)
    d = -np.stack([np.cos(elev) * np.cos(az), np.cos(elev) * np.sin(az), side * np.sin(elev)], axis=1)
    dist = np.where(rng.uniform(size=n) < 0.1, 10.0 ** rng.uniform(-6, -3, n), rng.uniform(0.5, 6.0, n))
    o = target - d * dist[:, None]
    return np.concatenate([o, d, np.full((n, 1), np.inf)], axis=1).astype(np.float32)


def random_rays(rng: np.random.Generator, n: int, lo, hi, finite_tmax: bool = False):
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    c = (lo + hi) / 2
    r = np.linalg.norm(hi - lo) / 2 + 1e-3
    o = c + rng.normal(size=(n, 3)) * r * 0.8
    target = c + rng.uniform(-1, 1, size=(n, 3)) * (hi - lo) / 2
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    tmax = np.full((n, 1), np.inf)
    if finite_tmax:
        tmax = rng.uniform(0.1, 2 * r, size=(n, 1))
    return np.concatenate([o, d, tmax], axis=1).astype(np.float32)


def grid_rays(n: int):
    """Rays aimed exactly at grid vertices / edge midpoints of grid_scene(n), axis-parallel and oblique."""
    rays = []
    for y in np.arange(0, n + 0.25, 0.5):
        for x in np.arange(0, n + 0.25, 0.5):
            rays.append([x, y, 3.0, 0.0, 0.0, -1.0, np.inf])
            rays.append([x + 3.0, y, 3.0, -1.0 / 2 ** 0.5, 0.0, -1.0 / 2 ** 0.5, np.inf])
            rays.append([x, y, -2.0, 0.0, 0.0, 1.0, np.inf])
    return np.array(rays, dtype=np.float32)


def furnace_scene(kind: str = "plain"):
    """pt/scenes-test/simple_*_white_sphere*: inward-facing emissive sphere (radiance 0.5, albedo 0.5) around the
    camera; expected image = RGB [1,1,1]. kind: plain | glass (index-matched sphere inside) | skip (material-less sphere)."""
    p, t = icosphere(4)
    # The reference's geometric normal is normalize((p0-p1) x (p2-p1)) (geom.rs:19-37): for counter-clockwise
    # outward-facing triangles it points INWARD, which is the side an enclosing furnace emitter must emit to.
    inward = t
    half = (0.0, 0.0, 0.0, 1.0)  # rgb 0.5 grey: scale 1.0, sigmoid(0) = 0.5 (as TristimulusSpectrum::new gives)
    meshes = [{"points": p * 10.0, "triangles": inward, "material_index": 0, "emission": half}]
    mats = [(pt.MAT_LAMBERTIAN, 0.0, half, (0, 0, 0, 0))]
    if kind == "glass":
        # Dielectric eta = Constant 1: scale 2, c2 = 0 -> 0.5 * 2 = 1.0
        mats.append((pt.MAT_DIELECTRIC, 0.0, (0.0, 0.0, 0.0, 2.0), (0, 0, 0, 0)))
        meshes.append({"points": p * 5.0, "triangles": t, "material_index": 1})
    elif kind == "skip":
        meshes.append({"points": p * 2.5 + np.array([0, 0, 5.0], dtype=np.float32), "triangles": t, "material_index": -1})
    return pt.SceneDescription(meshes, mats)


def deep_chain_scene(n: int, base: float, lit: bool = False):
    """n small triangles on the space diagonal at geometrically growing distances (base^i): the SAH build peels them off a
    few at a time, which makes the BVH as deep as the reference's 64-entry traversal stack allows (180 / 1.6 -> depth 59,
    200 / 1.5 -> depth 64). lit: grey diffuse emitters instead of triangles without a material. Returns (scene, centroids, sizes)."""
    pts, tris, cen, size = [], [], [], []
    for i in range(n):
        s = base ** i
        e = 0.01 * s
        pts += [[s, s, s], [s + e, s, s + e], [s, s + e, s]]
        tris.append([3 * i, 3 * i + 1, 3 * i + 2])
        cen.append([s + e / 3, s + e / 3, s + e / 3])
        size.append(e)
    mesh = {"points": np.array(pts, dtype=np.float32), "triangles": np.array(tris, dtype=np.uint32), "material_index": 0 if lit else -1}
    if lit:
        mesh["emission"] = GREY
    sc = pt.SceneDescription([mesh], [(pt.MAT_LAMBERTIAN, 0.0, GREY, (0, 0, 0, 0))] if lit else [])
    return sc, np.array(cen, dtype=np.float64), np.array(size, dtype=np.float64)


def bvh_depth(nodes) -> int:
    """Depth of the deepest node of a flattened BVH (node words as ptgpu_scene_bvh returns them)."""
    depth, deepest = {0: 0}, 0
    for i in range(len(nodes)):
        d = depth[i]
        deepest = max(deepest, d)
        if (int(nodes[i, 7]) & 0xffff) == 0:
            depth[i + 1] = d + 1
            depth[int(nodes[i, 6])] = d + 1
    return deepest
