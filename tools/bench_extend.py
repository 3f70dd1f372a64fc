"""Extend micro-benchmark: camera rays of C1 (cornell-box 512x512) and C2 (veach-mis 1280x720) at 1 spp
plus their first-bounce cosine rays, replayed through ptgpu_trace_batch (the closest-hit walk of k_extend behind the
C ABI's parity hook). One JSON line per scene:

  grays_per_s_kernel : rays / the walk kernel's own time (HIP events on the library's stream; rays resident in HBM)
  grays_per_s_call   : rays / wall time of the call (host buffers in and out: PCIe-inclusive, never the headline)
  nodes_per_ray, tris_per_ray : census of the instrumented walk (a second, untimed call)
  roofline           : algorithmic bytes per ray = 28 (ray) + 32 (hit record) + 32 nodes + 36 tris, against 8 TB/s.
                       Both scenes' node and triangle arrays fit in L2 (and the Cornell box's in LDS), so `achieved`
                       prices cache traffic at the HBM roof and can exceed it; it is printed for the formula's sake.

The rays are made here (numpy): pixel-centre camera rays as Camera::sample_we aims them (camera.rs:32-114) and, from
each hit, a cosine-hemisphere direction about the geometric normal on the side the ray came from (seeded).
Usage: python tools/bench_extend.py [repeats]
"""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import path_tracer_amd as pt  # noqa: E402

COPIES = 8


def camera_rays(cam):
    w2c = np.array(list(cam.world_to_camera), dtype=np.float64).reshape(4, 4)
    c2w = np.linalg.inv(w2c)
    W, H = cam.resx, cam.resy
    ps = 2.0 * np.tan(np.radians(cam.fov_deg) / 2.0) / H
    ix, iy = np.meshgrid(np.arange(W), np.arange(H))
    dcam = np.stack([(W / 2 - ix - 0.5) * ps, (H / 2 - iy - 0.5) * ps, np.ones_like(ix, dtype=np.float64)], axis=-1).reshape(-1, 3)
    d = dcam @ c2w[:3, :3].T
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.empty((W * H, 7), dtype=np.float32)
    rays[:, :3] = c2w[:3, 3]
    rays[:, 3:6] = d
    rays[:, 6] = np.inf
    return rays


def geometric_normals(sc, hits):
    desc = sc.pointer().contents
    n = np.zeros((hits.shape[0], 3), dtype=np.float64)
    # ptgpu_hit.mesh_index counts meshes as Scene::new orders them: emitters first, each group in input order
    order = [m for m in range(desc.n_meshes) if desc.meshes[m].emissive] + [m for m in range(desc.n_meshes) if not desc.meshes[m].emissive]
    for k, m in enumerate(order):
        sel = np.nonzero((hits["hit"] != 0) & (hits["mesh_index"] == k))[0]
        if sel.size == 0:
            continue
        mesh = desc.meshes[m]
        pts = np.ctypeslib.as_array(mesh.points, shape=(mesh.n_points, 3)).astype(np.float64)
        tris = np.ctypeslib.as_array(mesh.triangles, shape=(mesh.n_triangles, 3))
        t = tris[hits["triangle_index"][sel]]
        p0, p1, p2 = pts[t[:, 0]], pts[t[:, 1]], pts[t[:, 2]]
        g = np.cross(p0 - p1, p2 - p1)
        n[sel] = g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
    return n


def bounce_rays(rng, rays, hits, normals, scale):
    sel = np.nonzero(hits["hit"] != 0)[0]
    o, d = rays[sel, :3].astype(np.float64), rays[sel, 3:6].astype(np.float64)
    p = o + d * hits["t"][sel, None]
    n = normals[sel]
    n = np.where((np.sum(n * d, axis=1) > 0)[:, None], -n, n)  # the side the ray came from
    # cosine hemisphere, polar disk map (sampling.rs:251-255), in any frame around n
    u0, u1 = rng.random(sel.size), rng.random(sel.size)
    r, phi = np.sqrt(u0), 2 * np.pi * u1
    a = np.where((np.abs(n[:, 0]) > 0.9)[:, None], np.array([0.0, 1.0, 0.0]), np.array([1.0, 0.0, 0.0]))
    t = np.cross(n, a)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(n, t)
    wi = t * (r * np.cos(phi))[:, None] + b * (r * np.sin(phi))[:, None] + n * np.sqrt(np.maximum(0.0, 1.0 - u0))[:, None]
    out = np.empty((sel.size, 7), dtype=np.float32)
    out[:, :3] = p + n * (1e-4 * scale)
    out[:, 3:6] = wi
    out[:, 6] = np.inf
    return out


def main():
    repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    ctx = pt.Context(0)
    for label, name, resx, resy in (("C1", "cornell-box", 512, 512), ("C2", "veach-mis", 1280, 720)):
        sc = pt.LoadedScene(name, resx=resx, resy=resy)
        ctx.load(sc)
        info = ctx.scene_info()
        scale = float(np.max(info["bounds"][3:] - info["bounds"][:3]))
        cam = camera_rays(sc.camera)
        first = ctx.trace(cam)
        batch = np.concatenate([cam, bounce_rays(np.random.default_rng(11), cam, first, geometric_normals(sc, first), scale)])
        batch = np.ascontiguousarray(np.tile(batch, (COPIES, 1)))  # one frame's rays fill the chip for ~0.1 ms only: timed on COPIES of them
        n = batch.shape[0]
        ctx.set_profiling(True)
        ctx.trace(batch)  # warm-up (allocations, page-ins)
        best_kernel, best_call, hit_rate = None, None, 0.0
        for _ in range(repeats):
            t0 = time.perf_counter()
            h = ctx.trace(batch)
            call = time.perf_counter() - t0
            kernel = ctx.kernel_times()["extend"][0] * 1e-3
            best_kernel = kernel if best_kernel is None else min(best_kernel, kernel)
            best_call = call if best_call is None else min(best_call, call)
            hit_rate = float((h["hit"] != 0).mean())
        ctx.set_profiling(False)
        ctx.set_census(True)
        ctx.trace(batch)
        cen = ctx.census()
        ctx.set_census(False)
        nodes, tris = cen["extend_nodes"] / n, cen["extend_tris"] / n
        alg = 28 + 32 + 32 * nodes + 36 * tris
        achieved = alg * n / best_kernel * 1e-9
        print(json.dumps({"benchmark": "extend-micro", "config": f"{label}: {name} {resx}x{resy}, camera rays at 1 spp + first-bounce cosine rays, x{COPIES}",
                          "rays": n, "hit_rate": round(hit_rate, 4), "bvh_nodes": info["n_nodes"], "triangles": info["n_tris"],
                          "grays_per_s_kernel": round(n / best_kernel * 1e-9, 3), "grays_per_s_call": round(n / best_call * 1e-9, 4),
                          "kernel_ms": round(best_kernel * 1e3, 4), "nodes_per_ray": round(nodes, 2), "tris_per_ray": round(tris, 2),
                          "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                                       "alg_bytes_per_ray": round(alg, 1)}}), flush=True)
        sc.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
