#!/usr/bin/env python3
"""Writes the golden vectors under tests/golden/.

The reference (Rust nightly + un-vendored crates) cannot be built or run in this image, and it ships no test vectors,
so these are outputs of the CPU oracle (oracle/, a restatement of the reference's algorithm) on fixed seeded inputs,
plus one published known-answer vector (PCG32). They pin the oracle against silent drift and give the GPU path a
fixture that does not need the oracle at run time.   Usage: python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
sys.path.insert(0, str(HERE.parent))

import oracle_bind as ob  # noqa: E402
import path_tracer_amd as pt  # noqa: E402
import scenes_util as su  # noqa: E402


def soup_case():
    rng = np.random.default_rng(20240611)
    sc = su.random_soup(rng, 300, n_meshes=3)
    o = ob.OracleScene(sc)
    b = o.info()["bounds"]
    rays = np.concatenate([su.random_rays(rng, 1500, b[:3] - 0.5, b[3:] + 0.5), su.random_rays(rng, 500, b[:3] - 0.5, b[3:] + 0.5, finite_tmax=True)])
    return sc, o, rays


def grid_case():
    sc = su.grid_scene(4)
    return sc, ob.OracleScene(sc), su.grid_rays(4)


def hit_words(h):
    """(n, 8) uint32: mesh, triangle, leaf, hit, bits(t, b0, b1, b2)"""
    return np.stack([h["mesh_index"], h["triangle_index"], h["leaf_index"], h["hit"], h["t"].view(np.uint32), h["b0"].view(np.uint32),
                     h["b1"].view(np.uint32), h["b2"].view(np.uint32)], axis=1).astype(np.uint32)


def furnace_films():
    out = {}
    sc = su.furnace_scene("plain")
    o = ob.OracleScene(sc)
    cam = pt.make_camera(su.IDENTITY, 39.5, 12, 8)
    for name, kind, mode, spp in (("path_mis", pt.INTEGRATOR_PATH, pt.MODE_MIS, 5), ("path_pt", pt.INTEGRATOR_PATH, pt.MODE_PT, 3),
                                  ("lightpath", pt.INTEGRATOR_LIGHTPATH, pt.MODE_MIS, 6)):
        film, scale, st = o.render(cam, pt.make_integrator(kind, 6, mode), spp, n_threads=2)
        out[name + "_film"] = film
        out[name + "_meta"] = np.array([scale, st["samples"], st["extend_rays"], st["shadow_rays"], st["segments"]], dtype=np.float64)
    return out


def main():
    sc, o, rays = soup_case()
    nodes, leaf = o.bvh()
    np.savez_compressed(HERE / "soup300.npz", rays=rays, hits=hit_words(o.trace(rays)), nodes=nodes, leaf=leaf)
    sc, o, rays = grid_case()
    np.savez_compressed(HERE / "grid4.npz", rays=rays, hits=hit_words(o.trace(rays)))
    np.savez_compressed(HERE / "furnace_films.npz", **furnace_films())
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
