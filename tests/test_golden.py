"""Golden vectors (tests/golden/, written by tests/golden/make_golden.py from the CPU oracle on seeded inputs; the
reference itself cannot run here — see the script's header). CPU: the oracle still reproduces them bit for bit.
GPU: the HIP path reproduces them through the C ABI without the oracle in the loop."""
import sys
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE / "golden"))

import path_tracer_amd as pt
import scenes_util as su

import make_golden as mg

GOLD = HERE / "golden"


def _film_cases():
    return [("path_mis", pt.INTEGRATOR_PATH, pt.MODE_MIS, 5), ("path_pt", pt.INTEGRATOR_PATH, pt.MODE_PT, 3), ("lightpath", pt.INTEGRATOR_LIGHTPATH, pt.MODE_MIS, 6)]


# ------------------------------------------------------------------ CPU: oracle vs golden
@pytest.mark.parametrize("case,file", [(mg.soup_case, "soup300.npz"), (mg.grid_case, "grid4.npz")])
def test_oracle_reproduces_golden_hits(case, file):
    g = np.load(GOLD / file)
    sc, o, rays = case()
    assert (rays.view(np.uint32) == g["rays"].view(np.uint32)).all()       # the seeded inputs are the committed ones
    assert (mg.hit_words(o.trace(rays)) == g["hits"]).all()
    if "nodes" in g:
        nodes, leaf = o.bvh()
        assert (nodes == g["nodes"]).all() and (leaf == g["leaf"]).all()


def test_oracle_reproduces_golden_films():
    g = np.load(GOLD / "furnace_films.npz")
    now = mg.furnace_films()
    for k in g.files:
        assert np.array_equal(now[k], g[k]), k


# ------------------------------------------------------------------ GPU: HIP path vs golden
@pytest.mark.gpu
@pytest.mark.parametrize("case,file", [(mg.soup_case, "soup300.npz"), (mg.grid_case, "grid4.npz")])
def test_gpu_reproduces_golden_hits(gpu_ctx, case, file):
    g = np.load(GOLD / file)
    sc, _, _ = case()
    gpu_ctx.set_scene(sc)
    assert (mg.hit_words(gpu_ctx.trace(g["rays"])) == g["hits"]).all()      # bit exact
    if "nodes" in g:
        nodes, leaf = gpu_ctx.scene_bvh()
        assert (nodes == g["nodes"]).all() and (leaf == g["leaf"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,mode,spp", _film_cases())
def test_gpu_reproduces_golden_films(gpu_ctx, name, kind, mode, spp):
    g = np.load(GOLD / "furnace_films.npz")
    gpu_ctx.set_scene(su.furnace_scene("plain"))
    gpu_ctx.set_camera(pt.make_camera(su.IDENTITY, 39.5, 12, 8))
    film, scale, st = gpu_ctx.render(pt.make_integrator(kind, 6, mode), spp)
    meta = g[name + "_meta"]
    assert scale == meta[0]
    assert [st["samples"], st["extend_rays"], st["shadow_rays"], st["segments"]] == [int(v) for v in meta[1:]]   # same paths, to the ray
    ref = g[name + "_film"]
    # identical paths; device libm may differ from the host's in the last ulp of a few transcendental results
    np.testing.assert_allclose(film, ref, rtol=2e-4, atol=1e-6)
