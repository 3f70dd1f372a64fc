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


# ------------------------------------------------------------------ converged images (BASELINE metric: "relMSE vs CPU reference")
def _rel_terms(img, ref):
    return ((img - ref) ** 2 / (ref ** 2 + 1e-2)).mean(axis=2).ravel()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind,scene", [("path", pt.INTEGRATOR_PATH, "cornell-box"), ("lightpath", pt.INTEGRATOR_LIGHTPATH, "cornell-box"),
                                             ("guided", pt.INTEGRATOR_GUIDED, "cornell-box-guided")])
def test_gpu_relmse_against_converged_oracle_image(gpu_ctx, name, kind, scene):
    """north_star: "images at high spp match the reference renderer's converged images within a stated relMSE per
    integrator". Reference = the CPU oracle at 16 384 spp (converged_cornell_64.npz, tests/golden/make_converged.py);
    the GPU renders 1 024 spp (guided 1 023). The stated bound per integrator is the ORACLE's own relMSE at that sample
    count against the same reference (stored beside it) — Monte Carlo noise of the reference's estimators on this scene
    (path: 0.65, fireflies of BSDF-sampled light hits behind the glass cube; lightpath: 1.6e-4; guided: 2.0):
      * path, lightpath: the GPU walks the oracle's very paths, so its image equals the oracle's 1 024-spp image
        (relMSE < 1e-4 between the two) and its relMSE against the converged image equals the oracle's within 2 %;
      * guided: a free-running render (f32 atomic order decides near-ties in the tree): the pixel-wise relMSE terms have
        the oracle's distribution — median and 90th percentile within 25 %."""
    g = np.load(GOLD / "converged_cornell_64.npz")
    sc = pt.LoadedScene(scene, resx=64, resy=64, max_depth=16)
    gpu_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.kind = kind
    film, scale, _ = gpu_ctx.render(integ, int(g[f"{name}_spp"]))
    img, ref, own = film * scale, g[f"{name}_ref"], g[f"{name}_own"]
    mine, oracle = _rel_terms(img, ref), _rel_terms(own, ref)
    assert abs(oracle.mean() - float(g[f"{name}_oracle_relmse"])) < 1e-9
    print(f"{name}: relMSE vs converged oracle: gpu {mine.mean():.6g}, oracle {oracle.mean():.6g}; median terms {np.median(mine):.4g} / {np.median(oracle):.4g}")
    if kind != pt.INTEGRATOR_GUIDED:
        assert _rel_terms(img, own).mean() < 1e-4
        assert abs(mine.mean() / oracle.mean() - 1.0) < 0.02
    else:
        assert abs(img.mean() / own.mean() - 1.0) < 0.02
        for q in (0.5, 0.9):
            assert abs(np.quantile(mine, q) / np.quantile(oracle, q) - 1.0) < 0.25, (q, np.quantile(mine, q), np.quantile(oracle, q))
