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


def _block_means(img, k=4):
    h, w, c = img.shape
    return img.reshape(h // k, k, w // k, k, c).mean(axis=(1, 3))


def _block_relmse(img, ref):
    a, b = _block_means(img), _block_means(ref)
    return float(((a - b) ** 2 / (b ** 2 + 1e-2)).mean())


def _render(gpu_ctx, scene, kind, spp):
    sc = pt.LoadedScene(scene, resx=64, resy=64, max_depth=16)
    gpu_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.kind = kind
    film, scale, _ = gpu_ctx.render(integ, spp)
    return film * scale


# Stated relMSE per integrator (north_star), GPU against the CPU oracle's converged image
# (tests/golden/converged_cornell_64.npz: the Cornell box at 64 x 64, 16 384 spp, guided 16 383):
#   path, lightpath: at the reference's own sample count the GPU renders the reference's image       relMSE < 1e-4
#   guided (free-running, at 65 535 spp against the 16 383-spp reference, whose noise is the floor)  relMSE < 0.03,
#                                                                                  of 4 x 4 block means      < 0.003
CONVERGED_RELMSE = {"path": 1e-4, "lightpath": 1e-4, "guided": 0.03}
CONVERGED_BLOCK_RELMSE_GUIDED = 0.003


@pytest.mark.gpu
@pytest.mark.parametrize("name,kind", [("path", pt.INTEGRATOR_PATH), ("lightpath", pt.INTEGRATOR_LIGHTPATH)])
def test_gpu_converges_to_the_oracle_image(gpu_ctx, name, kind):
    """Convergence to the reference image, not agreement in noise. The oracle's converged image is the mean of samples
    0..16383 of every pixel; the GPU draws the same samples (same PCG streams, same wavelength iterates), so its image of
    the first N of them must approach the reference as var * (1/N - 1/M) — a subset mean against the full mean — and BE the
    reference at N = M. Measured: relMSE 0.649 / 0.0439 / < 1e-6 (path), 1.65e-4 / 1.1e-5 / < 1e-6 (lightpath) at
    N = 1 024 / 8 192 / 16 384: the ratio of the first two is the law's 15 within 2 %."""
    g = np.load(GOLD / "converged_cornell_64.npz")
    ref, m = g[f"{name}_ref"], int(g[f"{name}_ref_spp"])
    err = {}
    for n in (1024, 8192, m):
        img = _render(gpu_ctx, "cornell-box", kind, n)
        err[n] = float(_rel_terms(img, ref).mean())
        print(f"{name}: {n} spp against the oracle's {m}-spp image: relMSE {err[n]:.4g}, 4x4-block relMSE {_block_relmse(img, ref):.4g}, mean ratio {img.mean() / ref.mean():.6f}")
    assert err[m] < CONVERGED_RELMSE[name], err
    law = (1 / 1024 - 1 / m) / (1 / 8192 - 1 / m)
    assert 0.75 * law < err[1024] / err[8192] < 1.33 * law, (err, law)


@pytest.mark.gpu
def test_gpu_guided_converges_to_the_oracle_image_and_to_the_path_integrator(gpu_ctx):
    """`guided` runs free (f32 atomic order decides near-ties in the tree), and a render of N spp is not a prefix of a
    longer one (the last wave, half of the samples, forms the image from a tree the earlier waves trained), so it is
    compared as an independent estimate: against the oracle's 16 383-spp guided image its error falls with N until it
    reaches that image's own noise — relMSE 2.1 / 0.077 / 0.0158 at 1 023 / 8 191 / 65 535 spp (0.0146 at 131 071: the
    floor), faster than 1/N at first because more waves also mean a better-trained tree — and, as the check that the
    limit is the right one, against the GPU's own `path` image at 131 072 spp, an estimator that test above ties to the
    oracle sample by sample: relMSE 7.7e-3, 6.1e-4 over 4 x 4 block means. (Fireflies — BSDF-sampled light paths through
    the glass cube, single samples worth 10^4 pixel means — are why the plain figures sit an order of magnitude above the
    block figures.)"""
    g = np.load(GOLD / "converged_cornell_64.npz")
    ref = g["guided_ref"]
    err, block = {}, {}
    for n in (1023, 8191, 65535):
        img = _render(gpu_ctx, "cornell-box-guided", pt.INTEGRATOR_GUIDED, n)
        err[n], block[n] = float(_rel_terms(img, ref).mean()), _block_relmse(img, ref)
        print(f"guided: {n} spp against the oracle's 16383-spp image: relMSE {err[n]:.4g}, 4x4-block relMSE {block[n]:.4g}, mean ratio {img.mean() / ref.mean():.5f}")
    assert err[1023] > 8 * err[8191] and err[8191] > 2.5 * err[65535], err     # at least as 1/N (8x, then towards the floor)
    assert err[65535] < CONVERGED_RELMSE["guided"] and block[65535] < CONVERGED_BLOCK_RELMSE_GUIDED, (err, block)
    hi = _render(gpu_ctx, "cornell-box-guided", pt.INTEGRATOR_GUIDED, (1 << 17) - 1)
    path = _render(gpu_ctx, "cornell-box", pt.INTEGRATOR_PATH, 1 << 17)
    e, b = float(_rel_terms(hi, path).mean()), _block_relmse(hi, path)
    print(f"guided at 131071 spp against path at 131072 spp (both GPU): relMSE {e:.4g}, 4x4-block relMSE {b:.4g}, mean ratio {hi.mean() / path.mean():.5f}")
    assert e < 0.015 and b < 1.5e-3 and abs(hi.mean() / path.mean() - 1.0) < 0.01, (e, b)
