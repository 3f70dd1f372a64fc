"""GPU tests for the four-wide traversal (ptgpu_set_traversal): it walks a four-wide image of the SAME
tree the reference builds (bvh.rs:148-304) and must report the hits of the reference's binary walk (bvh.rs:86-142) bit
for bit — not merely "tie-equivalent" ones: the leaves, their order and the per-octant visit order are the binary
tree's, and skipping the collapsed inner boxes cannot change which leaves are reached (pt_device.h:
bvh_intersect_wide). Checked against the CPU oracle on random soups, exact ties, near ties, a 59-level tree, every
shipped scene, and rays built to sit exactly on box planes with zero direction components (0 * inf in the slab test)."""
import numpy as np
import pytest

import oracle_bind as ob
import path_tracer_amd as pt
import scenes_util as su
from test_gpu_parity import assert_hits_bit_exact

pytestmark = pytest.mark.gpu


@pytest.fixture
def wide_ctx(gpu_ctx):
    yield gpu_ctx
    try:  # the session's other tests run with the default of the next scene they set; leave this scene on it as well
        gpu_ctx.set_traversal(False)
    except pt.PtError:
        pass


def check_both_modes(ctx, o, rays, segments=None):
    c = o.trace(rays)
    for wide in (False, True):
        ctx.set_traversal(wide)
        assert ctx.traversal() == int(wide)
        assert_hits_bit_exact(ctx.trace(rays), c)
        if segments is not None:
            assert (ctx.occluded(*segments) == o.occluded(*segments)).all()
    return c


def plane_rays(nodes, rng, n):
    """Rays whose origins lie EXACTLY on planes of BVH node boxes and whose directions have one or two zero components
    (either sign of zero): the slab test then multiplies 0 by +-inf. The wide walk skips inner boxes on the argument
    that the slab test is monotone in the box, NaNs included; these rays are where that argument could fail."""
    boxes = nodes[:, :6].view(np.float32)
    k = rng.integers(0, len(boxes), n)
    o = boxes[k, :3] + (boxes[k, 3:6] - boxes[k, :3]) * rng.uniform(-0.2, 1.2, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    for i in range(n):
        ax = rng.integers(0, 3)
        o[i, ax] = boxes[k[i], ax + 3 * rng.integers(0, 2)]           # on the lo or hi plane of that axis
        zero = rng.permutation(3)[: rng.integers(1, 3)]
        if rng.random() < 0.7 and ax not in zero:
            zero[0] = ax
        d[i, zero] = rng.choice(np.array([0.0, -0.0], dtype=np.float32), len(zero))
        if not d[i].any():
            d[i, (ax + 1) % 3] = 1.0
    tmax = np.full((n, 1), np.inf, dtype=np.float32)
    return np.concatenate([o, d, tmax], axis=1).astype(np.float32)


@pytest.mark.parametrize("n_tris,seed", [(1, 0), (2, 1), (37, 2), (500, 3), (5000, 4)])
def test_wide_walk_on_random_soups(wide_ctx, n_tris, seed):
    rng = np.random.default_rng(seed)
    sc = su.random_soup(rng, n_tris, n_meshes=min(3, n_tris))
    wide_ctx.set_scene(sc)
    o = ob.OracleScene(sc)
    info = o.info()
    for finite in (False, True):
        rays = su.random_rays(rng, 20000, info["bounds"][:3] - 0.5, info["bounds"][3:] + 0.5, finite_tmax=finite)
        check_both_modes(wide_ctx, o, rays)
    check_both_modes(wide_ctx, o, plane_rays(wide_ctx.scene_bvh()[0], rng, 20000))


def test_wide_walk_on_exact_ties_and_near_ties(wide_ctx):
    sc = su.grid_scene(8)
    wide_ctx.set_scene(sc)
    o = ob.OracleScene(sc)
    c = check_both_modes(wide_ctx, o, su.grid_rays(8))
    assert c["hit"].sum() > 0.5 * len(c)
    rng = np.random.default_rng(11)
    check_both_modes(wide_ctx, o, plane_rays(wide_ctx.scene_bvh()[0], rng, 50000))
    sc = su.near_tie_scene(rng)
    wide_ctx.set_scene(sc)
    o = ob.OracleScene(sc)
    rays = su.near_tie_rays(rng, 300000)
    c = o.trace(rays)
    idx = np.nonzero(c["hit"])[0][:60000]
    p = rays[idx, :3] + rays[idx, 3:6] * c["t"][idx, None]
    check_both_modes(wide_ctx, o, rays, (p[:-1], p[1:]))


def test_wide_walk_in_a_59_level_tree(wide_ctx):
    sc, cen, size = su.deep_chain_scene(180, 1.6)
    wide_ctx.set_scene(sc)
    o = ob.OracleScene(sc)
    rng = np.random.default_rng(3)
    k = rng.integers(0, 80, 50000)
    target = cen[k] + rng.normal(size=(len(k), 3)) * size[k, None] * 0.3
    side = np.cross(rng.normal(size=(len(k), 3)), [1.0, 1.0, 1.0])
    side /= np.linalg.norm(side, axis=1, keepdims=True)
    origin = cen[k] + side * size[k, None] * rng.uniform(5.0, 50.0, (len(k), 1))
    rays = np.concatenate([origin, target - origin, np.full((len(k), 1), np.inf)], axis=1).astype(np.float32)
    c = check_both_modes(wide_ctx, o, rays, (origin[:2000].astype(np.float32), target[:2000].astype(np.float32)))
    assert c["hit"].mean() > 0.2


@pytest.mark.parametrize("name", ["cornell-box", "veach-bidir", "veach-mis"])
def test_wide_walk_on_shipped_scenes(wide_ctx, name):
    sc = pt.LoadedScene(name)
    wide_ctx.set_scene(sc)
    o = ob.OracleScene(sc)
    info = o.info()
    rng = np.random.default_rng(7)
    rays = su.random_rays(rng, 200000, info["bounds"][:3], info["bounds"][3:])
    c = o.trace(rays)
    idx = np.nonzero(c["hit"])[0][:50000]
    p = rays[idx, :3] + rays[idx, 3:6] * c["t"][idx, None]
    check_both_modes(wide_ctx, o, rays, (p[:-1], p[1:]))
    check_both_modes(wide_ctx, o, plane_rays(wide_ctx.scene_bvh()[0], rng, 100000))


@pytest.mark.parametrize("name,kind", [("cornell-box", pt.INTEGRATOR_PATH), ("veach-bidir", pt.INTEGRATOR_PATH), ("cornell-box", pt.INTEGRATOR_LIGHTPATH),
                                       ("veach-bidir", pt.INTEGRATOR_GUIDED)])
def test_renders_do_not_depend_on_the_traversal(wide_ctx, name, kind):
    """Same hits => same paths: ray counts equal to the ray; films equal up to the order of the film's atomic adds
    (guided: first wave only, later waves differ by the f32 order of the collector's atomics as any two runs do)."""
    sc = pt.LoadedScene(name, resx=96, resy=64, max_depth=8)
    wide_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.kind = kind
    spp = 1 if kind == pt.INTEGRATOR_GUIDED else 8
    out = []
    for wide in (False, True):
        wide_ctx.set_traversal(wide)
        f, s, st = wide_ctx.render(integ, spp)
        out.append((f * s, st))
    (fa, sa), (fb, sb) = out
    assert sa["extend_rays"] == sb["extend_rays"] and sa["shadow_rays"] == sb["shadow_rays"] and sa["segments"] == sb["segments"]
    np.testing.assert_allclose(fa, fb, rtol=1e-9, atol=1e-12)


def test_streamed_occlusion_of_the_guided_integrator(wide_ctx):
    """Batches above the fused walker's limit (786 432 paths) with the wide walk run k_occlude_guided_stream: the first
    wave (no tree yet, so no dependence on the order of the collector's atomics) must trace the rays and give the film of the
    binary walk's kernels."""
    sc = pt.LoadedScene("veach-bidir", resx=1024, resy=800, max_depth=6)
    wide_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.kind = pt.INTEGRATOR_GUIDED
    out = []
    for wide in (False, True):
        wide_ctx.set_traversal(wide)
        f, s, st = wide_ctx.render(integ, 1)
        out.append((f * s, st))
    (fa, sa), (fb, sb) = out
    assert sa["shadow_rays"] > 0
    assert sa["extend_rays"] == sb["extend_rays"] and sa["shadow_rays"] == sb["shadow_rays"] and sa["segments"] == sb["segments"]
    np.testing.assert_allclose(fa, fb, rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("kind", [pt.INTEGRATOR_PATH, pt.INTEGRATOR_LIGHTPATH, pt.INTEGRATOR_GUIDED])
def test_renders_in_a_59_level_tree(wide_ctx, kind):
    """The integrators' walkers with more than 64 KB of traversal stack per workgroup (every one of them needs the raised
    limit, the streamed and the fused ones included): a chain of emissive triangles seen from the origin, along the diagonal
    they sit on. path and lightpath at depth 4: later segments leave from triangles at 1e30 and beyond, where the triangle
    test's products overflow f32 and the reference's own hits are NaNs; a NaN closest hit must switch the distance cull
    off as it does in the reference (pt_device.h, "Culling"), or ray counts fall short of the oracle's by half."""
    sc, _, _ = su.deep_chain_scene(180, 1.6, lit=True)
    o = ob.OracleScene(sc)
    wide_ctx.set_scene(sc)
    assert su.bvh_depth(wide_ctx.scene_bvh()[0]) == 59
    cam = pt.make_camera(su.IDENTITY, 120.0, 96, 96)
    wide_ctx.set_camera(cam)
    integ = pt.make_integrator(kind, 1 if kind == pt.INTEGRATOR_GUIDED else 4, pt.MODE_MIS)
    spp = 1 if kind == pt.INTEGRATOR_GUIDED else 2  # guided: the first wave, as in test_guided_first_wave_equals_oracle
    _, _, cst = o.render(cam, integ, spp)
    assert cst["extend_rays"] > 0 and cst["shadow_rays"] > 0
    for wide in (False, True):
        wide_ctx.set_traversal(wide)
        _, _, st = wide_ctx.render(integ, spp)
        assert (st["extend_rays"], st["shadow_rays"]) == (cst["extend_rays"], cst["shadow_rays"]), wide


def test_set_traversal_needs_a_scene():
    ctx = pt.Context(0)
    with pytest.raises(pt.PtError, match="no scene"):
        ctx.set_traversal(True)
    ctx.close()
