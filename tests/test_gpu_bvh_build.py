"""GPU tests for Bvh::new on the device (pt_bvh_build.h, ptgpu_set_bvh_builder): the device builder must
produce the tree of bvh.rs:148-304 WORD FOR WORD — node bounds bits, child offsets, leaf sizes, split axes, and the
order of the triangles inside the leaves (which decides equal-t ties) — i.e. the array the host builder lays out and
the CPU oracle's Bvh. Checked on random soups of every size class (single triangle, fewer than a workgroup, several
workgroup trips per node), coincident centroids (SAH cost NaN -> leaf), coplanar duplicates, near-tie sheets, a
59-level chain, grids of equal triangles (ties in the bucket costs) and every shipped scene; inputs the device builder
declines (a negative zero) must come out of the host builder."""
import numpy as np
import pytest

import oracle_bind as ob
import path_tracer_amd as pt
import scenes_util as su

pytestmark = pytest.mark.gpu


@pytest.fixture
def build_ctx(gpu_ctx):
    yield gpu_ctx
    gpu_ctx.set_bvh_builder(2)


def trees(ctx, scene, want_device=True):
    """(nodes, leaf order) from the host builder, the device builder and the oracle."""
    ctx.set_bvh_builder(0)
    ctx.set_scene(scene)
    assert not ctx.bvh_built_on_device()
    host = ctx.scene_bvh()
    ctx.set_bvh_builder(1)
    ctx.set_scene(scene)
    assert ctx.bvh_built_on_device() == want_device
    dev = ctx.scene_bvh()
    return host, dev, ob.OracleScene(scene).bvh()


def assert_same_tree(a, b):
    assert a[0].shape == b[0].shape, (a[0].shape, b[0].shape)
    assert (a[0] == b[0]).all(), int((a[0] != b[0]).any(axis=1).sum())
    assert (a[1] == b[1]).all()


@pytest.mark.parametrize("n_tris,seed", [(1, 0), (2, 1), (3, 2), (37, 3), (255, 4), (256, 5), (257, 6), (1000, 7), (5000, 8), (40000, 9)])
def test_device_build_equals_host_build_on_soups(build_ctx, n_tris, seed):
    rng = np.random.default_rng(seed)
    sc = su.random_soup(rng, n_tris, n_meshes=min(3, n_tris))
    host, dev, oracle = trees(build_ctx, sc)
    assert_same_tree(dev, host)
    assert_same_tree(dev, oracle)
    assert len(dev[1]) == n_tris and sorted(dev[1][:, 0].tolist()) == sorted(host[1][:, 0].tolist())


def test_device_build_on_degenerate_and_tied_inputs(build_ctx):
    rng = np.random.default_rng(11)
    # (a) every centroid the same point: bucket index 0/0 -> the SAH costs are NaN -> one leaf
    k = 300
    c = np.array([1.0, 2.0, 3.0], dtype=np.float32)
    d = rng.normal(0, 0.5, size=(k, 2, 3)).astype(np.float32)
    pts = np.stack([c + d[:, 0], c + d[:, 1], c - d[:, 0] - d[:, 1]], axis=1)  # centroid of the bounds is not c exactly: fine, some split
    same = np.tile(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32), (k, 1, 1))  # (b) k identical triangles
    grid = su.grid_scene(12)                                                                   # (c) equal triangles on a lattice
    for sc in (pt.SceneDescription([{"points": pts.reshape(-1, 3), "triangles": np.arange(3 * k, dtype=np.uint32).reshape(k, 3), "material_index": -1}], []),
               pt.SceneDescription([{"points": same.reshape(-1, 3), "triangles": np.arange(3 * k, dtype=np.uint32).reshape(k, 3), "material_index": -1}], []),
               grid, su.near_tie_scene(rng)):
        host, dev, oracle = trees(build_ctx, sc)
        assert_same_tree(dev, host)
        assert_same_tree(dev, oracle)
    # identical triangles cannot be split: one leaf holds them all, in input order
    build_ctx.set_bvh_builder(1)
    build_ctx.set_scene(pt.SceneDescription([{"points": same.reshape(-1, 3), "triangles": np.arange(3 * k, dtype=np.uint32).reshape(k, 3), "material_index": -1}], []))
    nodes, leaves = build_ctx.scene_bvh()
    assert len(nodes) == 1 and (nodes[0, 7] & 0xffff) == k and (leaves[:, 1] == np.arange(k)).all()


def test_device_build_of_a_59_level_tree(build_ctx):
    sc, _, _ = su.deep_chain_scene(180, 1.6)
    host, dev, oracle = trees(build_ctx, sc)
    assert su.bvh_depth(dev[0]) == 59
    assert_same_tree(dev, host)
    assert_same_tree(dev, oracle)
    too_deep, _, _ = su.deep_chain_scene(200, 1.5)
    build_ctx.set_bvh_builder(1)
    with pytest.raises(pt.PtError, match="64-entry"):   # same limit, same message, whoever built the tree
        build_ctx.set_scene(too_deep)


@pytest.mark.parametrize("name", ["cornell-box", "veach-bidir", "veach-mis"])
def test_device_build_on_shipped_scenes(build_ctx, name):
    sc = pt.LoadedScene(name)
    host, dev, oracle = trees(build_ctx, sc)
    assert_same_tree(dev, host)
    assert_same_tree(dev, oracle)
    # the default picks the device builder by size
    build_ctx.set_bvh_builder(2)
    build_ctx.set_scene(sc)
    assert build_ctx.bvh_built_on_device() == (len(dev[1]) >= 16384)
    # and the scene built on the device traces like the oracle
    build_ctx.set_bvh_builder(1)
    build_ctx.load(sc)
    info = ob.OracleScene(sc).info()
    rays = su.random_rays(np.random.default_rng(5), 50000, info["bounds"][:3], info["bounds"][3:])
    g, c = build_ctx.trace(rays), ob.OracleScene(sc).trace(rays)
    assert (g["leaf_index"] == c["leaf_index"]).all() and (g["t"].view(np.uint32) == c["t"].view(np.uint32)).all()


def test_inputs_with_a_negative_zero_go_to_the_host_builder(build_ctx):
    # a triangle in the plane x = -0.0 (its bounds hold the negative zero whatever fminf does with mixed zeros) among others
    meshes = [{"points": np.array([[-0.0, 0.0, 0.0], [-0.0, 1.0, 0.0], [-0.0, 0.0, 1.0], [1.0, 0.0, 0.0], [0.0, 2.0, 1.0], [3.0, 0.0, 1.0]], dtype=np.float32),
               "triangles": np.array([[0, 1, 2], [0, 1, 3], [1, 2, 3], [3, 4, 5], [2, 4, 5]], dtype=np.uint32), "material_index": -1}]
    sc0 = pt.SceneDescription(meshes, [])
    host, dev, oracle = trees(build_ctx, sc0, want_device=False)
    assert_same_tree(dev, host)
    # (not compared with the oracle: which of -0.0 and +0.0 a minimum over both returns is the libm's choice — Rust's
    # f32::min leaves it open as well — so two host builds may differ in the sign of such a zero; hits do not depend on it)
    assert (dev[1] == oracle[1]).all() and ((dev[0] ^ oracle[0]) & 0x7fffffff == 0).all()
    with pytest.raises(pt.PtError):
        build_ctx.set_bvh_builder(3)
