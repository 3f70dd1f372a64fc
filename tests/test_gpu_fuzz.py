"""Seeded fuzzing of the bit-exact links of the path and of its record pass (needs an MI355X): random triangle soups of random sizes and
scales through both BVH builders (host, device) and both walks (binary, four-wide), closest hits and occlusion against
the CPU oracle, bit for bit (bvh.rs:86-304, geom.rs:180-262). The default is a handful of seeds (seconds);
PT_FUZZ_SEEDS=<n> runs n of them, PT_FUZZ_FIRST=<k> starts at seed k (the round-3 pass: 300 seeds, profiles/r03k_fuzz.log)."""
import os

import numpy as np
import pytest

import oracle_bind as ob
import path_tracer_amd as pt
import scenes_util as su
from test_gpu_parity import assert_hits_bit_exact
from test_gpu_wide_bvh import plane_rays

pytestmark = pytest.mark.gpu

FIRST = int(os.environ.get("PT_FUZZ_FIRST", "1000"))
SEEDS = int(os.environ.get("PT_FUZZ_SEEDS", "6"))


@pytest.fixture
def fuzz_ctx(gpu_ctx):
    yield gpu_ctx
    gpu_ctx.set_bvh_builder(2)
    try:
        gpu_ctx.set_traversal(False)
    except pt.PtError:
        pass


def soup_of(rng):
    """Sizes from one triangle to a few thousand, extents over six decades, triangle sizes from slivers to scene-sized."""
    n_tris = int(np.exp(rng.uniform(0.0, np.log(6000.0))))
    extent = float(10.0 ** rng.uniform(-2.0, 4.0))
    size = extent * float(10.0 ** rng.uniform(-2.5, 0.0))
    return su.random_soup(rng, n_tris, n_meshes=int(rng.integers(1, min(4, n_tris) + 1)), extent=extent, size=size)


@pytest.mark.parametrize("seed", range(FIRST, FIRST + SEEDS))
def test_fuzzed_soups_hit_for_hit(fuzz_ctx, seed):
    rng = np.random.default_rng(seed)
    sc = soup_of(rng)
    o = ob.OracleScene(sc)
    on, ol = o.bvh()
    info = o.info()
    lo, hi = info["bounds"][:3], info["bounds"][3:]
    pad = 0.1 * float((hi - lo).max()) + 1e-6
    finite = su.random_rays(rng, 6000, lo - pad, hi + pad)
    finite[:, 6] = rng.uniform(0.02, 2.0, len(finite)).astype(np.float32) * np.float32(np.linalg.norm(hi - lo) / 2 + pad)   # rays that end inside the scene
    rays = np.concatenate([su.random_rays(rng, 6000, lo - pad, hi + pad), finite, plane_rays(on, rng, 3000)])
    want = o.trace(rays)
    hitp = rays[:, :3] + rays[:, 3:6] * np.where(want["hit"] != 0, want["t"], 1.0)[:, None]
    seg = (hitp[:-1].astype(np.float32), hitp[1:].astype(np.float32))
    want_occ = o.occluded(*seg)
    for builder in (0, 1):
        fuzz_ctx.set_bvh_builder(builder)
        fuzz_ctx.set_scene(sc)
        gn, gl = fuzz_ctx.scene_bvh()
        assert gn.shape == on.shape and (gn == on).all() and (gl == ol).all(), (seed, builder)
        for wide in (False, True):
            fuzz_ctx.set_traversal(wide)
            assert_hits_bit_exact(fuzz_ctx.trace(rays), want)
            assert (fuzz_ctx.occluded(*seg) == want_occ).all(), (seed, builder, wide)


@pytest.mark.parametrize("seed", range(FIRST, FIRST + SEEDS))
def test_fuzzed_statistics_refine_like_the_oracle(gpu_ctx, seed):
    """STree::new_iteration on the device (pt_refine.h) from made-up statistics: between the waves of a small render the
    collector buffer is overwritten with heavy-tailed random sums (sparse cells, empty leaves, energy in one cell) and
    weights scattered around the spatial threshold up to 2^6 times above it (leaves that split several levels at once);
    the device's next generation must equal the oracle's node for node and bit for bit (stree.rs:42-101, dtree.rs:185-198,
    291-324), and the render goes on from it."""
    from test_gpu_sdtree import assert_same_tree
    rng = np.random.default_rng(seed)
    thr = float(10.0 ** rng.uniform(0.5, 2.5))
    sc = pt.LoadedScene("cornell-box-guided", resx=32, resy=16, max_depth=4, integrator="guided")
    gpu_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.spatial_threshold = thr
    n_waves = gpu_ctx.guided_begin(integ, 63)
    guards_before = gpu_ctx.guard_counts()
    for w in range(n_waves):
        gpu_ctx.guided_wave(w)
        if w + 1 == n_waves:
            break
        t = gpu_ctx.guided_tree()
        n_nodes, n_leaves = len(t["c_rad"]), len(t["c_weight"])
        spread = rng.uniform(0.3, 4.0)
        c_rad = np.exp(rng.normal(0.0, spread, (n_nodes, 4))) * (rng.random((n_nodes, 4)) < rng.uniform(0.02, 1.0))
        off = np.concatenate([t["leaf"][:, 2], [n_nodes]]).astype(np.int64)
        for k in np.nonzero(rng.random(n_leaves) < 0.15)[0]:
            c_rad[off[k]:off[k + 1]] = 0.0                                  # a leaf nothing was recorded in
        for k in np.nonzero(rng.random(n_leaves) < (0.1 if w < 3 else 0.0))[0]:   # (early on: every such cell deepens its quadtree by ~3 levels, and 24 is the cap)
            c_rad[off[k] + rng.integers(0, off[k + 1] - off[k]), rng.integers(0, 4)] = 1e6   # ... and one with all its energy in one cell
        t_now = thr * np.sqrt(2.0 ** w)
        splitting = rng.random(n_leaves) < (0.5 if n_leaves < 2000 else 0.02)
        c_weight = np.where(splitting, t_now * 2.0 ** rng.uniform(0.0, 6.0, n_leaves), t_now * rng.uniform(0.0, 1.0, n_leaves))
        c_weight[rng.random(n_leaves) < 0.05] = 0.0
        gpu_ctx.guided_stats_set(np.concatenate([c_rad.ravel(), c_weight]).astype(np.float32))
        t = gpu_ctx.guided_tree()                                            # as the device holds it
        O = ob.OracleSdTree.from_arrays(t, thr)
        gpu_ctx.guided_refine()
        O.new_iteration(w)
        got, want = ob.canonical(gpu_ctx.guided_tree(), thr), O.export()
        O.close()
        assert len(want["spatial"]) >= len(t["spatial"])
        assert want["deepest_level"] < 24, want["deepest_level"]             # the quadtree depth cap (a deviation) is not what is compared
        assert_same_tree(got, want)
        print(f"seed {seed} wave {w}: threshold {thr:.1f}, spatial nodes {len(t['spatial'])} -> {len(want['spatial'])}, quadtree nodes {n_nodes} -> {len(want['c_rad'])}, deepest level {want['deepest_level']}")
    film, scale, st = gpu_ctx.guided_end()
    assert np.isfinite(film).all() and st["samples"] == 32 * 16 * 63
    assert gpu_ctx.guard_counts() == guards_before


RECORD_FILTERS = [(s, d) for s in (pt.FILTER_STOCHASTIC, pt.FILTER_NEAREST, pt.FILTER_BOX) for d in (pt.FILTER_BOX, pt.FILTER_NEAREST)]


@pytest.mark.parametrize("seed", range(FIRST, FIRST + SEEDS))
def test_fuzzed_records_land_where_the_oracle_puts_them(gpu_ctx, seed):
    """STree::record / DTree::record (stree.rs:102-201, dtree.rs:200-289) through the production record pass on made-up
    records, for the six filter pairs a record with three draws can drive: the tree of a short render, its collectors
    emptied, then one batch of random records (points on the scene's surfaces, in its volume and, where the filter
    clips or descends blindly, outside its bounds; log-normal radiance) on both sides. Leaf weights are equal exactly
    where a record goes to one leaf, every collector cell to f32 summation order."""
    from test_gpu_sdtree import assert_same_tree, surface_points
    rng = np.random.default_rng(seed)
    filters = RECORD_FILTERS[seed % len(RECORD_FILTERS)]
    thr = float(10.0 ** rng.uniform(1.0, 2.3))
    sc = pt.LoadedScene("cornell-box-guided", resx=48, resy=48, max_depth=6)
    o = ob.OracleScene(sc)
    gpu_ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.spatial_threshold = thr
    integ.spatial_filter, integ.directional_filter = filters
    n_waves = gpu_ctx.guided_begin(integ, 31)
    trained = int(rng.integers(2, n_waves))
    for w in range(trained):
        gpu_ctx.guided_wave(w)
    t0 = gpu_ctx.guided_tree()
    _, count = gpu_ctx.guided_stats()
    gpu_ctx.guided_stats_set(np.zeros(count, dtype=np.float32))
    t0["c_rad"][:] = 0
    t0["c_weight"][:] = 0
    O = ob.OracleSdTree.from_arrays(t0, thr, filters)
    n = int(rng.integers(2000, 60000))
    lo, hi = t0["bounds"][:3], t0["bounds"][3:]
    inside = rng.uniform(lo, hi, (n // 4, 3)).astype(np.float32)
    outside = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), (n // 8, 3)).astype(np.float32)
    # (the box filter's footprint of a point outside the tree has no volume: 0 / 0 on both sides, nothing to compare)
    p = np.concatenate([surface_points(o, rng, n), inside] + ([] if filters[0] == pt.FILTER_BOX else [outside]))
    wi = rng.normal(size=(len(p), 3)).astype(np.float32)
    wi /= np.linalg.norm(wi, axis=1, keepdims=True).astype(np.float32)
    zero2 = np.zeros((len(wi), 2), dtype=np.float32)
    same = (gpu_ctx.guided_probe(wi, zero2)[:, 14:16] == O.probe(wi, zero2)[:, 14:16]).all(axis=1)   # directions both atan2f agree on (test (ii))
    assert same.mean() > 0.7
    p, wi = p[same], wi[same]
    rad = np.exp(rng.normal(0.0, 2.0, len(p))).astype(np.float32)
    rec = np.concatenate([p, wi, rad[:, None], rng.random((len(p), 3)).astype(np.float32)], axis=1).astype(np.float32)
    gpu_ctx.guided_record_batch(rec)
    O.record(rec)
    got, want = ob.canonical(gpu_ctx.guided_tree(), thr), O.export()
    O.close()
    assert_same_tree(got, want, sums=False)
    gw, ww = got["c_weight"].astype(np.float64), want["c_weight"].astype(np.float64)
    if filters[0] == pt.FILTER_BOX:   # shares of a record's footprint: sums of fractions, in another order
        assert abs(gw.sum() / ww.sum() - 1.0) < 1e-5 and np.abs(gw - ww).max() <= 1e-4 * max(1.0, ww.max()), (filters, np.abs(gw - ww).max())
    else:
        assert (gw == ww).all() and ww.sum() == len(rec), filters
    a, b = got["c_rad"].astype(np.float64), want["c_rad"].astype(np.float64)
    assert abs(a.sum() / b.sum() - 1.0) < 1e-5, (filters, a.sum(), b.sum())
    touched = b > 0
    assert not a[~touched].any()
    # per cell: the same terms in another order (a cell that received k terms is off by ~sqrt(k) * 2^-24 of its largest term)
    scale = np.maximum(b, 1e-3 * b.max())
    assert (np.abs(a - b)[touched] / scale[touched]).max() < 2e-5, (filters, (np.abs(a - b)[touched] / scale[touched]).max())
    for w in range(trained, n_waves):
        gpu_ctx.guided_wave(w)
    film, _, st = gpu_ctx.guided_end()
    assert np.isfinite(film).all()
