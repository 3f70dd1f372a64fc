"""world_size-2 and world_size-8 rehearsals of the multi-GPU driver (path-tracer_amd/multi_gpu.py) on CPU with the gloo backend.

What runs here is the SAME driver function bench.py uses with RCCL; the renderer behind it is a stand-in:
  * path / lightpath: the oracle rendering its sample-index shard — the reduced film must equal the unsharded render;
  * guided, protocol: a stub context that deals (pass, tile) pairs by the library's rule and deposits one unit per pair into
    a statistics buffer — after the per-wave all-reduce every rank must see every pair exactly once;
  * guided, end to end: the ORACLE behind the same four calls — every rank traces its (pass, tile) pairs of a wave into its
    own collector, the collector sums and leaf weights are all-reduced as one flat buffer, every rank refines the same tree
    from the sums — against the oracle's single-process render of the same job ("all-reduced tree sums equal to
    a serial sum within f32 reassociation tolerance", "1-GPU vs N-GPU image").
No GPU and no libptgpu compute is involved (the C ABI has no CPU path by design).
"""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))

WORLDS = (2, 8)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleShardContext:
    """render() of path_tracer_amd.Context, served by the oracle (same shard semantics as ptgpu_render_shard)."""

    def __init__(self, scene, cam):
        import oracle_bind as ob
        self.o = ob.OracleScene(scene)
        self.cam = cam

    def render(self, integ, spp, rank, world):
        return self.o.render(self.cam, integ, spp, shard_index=rank, shard_count=world, n_threads=2)


class StubGuidedContext:
    """guided_{begin,wave,stats,end} with the library's wave schedule and (pass, tile) ownership rule."""

    def __init__(self, n_tiles, wave_passes):
        self.n_tiles, self.wave_passes = n_tiles, wave_passes
        self.calls = []

    def guided_begin(self, integ, spp, rank, world):
        self.rank, self.world = rank, world
        self.seen_after_reduce = []
        return len(self.wave_passes)

    def guided_wave(self, w):
        from path_tracer_amd import multi_gpu as mg
        if w > 0:
            self.seen_after_reduce.append(self.stats.clone())  # what the host refinement of wave w-1 would read
        self.stats = torch.zeros(self.wave_passes[w] * self.n_tiles, dtype=torch.float32)
        for k in range(self.wave_passes[w]):
            for t in range(self.n_tiles):
                if mg.shard_of_pair(t, k, self.world) == self.rank:
                    self.stats[k * self.n_tiles + t] += 1.0
        self.calls.append(("wave", w))

    def guided_stats(self):
        self.calls.append(("stats",))
        return self.stats, self.stats.numel()

    def guided_end(self):
        film = np.full((4, 4, 3), float(self.stats.sum()), dtype=np.float64)  # last wave: this rank's pairs only
        return film, 1.0, {"samples": int(self.stats.sum())}


class OracleGuidedContext:
    """guided_{begin,wave,stats,end} of path_tracer_amd.Context, served by the oracle's Guider: the statistics buffer is the
    product's (collector sums, 4 per quadtree node, then one weight per spatial leaf: ptgpu_guided_stats), the tree of the
    next wave is refined from whatever the driver left in it."""

    def __init__(self, scene, cam, thr):
        import oracle_bind as ob
        self.ob, self.o, self.cam, self.thr = ob, ob.OracleScene(scene), cam, thr

    def guided_begin(self, integ, spp, rank, world):
        import ctypes as C
        n, buf = C.c_uint32(), (C.c_uint32 * 64)()
        self.ob.load().orc_wave_sample_counts(spp, C.byref(n), buf, 64)
        self.waves = [int(buf[i]) for i in range(n.value)]
        self.integ, self.rank, self.world, self.first = integ, rank, world, 0
        self.tree = self.ob.OracleSdTree.fresh(self.o, self.thr)
        self.nodes = []
        return len(self.waves)

    def guided_wave(self, w):
        if w > 0:   # STree::new_iteration from the (all-reduced) statistics the driver left in self.stats
            t = dict(self.flat)
            n = t["c_rad"].size
            t["c_rad"] = self.stats[:n].numpy().reshape(-1, 4).copy()
            t["c_weight"] = self.stats[n:].numpy().copy()
            self.summed = (t["c_rad"], t["c_weight"])
            self.tree.close()
            self.tree = self.ob.OracleSdTree.from_arrays(t, self.thr)
            self.tree.new_iteration(w - 1)
        self.film, _ = self.tree.wave(self.o, self.cam, self.integ, w, self.first, self.waves[w], n_threads=1, shard_index=self.rank, shard_count=self.world)
        self.first += self.waves[w]
        self.flat = self.tree.export()
        self.nodes.append((len(self.flat["spatial"]), len(self.flat["c_rad"])))
        self.stats = torch.from_numpy(np.concatenate([self.flat["c_rad"].ravel(), self.flat["c_weight"]]).astype(np.float32))

    def guided_stats(self):
        return self.stats, self.stats.numel()

    def guided_end(self):
        return self.film, 1.0 / self.waves[-1], {"samples": 0}


def _worker(rank, port, q, WORLD):
    import torch.distributed as dist

    import path_tracer_amd as pt
    import scenes_util as su
    from path_tracer_amd import multi_gpu as mg

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        out = {}
        # ---- path and lightpath: film of the shards == film of the whole
        sc = su.furnace_scene("plain")
        cam = pt.make_camera(su.IDENTITY, 39.5, 12, 8)
        ctx = OracleShardContext(sc, cam)
        for name, kind, spp in (("path", pt.INTEGRATOR_PATH, 7), ("lightpath", pt.INTEGRATOR_LIGHTPATH, 9)):   # at WORLD = 8 rank 7 has no `path` sample
            integ = pt.make_integrator(kind, 6, pt.MODE_MIS)
            buf = torch.zeros((8, 12, 3), dtype=torch.float64)
            film, scale, st = mg.render_sharded(ctx, integ, spp, rank, WORLD, False, dist=dist, film_buf=buf)
            n = torch.tensor([st["samples"]], dtype=torch.int64)
            dist.all_reduce(n)
            if rank == 0:
                whole, wscale, wst = ctx.o.render(cam, integ, spp, n_threads=2)
                out[name] = (film, scale, int(n.item()), whole, wscale, wst["samples"])
        # ---- guided: per-wave all-reduce sees each (pass, tile) pair once
        g = StubGuidedContext(n_tiles=13, wave_passes=[1, 2, 4, 8])
        buf = torch.zeros((4, 4, 3), dtype=torch.float64)
        film, _, _ = mg.render_sharded(g, None, 15, rank, WORLD, True, dist=dist, wrap_stats=lambda p, c: p, film_buf=buf)
        out["guided"] = ([s.numpy() for s in g.seen_after_reduce], g.calls, film)
        # ---- guided end to end: the oracle's Guider behind the driver; 48 x 32 = 6 tiles, so at WORLD = 8 two ranks idle per pass
        gsc = pt.LoadedScene("cornell-box-guided", resx=48, resy=32, max_depth=5, integrator="guided")
        ginteg = pt.IntegratorDesc.from_buffer_copy(gsc.integrator)
        ginteg.spatial_threshold = 150.0
        og = OracleGuidedContext(gsc, gsc.camera, 150.0)
        buf = torch.zeros((32, 48, 3), dtype=torch.float64)
        film, scale, _ = mg.render_sharded(og, ginteg, 31, rank, WORLD, True, dist=dist, wrap_stats=lambda p, c: p, film_buf=buf)
        out["guided_oracle"] = (film, scale, og.nodes, og.summed)
        q.put((rank, out))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.fixture(scope="module", params=WORLDS, ids=lambda w: f"world{w}")
def ranks(request):
    WORLD = request.param
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, WORLD)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=400) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return WORLD, results


@pytest.mark.parametrize("name", ["path", "lightpath"])
def test_sample_shards_reduce_to_the_unsharded_film(ranks, name):
    WORLD, two_ranks = ranks
    film, scale, n, whole, wscale, wn = two_ranks[0][name]
    assert n == wn and scale == wscale
    assert whole.sum() > 0
    # identical samples, only the f64 summation order differs
    np.testing.assert_allclose(film, whole, rtol=1e-12, atol=1e-12)


def test_guided_pairs_are_dealt_once_and_reduced_every_wave_but_the_last(ranks):
    WORLD, two_ranks = ranks
    for rank in range(WORLD):
        seen, calls, film = two_ranks[rank]["guided"]
        assert len(seen) == 3
        for s in seen:
            assert np.array_equal(s, np.ones_like(s))      # after the all-reduce: every pair exactly once
        assert calls == [("wave", 0), ("stats",), ("wave", 1), ("stats",), ("wave", 2), ("stats",), ("wave", 3)]
    # film reduce: rank 0 holds the pairs of the last wave from both ranks (8 passes x 13 tiles)
    assert np.all(two_ranks[0]["guided"][2] == 8 * 13)


def test_guided_shards_with_statistics_exchange_render_the_unsharded_image(ranks):
    """N ranks, each tracing its (pass, tile) pairs with the oracle's own Guider and exchanging the collector through the
    driver's all-reduce, against the oracle's single-process render (mod.rs:70-122). The f32 sums arrive in another order
    (and a record's stochastic-filter jitter is its path's own draw: no cross-rank RNG coupling), so trees agree up to
    near-tie refinement decisions: equal spatial node counts within 2 %, quadtree nodes within 2 %, images to relMSE < 2e-3."""
    import oracle_bind as ob
    import path_tracer_amd as pt
    WORLD, res = ranks
    film, scale, nodes, _ = res[0]["guided_oracle"]
    for r in range(1, WORLD):   # every rank refined the same tree in every wave, bit for bit, from the reduced statistics
        assert res[r]["guided_oracle"][2] == nodes
        for a, b in zip(res[r]["guided_oracle"][3], res[0]["guided_oracle"][3]):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    gsc = pt.LoadedScene("cornell-box-guided", resx=48, resy=32, max_depth=5, integrator="guided")
    integ = pt.IntegratorDesc.from_buffer_copy(gsc.integrator)
    integ.spatial_threshold = 150.0
    whole, wscale, wst = ob.OracleScene(gsc).render(gsc.camera, integ, 31)
    assert scale == wscale and nodes[-1][0] > 3
    assert abs(nodes[-1][0] - wst["stree_nodes"]) <= 0.02 * wst["stree_nodes"] + 2, (nodes[-1], wst)
    a, b = film * scale, whole * wscale
    assert abs(a.mean() / b.mean() - 1.0) < 5e-3 and ob.rel_mse(a, b) < 2e-3, (a.mean(), b.mean(), ob.rel_mse(a, b))


def test_shard_rules_balance():
    from path_tracer_amd import multi_gpu as mg
    for world in (1, 2, 4, 8):
        for spp in (1, 7, 512):
            got = sorted(i for r in range(world) for i in mg.samples_of_shard(spp, r, world))
            assert got == list(range(spp))
        for passes in (1, 2, 4, 8):
            load = [0] * world
            for k in range(passes):
                for t in range(1024):
                    load[mg.shard_of_pair(t, k, world)] += 1
            assert max(load) - min(load) <= passes   # a one-pass wave still feeds every GPU


# ------------------------------------------------------------------ bench.py as its own launcher
def _bench(args, **env_extra):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(REPO / "bench.py"), *args], env=env, capture_output=True, text=True, timeout=240)


def test_bench_starts_its_own_ranks_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must run TWO ranks (torch.distributed.run started by the parent before
    it touches any GPU) and rank 0 alone prints the one JSON line; --dry-run stops after the process group's first
    all-reduce (gloo here: no GPU), which is what reports how many ranks the collective really saw."""
    import json
    r = _bench(["--gpus", "2", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["requested"] == 2 and line["backend"] == "gloo" and line["scaling"] == "strong"


def test_bench_refuses_a_world_size_that_is_not_what_was_asked_for():
    r = _bench(["--gpus", "2", "--dry-run"], WORLD_SIZE="1", RANK="0")
    assert r.returncode == 2 and "refusing" in r.stderr and not r.stdout.strip()
