"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so). TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

import path_tracer_amd as pt

REPO = Path(__file__).resolve().parent.parent
LIB = REPO / "oracle" / "_build" / "liboracle.so"
_lib = None


def load():
    global _lib
    if _lib is None:
        if not LIB.exists():
            raise RuntimeError(f"{LIB} missing: run `make -C oracle`")
        lib = C.CDLL(str(LIB))
        lib.orc_scene_create.restype = C.c_void_p
        lib.orc_scene_create.argtypes = [C.POINTER(pt.SceneDesc)]
        lib.orc_scene_destroy.argtypes = [C.c_void_p]
        lib.orc_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        lib.orc_scene_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.orc_trace_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]
        lib.orc_occluded_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint8)]
        lib.orc_render.argtypes = [C.c_void_p, C.POINTER(pt.CameraDesc), C.POINTER(pt.IntegratorDesc), C.c_uint32, C.c_uint32, C.c_uint32,
                                   C.c_int, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        lib.orc_fresnel.restype = C.c_float
        lib.orc_fresnel.argtypes = [C.c_float, C.c_float]
        lib.orc_complex_fresnel.restype = C.c_float
        lib.orc_complex_fresnel.argtypes = [C.c_float, C.c_float, C.c_float]
        lib.orc_nonfinite_samples.restype = C.c_uint64
        lib.orc_sdtree_fresh.restype = C.c_void_p
        lib.orc_sdtree_fresh.argtypes = [C.c_void_p, C.c_float]
        lib.orc_sdtree_create.restype = C.c_void_p
        lib.orc_sdtree_create.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6 + [C.c_float]
        lib.orc_sdtree_destroy.argtypes = [C.c_void_p]
        lib.orc_sdtree_set_filters.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        lib.orc_sdtree_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.orc_sdtree_export.argtypes = [C.c_void_p] * 10
        lib.orc_sdtree_probe.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint64, C.c_void_p]
        lib.orc_sdtree_record.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
        lib.orc_sdtree_new_iteration.argtypes = [C.c_void_p, C.c_uint32]
        lib.orc_sdtree_log_records.argtypes = [C.c_void_p, C.c_int]
        lib.orc_sdtree_logged_records.restype = C.c_uint64
        lib.orc_sdtree_logged_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        lib.orc_guided_wave.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(pt.CameraDesc), C.POINTER(pt.IntegratorDesc), C.c_uint32, C.c_uint64, C.c_uint32,
                                        C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        _lib = lib
    return _lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class NonFiniteSample(RuntimeWarning):
    """PathIntegrator::render's assert (path_tracer.rs:46), as the oracle reports it."""


class OracleScene:
    def __init__(self, scene):
        self.lib = load()
        self.handle = self.lib.orc_scene_create(scene.pointer())
        if not self.handle:
            raise RuntimeError("orc_scene_create failed")
        self._scene = scene

    def close(self):
        if self.handle:
            self.lib.orc_scene_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        nn, nt, ne = C.c_uint32(), C.c_uint32(), C.c_uint32()
        b = (C.c_float * 6)()
        self.lib.orc_scene_info(self.handle, C.byref(nn), C.byref(nt), C.byref(ne), b)
        return {"n_nodes": nn.value, "n_tris": nt.value, "n_emissive_tris": ne.value, "bounds": np.array(list(b), dtype=np.float32)}

    def bvh(self):
        i = self.info()
        nodes = np.zeros((i["n_nodes"], 8), dtype=np.uint32)
        leaf = np.zeros((i["n_tris"], 2), dtype=np.uint32)
        self.lib.orc_scene_bvh(self.handle, nodes.ctypes.data_as(C.POINTER(C.c_uint32)), leaf.ctypes.data_as(C.POINTER(C.c_uint32)))
        return nodes, leaf

    def trace(self, rays, counters=False):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(rays.shape[0], dtype=pt.HIT_DTYPE)
        cnt = (C.c_uint64 * 2)()
        self.lib.orc_trace_batch(self.handle, fptr(rays), rays.shape[0], out.ctypes.data_as(C.c_void_p), cnt)
        return (out, (cnt[0], cnt[1])) if counters else out

    def occluded(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(a.shape[0], dtype=np.uint8)
        self.lib.orc_occluded_batch(self.handle, fptr(a), fptr(b), a.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def render(self, cam, integrator, spp, shard_index=0, shard_count=1, n_threads=0, max_waves=0):
        """-> (film sums (H,W,3), film_scale, stats dict)"""
        cam = pt.CameraDesc.from_buffer_copy(cam)
        integ = pt.IntegratorDesc.from_buffer_copy(integrator)
        film = np.zeros((cam.resy, cam.resx, 3), dtype=np.float64)
        scale = C.c_double()
        st = (C.c_uint64 * 8)()
        rc = self.lib.orc_render(self.handle, C.byref(cam), C.byref(integ), spp, shard_index, shard_count, n_threads, max_waves,
                                 film.ctypes.data_as(C.POINTER(C.c_double)), C.byref(scale), st)
        if rc == 2:   # complete, but where the reference would have panicked: say so, like the product's PTGPU_W_NONFINITE
            import warnings
            warnings.warn("orc_render: a sample's radiance is not finite (path_tracer.rs:46 asserts); left out of the film", NonFiniteSample, stacklevel=2)
        elif rc != 0:
            raise RuntimeError("orc_render failed")
        keys = ["samples", "extend_rays", "shadow_rays", "segments", "nodes", "tris", "stree_nodes", "dtree_nodes"]
        return film, scale.value, dict(zip(keys, list(st)))


TREE_KEYS = ("spatial", "leaf", "s_rad", "s_child", "s_weight", "c_rad", "c_child", "c_weight")


class OracleSdTree:
    """The oracle's Guider/STree behind a handle; flat arrays in the layout of ptgpu_guided_tree_download, so that the same
    tree can be held on the GPU and here (tests/test_gpu_sdtree.py)."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("orc_sdtree: inconsistent tree description")
        self.lib = load()
        self.handle = handle

    @classmethod
    def fresh(cls, oscene: "OracleScene", spatial_threshold: float, filters=None):
        return cls(load().orc_sdtree_fresh(oscene.handle, spatial_threshold)).with_filters(filters)

    def with_filters(self, filters):
        """filters = (spatial, directional) as in ptgpu_integrator_desc, or None for the reference parser's (stochastic, box)."""
        if filters is not None:
            self.lib.orc_sdtree_set_filters(self.handle, int(filters[0]), int(filters[1]))
        return self

    @classmethod
    def from_arrays(cls, t: dict, spatial_threshold: float, filters=None):
        a = {k: np.ascontiguousarray(t[k]) for k in TREE_KEYS}
        b = np.ascontiguousarray(t["bounds"], dtype=np.float32)
        assert a["spatial"].dtype == np.uint32 and a["leaf"].dtype == np.uint32 and a["s_child"].dtype == np.uint16 and a["c_child"].dtype == np.uint16
        vp = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
        return cls(load().orc_sdtree_create(vp(b), vp(a["spatial"]), len(a["spatial"]), vp(a["leaf"]), len(a["leaf"]), vp(a["s_rad"]), vp(a["s_child"]),
                                            vp(a["s_weight"]), vp(a["c_rad"]), vp(a["c_child"]), vp(a["c_weight"]), spatial_threshold)).with_filters(filters)

    def close(self):
        if self.handle:
            self.lib.orc_sdtree_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export(self) -> dict:
        """Canonical form: quadtree nodes renumbered in a fixed traversal order, unreachable (freed) nodes dropped."""
        sz = (C.c_uint32 * 5)()
        self.lib.orc_sdtree_sizes(self.handle, sz)
        t = {"bounds": np.zeros(6, dtype=np.float32), "spatial": np.zeros((sz[0], 4), dtype=np.uint32), "leaf": np.zeros((sz[1], 5), dtype=np.uint32),
             "s_rad": np.zeros((sz[2], 4), dtype=np.float32), "s_child": np.zeros((sz[2], 4), dtype=np.uint16), "s_weight": np.zeros(sz[1], dtype=np.float32),
             "c_rad": np.zeros((sz[3], 4), dtype=np.float32), "c_child": np.zeros((sz[3], 4), dtype=np.uint16), "c_weight": np.zeros(sz[1], dtype=np.float32),
             "deepest_level": int(sz[4])}
        self.lib.orc_sdtree_export(self.handle, *[t[k].ctypes.data_as(C.c_void_p) for k in ("bounds",) + TREE_KEYS])
        return t

    def probe(self, points, query_sq, seed=1) -> np.ndarray:
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        q = np.ascontiguousarray(query_sq, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((len(pts), 16), dtype=np.uint32)
        self.lib.orc_sdtree_probe(self.handle, fptr(pts), fptr(q), len(pts), C.c_uint64(seed), out.ctypes.data_as(C.c_void_p))
        return out

    def record(self, records):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 10)
        self.lib.orc_sdtree_record(self.handle, fptr(rec), len(rec))

    def new_iteration(self, iteration: int):
        self.lib.orc_sdtree_new_iteration(self.handle, iteration)

    def log_records(self, enabled: bool):
        """Keep Guider::add's arguments (mod.rs:106-115) of the waves traced from now on (cleared by every call)."""
        self.lib.orc_sdtree_log_records(self.handle, 1 if enabled else 0)

    def logged_records(self) -> np.ndarray:
        """-> (n, 14) uint32 words per record, the layout of ptgpu_guided_captured_records."""
        n = self.lib.orc_sdtree_logged_records(self.handle, None, 0)
        out = np.zeros((n, 14), dtype=np.uint32)
        if n:
            self.lib.orc_sdtree_logged_records(self.handle, out.ctypes.data_as(C.c_void_p), n)
        return out

    def wave(self, oscene: "OracleScene", cam, integrator, iteration, first_pass, n_passes, n_threads=0, shard_index=0, shard_count=1):
        """-> (film sums of this wave (H,W,3), stats dict); shard_*: this rank's (pass, tile) pairs of a multi-GPU job"""
        cam = pt.CameraDesc.from_buffer_copy(cam)
        integ = pt.IntegratorDesc.from_buffer_copy(integrator)
        film = np.zeros((cam.resy, cam.resx, 3), dtype=np.float64)
        st = (C.c_uint64 * 4)()
        rc = self.lib.orc_guided_wave(oscene.handle, self.handle, C.byref(cam), C.byref(integ), iteration, C.c_uint64(first_pass), n_passes, n_threads, shard_index, shard_count,
                                      film.ctypes.data_as(C.POINTER(C.c_double)), st)
        if rc != 0:
            raise RuntimeError("orc_guided_wave failed")
        return film, dict(zip(["samples", "extend_rays", "shadow_rays", "segments"], list(st)))


def canonical(t: dict, spatial_threshold: float = 1000.0) -> dict:
    """Canonical form of a flat tree (e.g. a GPU download): through the oracle's importer and exporter."""
    h = OracleSdTree.from_arrays(t, spatial_threshold)
    try:
        return h.export()
    finally:
        h.close()


def render_sharded(oscene: OracleScene, cam, integrator, spp, workers=8):
    """path / lightpath on an image of a single 16 x 16 tile would run on one thread (the reference parallelises over
    tiles): render the sample-index shards i % workers == k concurrently instead (one oracle thread each; ctypes drops
    the GIL) and add the films — every (pixel, sample) has its own RNG stream, so the sum is the unsharded render."""
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(workers, spp))
    with ThreadPoolExecutor(workers) as ex:
        parts = list(ex.map(lambda k: oscene.render(cam, integrator, spp, shard_index=k, shard_count=workers, n_threads=1), range(workers)))
    film = sum(p[0] for p in parts)
    stats = {k: sum(p[2][k] for p in parts) for k in parts[0][2]}
    return film, parts[0][1], stats


def rel_mse(img, ref):
    """mean((I-R)^2 / (R^2 + 1e-2)) on linear RGB."""
    img = np.asarray(img, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.mean((img - ref) ** 2 / (ref ** 2 + 1e-2)))
