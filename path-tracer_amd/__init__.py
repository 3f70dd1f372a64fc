"""path_tracer_amd — ctypes bindings over the two native libraries of this repo.

* ``libptgpu.so``  — the MI355X wavefront path tracer behind the C ABI of ``include/ptgpu.h``
  (drop-in for ``Integrator::render`` of the reference, pt/src/integrator/mod.rs:34-46).
* ``libpthost.so`` — the host side that stays above the hot path (``include/pthost.h``):
  scene_v3.xml loader (pt/src/parser/mod.rs:26-255), spectrum upsampling, film writer.

The Python layer is plumbing for tests and ``bench.py`` only: it marshals numpy arrays into the plain C
structs and raises :class:`PtError` on any non-zero return code. There is no CPU fallback: if the HIP library
is missing or no GPU is present, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_DIR = _PKG / "lib"
REPO = _PKG.parent


class PtError(RuntimeError):
    """A negative PTGPU_E_* / PTHOST_E_* return code (`code`)."""
    code = 0


class NonFiniteSampleWarning(RuntimeWarning):
    """PTGPU_W_NONFINITE: a `path` render met samples whose radiance is not finite (the reference asserts there,
    path_tracer.rs:46); the render is complete without them and stats["nonfinite_samples"] counts them."""


W_NONFINITE = 1


# ----------------------------------------------------------------------------- C structs (include/ptgpu.h)
class Spectrum(C.Structure):
    _fields_ = [("c0", C.c_float), ("c1", C.c_float), ("c2", C.c_float), ("scale", C.c_float)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("roughness", C.c_float), ("s0", Spectrum), ("s1", Spectrum)]


class Mesh(C.Structure):
    _fields_ = [("points", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("triangles", C.POINTER(C.c_uint32)),
                ("n_points", C.c_uint32), ("n_triangles", C.c_uint32), ("material_index", C.c_int32), ("emissive", C.c_int32),
                ("emission", Spectrum)]


class SceneDesc(C.Structure):
    _fields_ = [("meshes", C.POINTER(Mesh)), ("materials", C.POINTER(Material)), ("n_meshes", C.c_uint32), ("n_materials", C.c_uint32)]


class CameraDesc(C.Structure):
    _fields_ = [("world_to_camera", C.c_float * 16), ("fov_deg", C.c_float), ("resx", C.c_uint32), ("resy", C.c_uint32)]


class IntegratorDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("max_depth", C.c_uint32), ("mode", C.c_int32), ("spatial_threshold", C.c_float),
                ("directional_threshold", C.c_float), ("spatial_filter", C.c_int32), ("directional_filter", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("extend_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("segments", C.c_uint64),
                ("kernel_launches", C.c_uint64), ("render_ms", C.c_double), ("extend_ms", C.c_double), ("shade_ms", C.c_double),
                ("stree_nodes", C.c_uint64), ("dtree_nodes", C.c_uint64), ("nonfinite_samples", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class SdTreeInfo(C.Structure):
    _fields_ = [("n_spatial_nodes", C.c_uint32), ("n_leaves", C.c_uint32), ("n_sampling_nodes", C.c_uint32), ("n_collector_nodes", C.c_uint32),
                ("iteration", C.c_uint32), ("refined_ahead", C.c_uint32), ("bounds", C.c_float * 6), ("refine_repeats", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("mesh_index", C.c_uint32), ("triangle_index", C.c_uint32), ("leaf_index", C.c_uint32), ("hit", C.c_uint32),
                ("t", C.c_float), ("b0", C.c_float), ("b1", C.c_float), ("b2", C.c_float)]


HIT_DTYPE = np.dtype([("mesh_index", "<u4"), ("triangle_index", "<u4"), ("leaf_index", "<u4"), ("hit", "<u4"),
                      ("t", "<f4"), ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4")])

INTEGRATOR_LIGHTPATH, INTEGRATOR_PATH, INTEGRATOR_GUIDED = 0, 1, 2
MODE_PT, MODE_NEE, MODE_MIS = 0, 1, 2
FILTER_NEAREST, FILTER_BOX, FILTER_STOCHASTIC = 0, 1, 2
MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_CONDUCTOR = 0, 1, 2
NOALLOC_MAX_RAYS = 1 << 18  # PTGPU_NOALLOC_MAX_RAYS

# every symbol include/ptgpu.h declares (tests check the library exports all of them)
PTGPU_SYMBOLS = [
    "ptgpu_last_error", "ptgpu_init", "ptgpu_destroy", "ptgpu_scene_create", "ptgpu_camera_set", "ptgpu_render",
    "ptgpu_render_shard", "ptgpu_guided_begin", "ptgpu_guided_wave", "ptgpu_guided_stats", "ptgpu_guided_end",
    "ptgpu_trace_batch", "ptgpu_occluded_batch", "ptgpu_scene_info", "ptgpu_scene_bvh", "ptgpu_set_profiling",
    "ptgpu_set_batch_paths", "ptgpu_kernel_times", "ptgpu_set_census", "ptgpu_census", "ptgpu_guided_selfcheck",
    "ptgpu_trace_batch_noalloc", "ptgpu_guided_tree_info", "ptgpu_guided_tree_download", "ptgpu_guided_probe",
    "ptgpu_guided_record_batch", "ptgpu_guided_refine", "ptgpu_guided_capture_records", "ptgpu_guided_captured_records", "ptgpu_guard_counts", "ptgpu_film_device", "ptgpu_set_traversal",
    "ptgpu_get_traversal", "ptgpu_set_record_overlap", "ptgpu_set_bvh_builder", "ptgpu_bvh_built_on_device",
    "ptgpu_set_final_training",
]
PTHOST_SYMBOLS = [
    "pthost_last_error", "pthost_scene_load", "pthost_scene_free", "pthost_scene_desc", "pthost_camera_desc",
    "pthost_integrator_desc", "pthost_spp", "pthost_output_name", "pthost_rgb_to_spectrum", "pthost_rgb_to_spectrum_fit", "pthost_rgb2spec_use",
    "pthost_rgb2spec_fetch", "pthost_rgb2spec_find_interval", "pthost_save_image", "pthost_save_pfm", "pthost_save_image_film",
]

_gpu = None
_host = None


def load_host() -> C.CDLL:
    global _host
    if _host is None:
        path = LIB_DIR / "libpthost.so"
        if not path.exists():
            raise PtError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        lib = C.CDLL(str(path))
        lib.pthost_last_error.restype = C.c_char_p
        lib.pthost_scene_load.restype = C.c_void_p
        lib.pthost_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]
        lib.pthost_scene_free.argtypes = [C.c_void_p]
        lib.pthost_scene_desc.restype = C.POINTER(SceneDesc)
        lib.pthost_scene_desc.argtypes = [C.c_void_p]
        lib.pthost_camera_desc.restype = C.POINTER(CameraDesc)
        lib.pthost_camera_desc.argtypes = [C.c_void_p]
        lib.pthost_integrator_desc.restype = C.POINTER(IntegratorDesc)
        lib.pthost_integrator_desc.argtypes = [C.c_void_p]
        lib.pthost_spp.restype = C.c_uint32
        lib.pthost_spp.argtypes = [C.c_void_p]
        lib.pthost_output_name.restype = C.c_char_p
        lib.pthost_output_name.argtypes = [C.c_void_p]
        lib.pthost_rgb_to_spectrum.argtypes = [C.POINTER(C.c_float), C.POINTER(Spectrum)]
        lib.pthost_rgb_to_spectrum_fit.argtypes = [C.POINTER(C.c_float), C.POINTER(Spectrum)]
        lib.pthost_rgb2spec_use.argtypes = [C.c_char_p]
        lib.pthost_rgb2spec_fetch.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pthost_rgb2spec_find_interval.argtypes = [C.POINTER(C.c_float), C.c_uint32, C.c_float]
        lib.pthost_save_image.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_char_p, C.c_char_p]
        lib.pthost_save_image_film.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_double, C.c_char_p, C.c_char_p]
        lib.pthost_save_pfm.argtypes = [C.POINTER(C.c_double), C.c_uint32, C.c_uint32, C.c_double, C.c_char_p]
        _host = lib
    return _host


def load_gpu() -> C.CDLL:
    """Load libptgpu.so. Raises if it was not built — the product path never falls back to a CPU path."""
    global _gpu
    if _gpu is None:
        path = Path(os.environ.get("PTGPU_LIB", LIB_DIR / "libptgpu.so"))  # PTGPU_LIB: A/B builds of the same library
        if not path.exists():
            raise PtError(f"{path} is missing: run `make` (or __graft_entry__.build()) first")
        lib = C.CDLL(str(path))
        lib.ptgpu_last_error.restype = C.c_char_p
        lib.ptgpu_init.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        lib.ptgpu_destroy.argtypes = [C.c_void_p]
        lib.ptgpu_destroy.restype = None
        lib.ptgpu_scene_create.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        lib.ptgpu_camera_set.argtypes = [C.c_void_p, C.POINTER(CameraDesc)]
        lib.ptgpu_render.argtypes = [C.c_void_p, C.POINTER(IntegratorDesc), C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Stats)]
        lib.ptgpu_render_shard.argtypes = [C.c_void_p, C.POINTER(IntegratorDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                           C.POINTER(C.c_double), C.POINTER(Stats)]
        lib.ptgpu_guided_begin.argtypes = [C.c_void_p, C.POINTER(IntegratorDesc), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        lib.ptgpu_guided_wave.argtypes = [C.c_void_p, C.c_uint32]
        lib.ptgpu_guided_stats.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.ptgpu_guided_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(Stats)]
        lib.ptgpu_guided_selfcheck.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.ptgpu_guided_tree_info.argtypes = [C.c_void_p, C.POINTER(SdTreeInfo)]
        lib.ptgpu_guided_tree_download.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        lib.ptgpu_guided_probe.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint64, C.c_void_p]
        lib.ptgpu_guided_record_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32]
        lib.ptgpu_guided_refine.argtypes = [C.c_void_p]
        lib.ptgpu_guided_capture_records.argtypes = [C.c_void_p, C.c_int]
        lib.ptgpu_guided_captured_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
        lib.ptgpu_trace_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
        lib.ptgpu_trace_batch_noalloc.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
        lib.ptgpu_occluded_batch.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.POINTER(C.c_uint8)]
        lib.ptgpu_scene_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_float)]
        lib.ptgpu_scene_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        lib.ptgpu_set_profiling.argtypes = [C.c_void_p, C.c_int]
        lib.ptgpu_set_batch_paths.argtypes = [C.c_void_p, C.c_uint32]
        lib.ptgpu_set_census.argtypes = [C.c_void_p, C.c_int]
        lib.ptgpu_census.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.ptgpu_film_device.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        lib.ptgpu_guard_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        lib.ptgpu_kernel_times.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_uint32]
        _gpu = lib
    return _gpu


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


# ----------------------------------------------------------------------------- scene descriptions
class SceneDescription:
    """Owns the numpy buffers behind a ptgpu_scene_desc built in Python (synthetic test scenes)."""

    def __init__(self, meshes, materials=()):
        """meshes: list of dicts {points (n,3) f32, triangles (m,3) u32, normals (n,3) f32 | None,
        material_index int (-1 = none), emission Spectrum-tuple | None}; materials: list of
        (kind, roughness, (c0,c1,c2,scale), (c0,c1,c2,scale))."""
        self._keep = []
        self.meshes = (Mesh * len(meshes))()
        for i, m in enumerate(meshes):
            pts = np.ascontiguousarray(m["points"], dtype=np.float32).reshape(-1, 3)
            tris = np.ascontiguousarray(m["triangles"], dtype=np.uint32).reshape(-1, 3)
            self._keep += [pts, tris]
            cm = self.meshes[i]
            cm.points = _fptr(pts)
            cm.triangles = tris.ctypes.data_as(C.POINTER(C.c_uint32))
            cm.n_points = pts.shape[0]
            cm.n_triangles = tris.shape[0]
            nrm = m.get("normals")
            if nrm is not None:
                nrm = np.ascontiguousarray(nrm, dtype=np.float32).reshape(-1, 3)
                assert nrm.shape == pts.shape
                self._keep.append(nrm)
                cm.normals = _fptr(nrm)
            cm.material_index = int(m.get("material_index", -1))
            em = m.get("emission")
            cm.emissive = 1 if em is not None else 0
            if em is not None:
                cm.emission = Spectrum(*[float(v) for v in em])
        self.materials = (Material * max(1, len(materials)))()
        for i, (kind, rough, s0, s1) in enumerate(materials):
            self.materials[i] = Material(int(kind), float(rough), Spectrum(*[float(v) for v in s0]), Spectrum(*[float(v) for v in s1]))
        self.desc = SceneDesc(self.meshes, self.materials, len(meshes), len(materials))

    def pointer(self):
        return C.pointer(self.desc)


class LoadedScene:
    """parser::parse result (pt/src/parser/mod.rs:17-26) held by libpthost."""

    def __init__(self, name: str, collection: str | os.PathLike | None = None, **overrides):
        self.lib = load_host()
        collection = str(collection if collection is not None else REPO / "scenes")
        ov = ";".join(f"{k}={v}" for k, v in overrides.items())
        self.handle = self.lib.pthost_scene_load(collection.encode(), name.encode(), ov.encode() if ov else None, None)
        if not self.handle:
            raise PtError(self.lib.pthost_last_error().decode())
        self.name = name

    def pointer(self):
        return self.lib.pthost_scene_desc(self.handle)

    @property
    def camera(self) -> CameraDesc:
        return self.lib.pthost_camera_desc(self.handle).contents

    @property
    def integrator(self) -> IntegratorDesc:
        return self.lib.pthost_integrator_desc(self.handle).contents

    @property
    def spp(self) -> int:
        return self.lib.pthost_spp(self.handle)

    @property
    def output_name(self) -> str:
        return self.lib.pthost_output_name(self.handle).decode()

    def close(self):
        if self.handle:
            self.lib.pthost_scene_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rgb_to_spectrum(rgb, fit: bool = False) -> Spectrum:
    """TristimulusSpectrum::new: table lookup (rgb2spec::fetch_coeffs); fit=True: the direct fit of that colour instead."""
    lib = load_host()
    a = (C.c_float * 3)(*[float(v) for v in rgb])
    out = Spectrum()
    if (lib.pthost_rgb_to_spectrum_fit if fit else lib.pthost_rgb_to_spectrum)(a, C.byref(out)) != 0:
        raise PtError(lib.pthost_last_error().decode())
    return out


def rgb2spec_use(path=None):
    """Use the coefficient table at `path` (the reference's srgb.coeffs layout or this repo's excerpt); None: the default."""
    lib = load_host()
    if lib.pthost_rgb2spec_use(str(path).encode() if path is not None else None) != 0:
        raise PtError(lib.pthost_last_error().decode())


def rgb2spec_fetch(rgb) -> np.ndarray:
    """rgb2spec::fetch_coeffs (rgb2spec.rs:8-54) on the table in use."""
    lib = load_host()
    a = (C.c_float * 3)(*[float(v) for v in rgb])
    out = (C.c_float * 3)()
    if lib.pthost_rgb2spec_fetch(a, out) != 0:
        raise PtError("rgb2spec_fetch: no table, or the colour is outside the shipped slices")
    return np.array(list(out), dtype=np.float32)


def rgb2spec_find_interval(values, x) -> int:
    v = np.ascontiguousarray(values, dtype=np.float32)
    return load_host().pthost_rgb2spec_find_interval(v.ctypes.data_as(C.POINTER(C.c_float)), len(v), C.c_float(x))


def make_camera(world_to_camera, fov_deg, resx, resy) -> CameraDesc:
    cam = CameraDesc()
    w = np.asarray(world_to_camera, dtype=np.float32).reshape(16)
    for i in range(16):
        cam.world_to_camera[i] = float(w[i])
    cam.fov_deg = float(fov_deg)
    cam.resx = int(resx)
    cam.resy = int(resy)
    return cam


def make_integrator(kind=INTEGRATOR_PATH, max_depth=16, mode=MODE_MIS, spatial_threshold=1000.0) -> IntegratorDesc:
    return IntegratorDesc(kind, max_depth, mode, spatial_threshold, 0.1, FILTER_STOCHASTIC, FILTER_BOX)


KERNEL_NAMES = ["raygen", "extend", "shade", "occlude", "guide_record", "other"]


# ----------------------------------------------------------------------------- the GPU context
class Context:
    """ptgpu_ctx: one GPU. Mirrors the call order of main.rs:32-49 (scene -> camera -> render)."""

    def __init__(self, device: int = 0):
        self.lib = load_gpu()
        h = C.c_void_p()
        self._check(self.lib.ptgpu_init(C.byref(h), device))
        self.handle = h
        self.cam = None

    def _check(self, rc):
        if rc == W_NONFINITE:
            import warnings
            warnings.warn(self.lib.ptgpu_last_error().decode(), NonFiniteSampleWarning, stacklevel=3)
        elif rc != 0:
            e = PtError(f"ptgpu error {rc}: {self.lib.ptgpu_last_error().decode()}")
            e.code = rc
            raise e

    def close(self):
        if getattr(self, "handle", None):
            self.lib.ptgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, scene):
        self._check(self.lib.ptgpu_scene_create(self.handle, scene.pointer()))

    def set_camera(self, cam: CameraDesc):
        self.cam = CameraDesc.from_buffer_copy(cam)
        self._check(self.lib.ptgpu_camera_set(self.handle, C.byref(self.cam)))

    def load(self, scene: LoadedScene):
        self.set_scene(scene)
        self.set_camera(scene.camera)

    def set_profiling(self, on: bool):
        self._check(self.lib.ptgpu_set_profiling(self.handle, 1 if on else 0))

    def set_bvh_builder(self, mode: int):
        """Where the BVH of the scenes set after this call is built: 0 host, 1 device, 2 device from 16 384 triangles on (default)."""
        self.lib.ptgpu_set_bvh_builder.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.ptgpu_set_bvh_builder(self.handle, mode))

    def bvh_built_on_device(self) -> bool:
        self.lib.ptgpu_bvh_built_on_device.argtypes = [C.c_void_p]
        return bool(self.lib.ptgpu_bvh_built_on_device(self.handle))

    def set_final_training(self, on: bool):
        """guided: record the last wave's paths into the SD-tree as the reference does (default), or skip that dead work."""
        self.lib.ptgpu_set_final_training.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.ptgpu_set_final_training(self.handle, 1 if on else 0))

    def set_record_overlap(self, on: bool):
        """guided: record pass of one batch beside the walk of the next (default) or every kernel alone on the GPU."""
        self.lib.ptgpu_set_record_overlap.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.ptgpu_set_record_overlap(self.handle, 1 if on else 0))

    def set_batch_paths(self, n: int):
        self._check(self.lib.ptgpu_set_batch_paths(self.handle, n))

    def set_traversal(self, wide: bool):
        """0 = the reference's binary BVH walk, 1 = the four-wide image of the same tree (same hits)."""
        self.lib.ptgpu_set_traversal.argtypes = [C.c_void_p, C.c_int]
        self._check(self.lib.ptgpu_set_traversal(self.handle, 1 if wide else 0))

    def traversal(self) -> int:
        self.lib.ptgpu_get_traversal.argtypes = [C.c_void_p]
        return int(self.lib.ptgpu_get_traversal(self.handle))

    def set_census(self, on: bool):
        self._check(self.lib.ptgpu_set_census(self.handle, 1 if on else 0))

    def census(self):
        out = (C.c_uint64 * 4)()
        self._check(self.lib.ptgpu_census(self.handle, out))
        return {"extend_nodes": out[0], "extend_tris": out[1], "occlude_nodes": out[2], "occlude_tris": out[3]}

    def guard_counts(self):
        """How often a documented safety limit cut something short since the context was created (ptgpu_guard_counts)."""
        out = (C.c_uint64 * 4)()
        self._check(self.lib.ptgpu_guard_counts(self.handle, out))
        return {"pass_through": out[0], "vertex_store": out[1], "quad_level": out[2], "non_finite": out[3]}

    def kernel_times(self):
        """{stage: (total_ms, launches)} of the last render with profiling on."""
        ms = (C.c_double * len(KERNEL_NAMES))()
        cnt = (C.c_uint64 * len(KERNEL_NAMES))()
        self._check(self.lib.ptgpu_kernel_times(self.handle, ms, cnt, len(KERNEL_NAMES)))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_NAMES)}

    def scene_info(self):
        nn, nt, ne = C.c_uint32(), C.c_uint32(), C.c_uint32()
        b = (C.c_float * 6)()
        self._check(self.lib.ptgpu_scene_info(self.handle, C.byref(nn), C.byref(nt), C.byref(ne), b))
        return {"n_nodes": nn.value, "n_tris": nt.value, "n_emissive_tris": ne.value, "bounds": np.array(list(b), dtype=np.float32)}

    def scene_bvh(self):
        info = self.scene_info()
        nodes = np.zeros((info["n_nodes"], 8), dtype=np.uint32)
        leaf = np.zeros((info["n_tris"], 2), dtype=np.uint32)
        self._check(self.lib.ptgpu_scene_bvh(self.handle, nodes.ctypes.data_as(C.POINTER(C.c_uint32)), leaf.ctypes.data_as(C.POINTER(C.c_uint32))))
        return nodes, leaf

    def trace(self, rays) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(self.lib.ptgpu_trace_batch(self.handle, _fptr(rays), rays.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def trace_noalloc(self, rays) -> np.ndarray:
        """ptgpu_trace_batch_noalloc: same hits through the library's static scratch (no device allocation)."""
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 7)
        out = np.zeros(rays.shape[0], dtype=HIT_DTYPE)
        self._check(self.lib.ptgpu_trace_batch_noalloc(self.handle, _fptr(rays), rays.shape[0], out.ctypes.data_as(C.c_void_p)))
        return out

    def occluded(self, p_from, p_to) -> np.ndarray:
        a = np.ascontiguousarray(p_from, dtype=np.float32).reshape(-1, 3)
        b = np.ascontiguousarray(p_to, dtype=np.float32).reshape(-1, 3)
        out = np.zeros(a.shape[0], dtype=np.uint8)
        self._check(self.lib.ptgpu_occluded_batch(self.handle, _fptr(a), _fptr(b), a.shape[0], out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def render(self, integrator: IntegratorDesc, spp: int, shard_index: int = 0, shard_count: int = 1, host_film: bool = True):
        """Integrator::render. Returns (film_sums (H,W,3) f64, film_scale, stats dict); host_film=False leaves the film on the
        device (film_device()) and returns None for it."""
        film = np.zeros((self.cam.resy, self.cam.resx, 3), dtype=np.float64) if host_film else None
        scale = C.c_double()
        st = Stats()
        integ = IntegratorDesc.from_buffer_copy(integrator)
        self._check(self.lib.ptgpu_render_shard(self.handle, C.byref(integ), spp, shard_index, shard_count,
                                                film.ctypes.data_as(C.POINTER(C.c_double)) if host_film else None, C.byref(scale), C.byref(st)))
        return film, scale.value, st.as_dict()

    def film_device(self):
        """(device pointer, double count) of the film sums of the last completed render (ptgpu_film_device)."""
        p = C.c_void_p()
        n = C.c_size_t()
        self._check(self.lib.ptgpu_film_device(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    # guided, wave by wave (multi-GPU drivers all-reduce guided_stats() between waves)
    def guided_begin(self, integrator: IntegratorDesc, spp: int, shard_index: int = 0, shard_count: int = 1) -> int:
        n = C.c_uint32()
        integ = IntegratorDesc.from_buffer_copy(integrator)
        self._check(self.lib.ptgpu_guided_begin(self.handle, C.byref(integ), spp, shard_index, shard_count, C.byref(n)))
        return n.value

    def guided_wave(self, wave: int):
        self._check(self.lib.ptgpu_guided_wave(self.handle, wave))

    def guided_stats(self):
        """(device pointer, float count) of the collector statistics buffer of the wave just traced."""
        p = C.c_void_p()
        n = C.c_size_t()
        self._check(self.lib.ptgpu_guided_stats(self.handle, C.byref(p), C.byref(n)))
        return p.value, n.value

    def guided_stats_host(self):
        """Host copy (float32) of the statistics buffer (diagnostics and tests; the product path keeps it on the device)."""
        ptr, count = self.guided_stats()
        out = np.empty(count, dtype=np.float32)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), C.c_size_t(count * 4), 2)  # hipMemcpyDeviceToHost
        if rc != 0:
            raise PtError(f"hipMemcpy of the statistics buffer failed ({rc})")
        return out

    def guided_stats_set(self, values):
        """Overwrites the statistics buffer from a host array (tests: what an all-reduce would leave there)."""
        ptr, count = self.guided_stats()
        arr = np.ascontiguousarray(values, dtype=np.float32)
        if arr.size != count:
            raise PtError(f"statistics buffer holds {count} floats, got {arr.size}")
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(count * 4), 1)  # hipMemcpyHostToDevice
        if rc == 0:
            # a copy from pageable memory on the null stream may return once the data is staged; the library's streams are
            # non-blocking and would not wait for it: whoever writes the buffer completes the write before the next call
            # (ptgpu.h: ptgpu_guided_stats), as the multi-GPU driver does after its all-reduce
            rc = hip.hipDeviceSynchronize()
        if rc != 0:
            raise PtError(f"hipMemcpy into the statistics buffer failed ({rc})")

    def guided_selfcheck(self, n_probes, seed=1):
        """Fused quadtree walk vs separate sample + pdf descents on the resident tree -> (sampled, sample_diff, pdf_diff, pdf_only_diff, spatial_diff)."""
        out = (C.c_uint64 * 5)()
        self._check(self.lib.ptgpu_guided_selfcheck(self.handle, C.c_uint32(n_probes), C.c_uint64(seed), out))
        return tuple(int(v) for v in out)

    # SD-tree parity hooks
    def guided_tree(self):
        """The resident SD-tree as a dict of numpy arrays (layout: ptgpu_guided_tree_download in include/ptgpu.h)."""
        info = SdTreeInfo()
        self._check(self.lib.ptgpu_guided_tree_info(self.handle, C.byref(info)))
        t = {"iteration": info.iteration, "refine_repeats": info.refine_repeats, "bounds": np.array(list(info.bounds), dtype=np.float32),
             "spatial": np.zeros((info.n_spatial_nodes, 4), dtype=np.uint32), "leaf": np.zeros((info.n_leaves, 5), dtype=np.uint32),
             "s_rad": np.zeros((info.n_sampling_nodes, 4), dtype=np.float32), "s_child": np.zeros((info.n_sampling_nodes, 4), dtype=np.uint16),
             "s_weight": np.zeros(info.n_leaves, dtype=np.float32),
             "c_rad": np.zeros((info.n_collector_nodes, 4), dtype=np.float32), "c_child": np.zeros((info.n_collector_nodes, 4), dtype=np.uint16),
             "c_weight": np.zeros(info.n_leaves, dtype=np.float32)}
        ptrs = [t[k].ctypes.data_as(C.c_void_p) for k in ("spatial", "leaf", "s_rad", "s_child", "s_weight", "c_rad", "c_child", "c_weight")]
        self._check(self.lib.ptgpu_guided_tree_download(self.handle, *ptrs))
        return t

    def guided_probe(self, points, query_sq, seed=1) -> np.ndarray:
        """-> (n, 16) uint32 words per probe (see ptgpu_guided_probe)."""
        pts = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 3)
        q = np.ascontiguousarray(query_sq, dtype=np.float32).reshape(-1, 2)
        assert len(pts) == len(q)
        out = np.zeros((len(pts), 16), dtype=np.uint32)
        self._check(self.lib.ptgpu_guided_probe(self.handle, _fptr(pts), _fptr(q), len(pts), C.c_uint64(seed), out.ctypes.data_as(C.c_void_p)))
        return out

    def guided_record_batch(self, records):
        rec = np.ascontiguousarray(records, dtype=np.float32).reshape(-1, 10)
        self._check(self.lib.ptgpu_guided_record_batch(self.handle, _fptr(rec), len(rec)))

    def guided_refine(self):
        self._check(self.lib.ptgpu_guided_refine(self.handle))

    def guided_capture_records(self, enabled: bool):
        self._check(self.lib.ptgpu_guided_capture_records(self.handle, 1 if enabled else 0))

    def guided_captured_records(self) -> np.ndarray:
        """-> (n, 14) uint32 words per record of the last wave traced with capture on (see include/ptgpu.h)."""
        n = C.c_uint64()
        self._check(self.lib.ptgpu_guided_captured_records(self.handle, None, 0, C.byref(n)))
        out = np.zeros((n.value, 14), dtype=np.uint32)
        if n.value:
            self._check(self.lib.ptgpu_guided_captured_records(self.handle, out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)))
        return out

    def guided_end(self, host_film: bool = True):
        film = np.zeros((self.cam.resy, self.cam.resx, 3), dtype=np.float64) if host_film else None
        scale = C.c_double()
        st = Stats()
        self._check(self.lib.ptgpu_guided_end(self.handle, film.ctypes.data_as(C.POINTER(C.c_double)) if host_film else None, C.byref(scale), C.byref(st)))
        return film, scale.value, st.as_dict()


def save_image(film, scale, as_splats, directory, file_name):
    lib = load_host()
    film = np.ascontiguousarray(film, dtype=np.float64)
    h, w = film.shape[:2]
    if lib.pthost_save_image(film.ctypes.data_as(C.POINTER(C.c_double)), w, h, float(scale), 1 if as_splats else 0,
                             str(directory).encode(), file_name.encode()) != 0:
        raise PtError("save_image failed")


def save_image_film(tile_film, splat_film, scale, directory, file_name):
    """Film::save_image with both buffers (either may be None)."""
    lib = load_host()
    bufs = [None if f is None else np.ascontiguousarray(f, dtype=np.float64) for f in (tile_film, splat_film)]
    h, w = next(b for b in bufs if b is not None).shape[:2]
    ptrs = [None if b is None else b.ctypes.data_as(C.POINTER(C.c_double)) for b in bufs]
    if lib.pthost_save_image_film(ptrs[0], ptrs[1], w, h, float(scale), str(directory).encode(), file_name.encode()) != 0:
        raise PtError("save_image_film failed")


def save_pfm(film, scale, path):
    lib = load_host()
    film = np.ascontiguousarray(film, dtype=np.float64)
    h, w = film.shape[:2]
    if lib.pthost_save_pfm(film.ctypes.data_as(C.POINTER(C.c_double)), w, h, float(scale), str(path).encode()) != 0:
        raise PtError("save_pfm failed")
