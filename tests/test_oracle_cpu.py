"""CPU suite: pins the oracle (oracle/*.hpp) against the only fixed points the reference offers for this path —
published PCG32 vectors, analytic invariants and the furnace scenes' RGB [1,1,1] (README.md:6) — and checks the host
logic and that the C-ABI libraries export every declared symbol. No GPU needed."""
import cmath
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import oracle_bind as ob
import path_tracer_amd as pt
import scenes_util as su

REPO = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------ ABI surface
def test_ptgpu_exports_every_declared_symbol():
    header = (REPO / "include" / "ptgpu.h").read_text()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(ptgpu_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared == set(pt.PTGPU_SYMBOLS), declared ^ set(pt.PTGPU_SYMBOLS)
    lib = C.CDLL(str(pt.LIB_DIR / "libptgpu.so"))
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_pthost_exports_every_declared_symbol():
    header = (REPO / "include" / "pthost.h").read_text()
    declared = set(re.findall(r"^(?:int|void|uint32_t|const [a-z_]+\*|pthost_scene\*)\s+(pthost_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared == set(pt.PTHOST_SYMBOLS), declared ^ set(pt.PTHOST_SYMBOLS)
    lib = pt.load_host()
    for name in sorted(declared):
        assert hasattr(lib, name), name


# ------------------------------------------------------------------ sampler
def test_pcg32_published_vectors(oracle):
    # pcg32-demo (pcg-random.org, pcg_basic): pcg32_srandom_r(42, 54) -> first six outputs
    out = np.zeros(6, dtype=np.uint32)
    oracle.orc_pcg_raw(C.c_uint64(54), C.c_uint64(42), 6, out.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert [hex(v) for v in out] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]


def test_pcg_advance_equals_stepping(oracle):
    # advance(0xffff * i) must land where 0xffff * i single steps land (sampling.rs:43-59)
    a = np.zeros(4, dtype=np.uint32)
    b = np.zeros(0xffff * 2 + 4, dtype=np.uint32)
    oracle.orc_pcg_stream(C.c_uint64(3), C.c_uint64(5), C.c_uint64(2), 4, a.ctypes.data_as(C.POINTER(C.c_uint32)))
    oracle.orc_pcg_stream(C.c_uint64(3), C.c_uint64(5), C.c_uint64(0), b.size, b.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert (a == b[0xffff * 2:]).all()


def test_sampler_floats_in_unit_interval(oracle):
    f = np.zeros(4096, dtype=np.float32)
    oracle.orc_pcg_floats(C.c_uint64(17), C.c_uint64(4), C.c_uint64(1), f.size, f.ctypes.data_as(C.POINTER(C.c_float)))
    assert f.min() >= 0.0 and f.max() < 1.0
    assert abs(f.mean() - 0.5) < 0.02


def test_wavelength_iterates_are_golden_ratio(oracle):
    u = np.zeros(64, dtype=np.float32)
    oracle.orc_wavelength_iterates(u.size, u.ctypes.data_as(C.POINTER(C.c_float)))
    phi = (1 + 5 ** 0.5) / 2
    expect = np.array([((i + 1) * phi) % 1.0 for i in range(64)])
    assert np.abs(u - expect).max() < 1e-4


def test_sample_visible_pdf_normalised(oracle):
    # sampling lambda ~ p(lambda): E[1/p] over the sampled wavelengths = length of the support (360..830)
    lam = np.zeros(8, dtype=np.float32)
    pdf = np.zeros(8, dtype=np.float32)
    acc, n = 0.0, 0
    for u in np.linspace(0.0005, 0.9995, 400):
        oracle.orc_sample_visible(C.c_float(u), lam.ctypes.data_as(C.POINTER(C.c_float)), pdf.ctypes.data_as(C.POINTER(C.c_float)))
        assert lam.min() >= 360.0 and lam.max() <= 830.0
        acc += float((1.0 / pdf.astype(np.float64)).sum())
        n += 8
    assert abs(acc / n - 470.0) / 470.0 < 0.01


# ------------------------------------------------------------------ Fresnel / complex arithmetic (num-complex: parity unpinned, pinned vs cmath)
def test_complex_sqrt_matches_cmath(oracle):
    out = np.zeros(2, dtype=np.float32)
    for re_, im_ in [(1.0, 0.0), (-4.0, 0.0), (0.0, 2.0), (0.0, -2.0), (3.0, 4.0), (-3.0, 4.0), (-3.0, -4.0), (0.2, -3.9), (1e-3, 7.0)]:
        oracle.orc_csqrt(C.c_float(re_), C.c_float(im_), out.ctypes.data_as(C.POINTER(C.c_float)))
        ref = cmath.sqrt(complex(re_, im_))
        assert abs(complex(out[0], out[1]) - ref) < 2e-6 * max(1.0, abs(ref)), (re_, im_, out, ref)


def test_complex_fresnel_known_answers(oracle):
    def ref(cos_i, eta):
        s2 = 1 - cos_i * cos_i
        ct = cmath.sqrt(1 - s2 / (eta * eta))
        rp = (eta * cos_i - ct) / (eta * cos_i + ct)
        rs = (cos_i - eta * ct) / (cos_i + eta * ct)
        return (abs(rp) ** 2 + abs(rs) ** 2) / 2

    for cos_i in (0.0, 0.25, 0.5, 1.0):
        for eta, k in ((0.2, 3.9), (0.924, 2.45), (2.87, 3.03)):
            got = oracle.orc_complex_fresnel(cos_i, eta, k)
            assert abs(got - ref(cos_i, complex(eta, k))) < 2e-5, (cos_i, eta, k, got)
    assert abs(oracle.orc_complex_fresnel(0.0, 0.2, 3.9) - 1.0) < 1e-6  # grazing incidence reflects everything


def test_dielectric_fresnel_known_answers(oracle):
    assert abs(oracle.orc_fresnel(1.0, 1.5) - 0.04) < 1e-6      # ((1.5-1)/(1.5+1))^2
    assert abs(oracle.orc_fresnel(-1.0, 1.5) - 0.04) < 1e-6     # from inside
    assert oracle.orc_fresnel(-0.5, 1.5) == 1.0                  # beyond the critical angle: total internal reflection
    assert oracle.orc_fresnel(0.3, 1.0) < 1e-12                  # index matched (f32 rounding of 0.3 leaves ~1e-15)


# ------------------------------------------------------------------ spectra
def test_grey_spectrum_closed_form():
    s = pt.rgb_to_spectrum((0.5, 0.5, 0.5))
    assert (s.c0, s.c1, s.c2, s.scale) == (0.0, 0.0, 0.0, 1.0)
    s = pt.rgb_to_spectrum((1.5, 1.5, 1.5))  # dielectric eta as Constant
    assert (s.c0, s.c1, s.c2, s.scale) == (0.0, 0.0, 0.0, 3.0)
    s = pt.rgb_to_spectrum((0.0, 0.0, 0.0))
    assert s.scale == 0.0


def test_rgb_upsampling_round_trips_through_d65(oracle):
    """Spectrum::new(rgb) -> reflectance spectrum; integrating it against D65 and the CIE observer gives back rgb.
    Exercises the Gauss-Newton fit that stands in for the missing srgb.coeffs table."""
    from oracle_bind import fptr
    lam = np.arange(360, 831, dtype=np.float32)
    d65 = pt.Spectrum(0.0, 0.0, 0.0, 2.0)  # white illuminant: sigmoid(0)*2 = 1 times D65
    for rgb in [(0.63, 0.065, 0.05), (0.14, 0.45, 0.091), (0.725, 0.71, 0.68), (0.32963, 0.257976, 0.150292), (0.1, 0.2, 0.9)]:
        s = pt.rgb_to_spectrum(rgb)
        refl = np.zeros(lam.size, dtype=np.float32)
        illum = np.zeros(lam.size, dtype=np.float32)
        oracle.orc_spectrum_eval(C.byref(s), 0, fptr(lam), lam.size, fptr(refl))
        oracle.orc_spectrum_eval(C.byref(d65), 1, fptr(lam), lam.size, fptr(illum))
        assert refl.min() >= 0.0 and refl.max() <= s.scale + 1e-6
        # Riemann sum with the oracle's own to_xyz path: feed 8 wavelengths at a time with a uniform "pdf"
        import oracle_tables
        xyz = (oracle_tables.cie_xyz() * (refl * illum)[:, None].astype(np.float64)).sum(axis=0)
        m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
        got = m @ xyz
        assert np.abs(got - np.array(rgb)).max() < 6e-3, (rgb, got)


# ------------------------------------------------------------------ geometry
def test_ray_triangle_known_answers(oracle):
    from oracle_bind import fptr
    tri = np.array([0, 0, 0, 1, 0, 0, 0, 1, 0], dtype=np.float32)
    tb = np.zeros(4, dtype=np.float32)

    def hit(o, d, tmax=np.inf):
        o = np.array(o, dtype=np.float32)
        d = np.array(d, dtype=np.float32)
        return oracle.orc_ray_triangle(fptr(o), fptr(d), fptr(tri), C.c_float(tmax), fptr(tb))

    assert hit([0.25, 0.25, 1], [0, 0, -1]) == 1
    assert abs(tb[0] - 1.0) < 1e-6 and np.allclose(tb[1:], [0.5, 0.25, 0.25], atol=1e-6)
    assert hit([0.25, 0.25, -1], [0, 0, 1]) == 1          # back face also hits (det < 0 branch)
    assert hit([0.25, 0.25, 1], [0, 0, 1]) == 0           # behind the origin
    assert hit([2, 2, 1], [0, 0, -1]) == 0                # outside
    assert hit([0.25, 0.25, 1], [0, 0, -1], 0.5) == 0     # beyond t_max
    assert hit([0.25, 0.25, 1], [0, 0, -1], 1.0) == 1     # t == t_max is accepted (inclusive, geom.rs:251-256)
    assert hit([0.5, 0.0, 1], [0, 0, -1]) == 1            # exactly on an edge: f64 fallback keeps it watertight
    assert hit([0.0, 0.0, 1], [0, 0, -1]) == 1            # exactly on a vertex


def test_wave_sample_counts(oracle):
    def waves(spp):
        n = C.c_uint32()
        buf = np.zeros(64, dtype=np.uint32)
        oracle.orc_wave_sample_counts(spp, C.byref(n), buf.ctypes.data_as(C.POINTER(C.c_uint32)), 64)
        return list(buf[:n.value])

    assert waves(512) == [1, 2, 4, 8, 16, 32, 64, 128, 257]
    assert waves(4096)[-1] == 2049 and sum(waves(4096)) == 4096
    assert waves(1) == [1] and waves(2) == [2] and waves(3) == [1, 2] and waves(7) == [1, 2, 4]


def test_dir_square_round_trip(oracle):
    from oracle_bind import fptr
    rng = np.random.default_rng(1)
    sq = np.zeros(2, dtype=np.float32)
    back = np.zeros(3, dtype=np.float32)
    for _ in range(200):
        d = rng.normal(size=3)
        d = (d / np.linalg.norm(d)).astype(np.float32)
        oracle.orc_dir_square_roundtrip(fptr(d), fptr(sq), fptr(back))
        assert 0.0 <= sq[0] <= 1.0 and 0.0 <= sq[1] <= 1.0
        assert np.abs(back - d).max() < 2e-3


def test_dtree_pdf_integrates_to_one_and_follows_the_data(oracle):
    from oracle_bind import fptr
    rng = np.random.default_rng(5)
    pts = np.clip(rng.normal([0.7, 0.3], 0.05, size=(4000, 2)), 0, 0.999).astype(np.float32)
    rad = np.ones(len(pts), dtype=np.float32)
    g = 128
    qx, qy = np.meshgrid((np.arange(g) + 0.5) / g, (np.arange(g) + 0.5) / g, indexing="xy")
    query = np.stack([qx.ravel(), qy.ravel()], axis=1).astype(np.float32)
    pdf = np.zeros(len(query), dtype=np.float32)
    samples = np.zeros((4000, 2), dtype=np.float32)
    n_nodes = C.c_uint32()
    oracle.orc_dtree_exercise(fptr(pts), fptr(rad), len(pts), pt.FILTER_BOX, 4, C.c_float(0.01), fptr(query), len(query), fptr(pdf),
                              len(samples), fptr(samples), C.byref(n_nodes))
    assert n_nodes.value > 5
    # pdf is w.r.t. solid angle: integral over the unit square * 4 pi = 1
    assert abs(pdf.astype(np.float64).mean() * 4 * np.pi - 1.0) < 1e-3
    assert np.abs(samples.mean(axis=0) - [0.7, 0.3]).max() < 0.03   # samples follow the recorded blob
    assert pdf[np.argmin(np.abs(query - [0.7, 0.3]).sum(axis=1))] > 10 * pdf[np.argmin(np.abs(query - [0.2, 0.8]).sum(axis=1))]


# ------------------------------------------------------------------ BVH / traversal on the oracle alone
def test_bvh_structure_invariants():
    rng = np.random.default_rng(0)
    sc = su.random_soup(rng, 500)
    o = ob.OracleScene(sc)
    nodes, leaf = o.bvh()
    info = o.info()
    assert sorted(map(tuple, leaf)) == sorted(set(map(tuple, leaf))) and len(leaf) == 500   # a permutation of all triangles
    n_tris = nodes[:, 7] & 0xffff
    assert n_tris.sum() == 500
    # pre-order layout: interior node i has its left child at i+1 and offset = right child > i+1
    interior = np.nonzero(n_tris == 0)[0]
    assert (nodes[interior, 6] > interior + 1).all() and (nodes[interior, 6] < info["n_nodes"]).all()
    # leaves tile the triangle array in order
    leaves = np.nonzero(n_tris > 0)[0]
    assert (np.cumsum(np.concatenate([[0], n_tris[leaves][:-1]])) == nodes[leaves, 6]).all()


def test_trace_agrees_with_brute_force():
    rng = np.random.default_rng(3)
    sc = su.random_soup(rng, 300)
    o = ob.OracleScene(sc)
    info = o.info()
    rays = su.random_rays(rng, 2000, info["bounds"][:3], info["bounds"][3:])
    hits = o.trace(rays)
    assert 0.2 < hits["hit"].mean() < 1.0
    # brute force over all triangles with the oracle's own triangle test: same nearest t
    from oracle_bind import fptr
    lib = ob.load()
    d = sc.desc
    tb = np.zeros(4, dtype=np.float32)
    for r, h in list(zip(rays, hits))[:300]:
        best = np.inf
        for m in range(d.n_meshes):
            mesh = d.meshes[m]
            P = np.ctypeslib.as_array(mesh.points, shape=(mesh.n_points, 3))
            T = np.ctypeslib.as_array(mesh.triangles, shape=(mesh.n_triangles, 3))
            for t in T:
                tri = np.ascontiguousarray(P[t].reshape(9))
                if lib.orc_ray_triangle(fptr(np.ascontiguousarray(r[:3])), fptr(np.ascontiguousarray(r[3:6])), fptr(tri), C.c_float(best), fptr(tb)):
                    best = float(tb[0])
        if h["hit"]:
            assert best == float(h["t"])
        else:
            assert best == np.inf


# ------------------------------------------------------------------ furnace scenes on the oracle (README.md:6)
FURNACE_SCENES = {  # pt/scenes-test/<name>/scene.json -> (integrator, mode, fov, meshes, triangles)
    "simple_pt_white_sphere": (pt.INTEGRATOR_PATH, pt.MODE_PT, 19.5, 1, 20480),
    "simple_nee_white_sphere": (pt.INTEGRATOR_PATH, pt.MODE_NEE, 19.5, 1, 20480),
    "simple_mis_white_sphere": (pt.INTEGRATOR_PATH, pt.MODE_MIS, 19.5, 1, 20480),
    "simple_pt_white_sphere_glass": (pt.INTEGRATOR_PATH, pt.MODE_PT, 19.5, 2, 40960),
    "simple_nee_white_sphere_glass": (pt.INTEGRATOR_PATH, pt.MODE_NEE, 19.5, 2, 40960),
    "simple_mis_white_sphere_glass": (pt.INTEGRATOR_PATH, pt.MODE_MIS, 19.5, 2, 40960),
    "simple_mis_white_sphere_skip": (pt.INTEGRATOR_PATH, pt.MODE_MIS, 19.5, 2, 40960),
    "lt_white_sphere": (pt.INTEGRATOR_LIGHTPATH, pt.MODE_MIS, 79.5, 1, 20480),
}


@pytest.mark.parametrize("name", sorted(FURNACE_SCENES))
def test_oracle_furnace_is_white(name):
    """The reference's only fixtures with a known answer, all eight, through the real loader (scenes-test/*/scene_v3.xml,
    written by tools/make_furnace_scenes.py from pt/scenes-test/*/scene.json): this is what pins the oracle to the reference."""
    kind, mode, fov, n_meshes, n_tris = FURNACE_SCENES[name]
    glass = name.endswith("_glass")
    sc = pt.LoadedScene(name, collection=REPO / "scenes-test", **({"max_depth": 64} if glass else {}))  # see the GPU twin of this test
    d = sc.pointer().contents
    assert (sc.integrator.kind, sc.camera.fov_deg, sc.camera.resx, sc.camera.resy, sc.spp) == (kind, fov, 16, 16, 64)
    assert kind == pt.INTEGRATOR_LIGHTPATH or sc.integrator.mode == mode
    assert d.n_meshes == n_meshes and sum(d.meshes[i].n_triangles for i in range(d.n_meshes)) == n_tris
    assert d.meshes[0].emissive and d.meshes[0].material_index == 0
    if name.endswith("_skip"):
        assert d.meshes[1].material_index == -1              # no <ref>: pass-through surface
    if glass:
        assert d.materials[1].kind == pt.MAT_DIELECTRIC
    # the enclosing emitter has radius 10 (to_world scale of the file PLY) and faces inward
    p0 = np.ctypeslib.as_array(d.meshes[0].points, shape=(d.meshes[0].n_points, 3))
    assert np.allclose(np.linalg.norm(p0, axis=1), 10.0, rtol=1e-5)
    o = ob.OracleScene(sc)
    # measured deviations at these counts: <= 1.9e-3 (the spectral-to-RGB integration over 128 wavelength samples, equal in every path variant)
    spp = 1024 if kind == pt.INTEGRATOR_LIGHTPATH else 128
    film, scale, st = ob.render_sharded(o, sc.camera, sc.integrator, spp)   # one 16 x 16 tile: parallel over sample shards instead
    assert st["samples"] == 16 * 16 * spp
    mean = (film * scale).mean(axis=(0, 1))
    assert np.isfinite(film).all() and np.abs(mean - 1.0).max() < 5e-3, mean


@pytest.mark.parametrize("name", [n for n in sorted(FURNACE_SCENES) if n.endswith("_glass")])
def test_oracle_glass_furnace_as_written(name):
    """The three glass fixtures at the fixture's own max_depth = 16 (the test above lets them converge at 64). The camera
    sits inside the index-matched sphere, and every crossing of it spends a bounce of the 16 without scattering anything:
    a path that has made k diffuse bounces when the budget runs out leaves 2^-k of the answer uncollected, and with one or
    two crossings between wall bounces k is 5-15 instead of 16. The missing tail, 1 - RGB, is 2e-3 (measured on both
    sides; the GPU twin of this test holds the GPU to the oracle's value), so |RGB - 1| <= 3e-3 and RGB < 1."""
    sc = pt.LoadedScene(name, collection=REPO / "scenes-test")
    assert sc.integrator.max_depth == 16
    o = ob.OracleScene(sc)
    film, scale, st = ob.render_sharded(o, sc.camera, sc.integrator, 128)
    mean = (film * scale).mean(axis=(0, 1))
    # (the 128 wavelength samples per pixel put +-1.9e-3 of spectral-to-RGB integration noise on top of the deficit)
    assert np.abs(mean - 1.0).max() < 5e-3 and mean.mean() < 1.0, mean


def test_oracle_reports_a_nonfinite_sample():
    """PathIntegrator::render asserts that every sample is finite (path_tracer.rs:46). The oracle leaves such a sample out
    of the film, counts it, and flags the `path` render (orc_render returns 2: a warning here); lightpath's and guided's
    loops have no such check in the reference, there the count is the report."""
    hot = (0.0, 0.0, 0.0, 3e38)   # radiance 1.5e38: the second emitter hit of a path overflows f32
    p, t = su.icosphere(2)
    scene = pt.SceneDescription([{"points": p * 10.0, "triangles": t, "material_index": 0, "emission": hot}], [(pt.MAT_LAMBERTIAN, 0.0, su.GREY, (0, 0, 0, 0))])
    cam = pt.make_camera(su.IDENTITY, 39.5, 4, 4)
    o = ob.OracleScene(scene)
    before = ob.load().orc_nonfinite_samples()
    with pytest.warns(ob.NonFiniteSample):
        film, _, _ = o.render(cam, pt.make_integrator(pt.INTEGRATOR_PATH, 4, pt.MODE_PT), 2)
    assert np.isfinite(film).all()
    assert ob.load().orc_nonfinite_samples() - before == 4 * 4 * 2
    film, _, _ = o.render(cam, pt.make_integrator(pt.INTEGRATOR_LIGHTPATH, 4, pt.MODE_MIS), 2)   # no assert in light_path_tracer.rs
    assert np.isfinite(film).all() and ob.load().orc_nonfinite_samples() - before > 4 * 4 * 2


def test_oracle_guided_furnace_first_wave_white_then_reference_energy_loss():
    """Wave 0 of `guided` samples the BSDF only and must be white. From wave 1 on the reference drops the vertex
    (with its emitted / direct light) whenever Guider::sample returns None for a tree-sampled direction below the
    surface (path_guiding/mod.rs:245-266: `break` comes before the push), so its furnace settles near 0.81, not 1.
    The oracle restates that; the GPU parity tests compare against the same number."""
    sc = su.furnace_scene("plain")
    o = ob.OracleScene(sc)
    cam = pt.make_camera(su.IDENTITY, 19.5, 16, 16)
    film, scale, _ = o.render(cam, pt.make_integrator(pt.INTEGRATOR_GUIDED, 16, pt.MODE_MIS), 1)
    assert scale == 1.0 and abs((film * scale).mean() - 1.0) < 0.05
    film, scale, st = o.render(cam, pt.make_integrator(pt.INTEGRATOR_GUIDED, 16, pt.MODE_MIS), 15)
    assert scale == 1.0 / 8.0 and st["stree_nodes"] > 1
    assert 0.75 < (film * scale).mean() < 0.87


# ------------------------------------------------------------------ host loader
def test_loader_reads_the_shipped_scenes():
    sc = pt.LoadedScene("cornell-box")
    d = sc.pointer().contents
    assert (sc.camera.resx, sc.camera.resy, sc.spp) == (512, 512, 512)
    assert sc.integrator.kind == pt.INTEGRATOR_PATH and sc.integrator.max_depth == 16 and sc.integrator.mode == pt.MODE_MIS
    assert d.n_meshes == 7 and d.n_materials == 9
    assert sum(d.meshes[i].n_triangles for i in range(d.n_meshes)) == 24
    assert sum(d.meshes[i].emissive for i in range(d.n_meshes)) == 1
    assert sc.output_name == "test1.png"
    g = pt.LoadedScene("cornell-box-guided")
    assert g.integrator.kind == pt.INTEGRATOR_GUIDED and g.integrator.spatial_threshold == 1000.0
    v = pt.LoadedScene("veach-mis", resx=1280, resy=720, spp=1024)
    dv = v.pointer().contents
    assert (v.camera.resx, v.camera.resy, v.spp) == (1280, 720, 1024)
    assert sum(dv.meshes[i].n_triangles for i in range(dv.n_meshes)) == 61492
    b = pt.LoadedScene("veach-bidir", integrator="guided", max_depth=32)
    db = b.pointer().contents
    assert sum(db.meshes[i].n_triangles for i in range(db.n_meshes)) == 11544
    assert b.integrator.kind == pt.INTEGRATOR_GUIDED and b.integrator.max_depth == 32


def test_loader_errors_are_reported_not_fatal():
    with pytest.raises(pt.PtError):
        pt.LoadedScene("no-such-scene")


def test_loader_reads_ply_and_obj_files(tmp_path):
    """`<shape type="ply">` (ascii and binary little-endian, with and without normals; parser/mod.rs:1173-1256),
    `<shape type="obj">` and `<shape type="serialized">` (second mesh of a two-mesh file) from files next to the scene,
    with the shape's to_world applied."""
    import struct
    scene = tmp_path / "filemesh"
    scene.mkdir()
    verts = [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0)]
    faces = [(0, 1, 2), (0, 2, 3)]
    (scene / "quad_ascii.ply").write_text(
        "ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
        "element face 2\nproperty list uchar int vertex_indices\nend_header\n"
        + "".join(f"{x} {y} {z}\n" for x, y, z in verts) + "".join(f"3 {a} {b} {c}\n" for a, b, c in faces))
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\n"
              "property float nx\nproperty float ny\nproperty float nz\nelement face 2\nproperty list uchar uint vertex_indices\nend_header\n").encode()
    body = b"".join(struct.pack("<6f", x, y, z, 0.0, 0.0, 1.0) for x, y, z in verts) + b"".join(struct.pack("<B3I", 3, a, b, c) for a, b, c in faces)
    (scene / "quad_bin.ply").write_bytes(header + body)
    (scene / "quad.obj").write_text("".join(f"v {x} {y} {z}\n" for x, y, z in verts) + "vn 0 0 1\n" * 4 + "f 1//1 2//2 3//3\nf 1//1 3//3 4//4\n")  # one normal per position (parser/mod.rs:1297)
    # Mitsuba .serialized, version 3 (parser/mod.rs:1305-1412): two meshes, the second one is the quad (with normals)
    import zlib

    def ser_mesh(vs, fs, normals):
        flags = 0x1000 | (0x0001 if normals else 0)
        raw = struct.pack("<IQQ", flags, len(vs), len(fs)) + b"".join(struct.pack("<3f", *v) for v in vs)
        if normals:
            raw += b"".join(struct.pack("<3f", 0.0, 0.0, 1.0) for _ in vs)
        raw += b"".join(struct.pack("<3I", *f) for f in fs)
        return struct.pack("<HH", 0x041C, 3) + zlib.compress(raw)
    m0, m1 = ser_mesh([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0, 1, 2)], False), ser_mesh(verts, faces, True)
    (scene / "two.serialized").write_bytes(m0 + m1 + struct.pack("<III", 0, len(m0), 2))
    ser_shape = ('<shape type="serialized"><string name="filename" value="two.serialized"/><integer name="shape_index" value="1"/>'
                 '<transform name="to_world"><scale value="2"/><translate value="30, 0, 0"/></transform><ref id="M"/></shape>')
    shape = lambda kind, fn, dx: (f'<shape type="{kind}"><string name="filename" value="{fn}"/><transform name="to_world"><scale value="2"/>'
                                  f'<translate value="{dx}, 0, 0"/></transform><ref id="M"/></shape>')
    (scene / "scene_v3.xml").write_text(
        '<scene version="3.0.0" output="filemesh.png"><integrator type="path"><integer name="max_depth" value="4"/></integrator>'
        '<sensor type="perspective"><float name="fov" value="40"/><transform name="to_world"><matrix value="1 0 0 0 0 1 0 0 0 0 1 -5 0 0 0 1"/></transform>'
        '<sampler type="independent"><integer name="sample_count" value="4"/></sampler>'
        '<film type="hdrfilm"><integer name="width" value="32"/><integer name="height" value="32"/></film></sensor>'
        '<bsdf type="diffuse" id="M"><rgb name="reflectance" value="0.5, 0.5, 0.5"/></bsdf>'
        + shape("ply", "quad_ascii.ply", 0) + shape("ply", "quad_bin.ply", 10) + shape("obj", "quad.obj", 20) + ser_shape + "</scene>")
    sc = pt.LoadedScene("filemesh", collection=tmp_path)
    d = sc.pointer().contents
    assert d.n_meshes == 4
    los = []
    for m in range(4):
        mesh = d.meshes[m]
        assert mesh.n_triangles == 2 and mesh.n_points >= 4
        P = np.ctypeslib.as_array(mesh.points, shape=(mesh.n_points, 3))
        assert np.allclose(P.max(axis=0) - P.min(axis=0), [2, 2, 0])   # the unit quad, scaled by 2
        los.append(P[:, 0].min())
    assert sorted(np.round(los).tolist()) == [0.0, 10.0, 20.0, 30.0]       # each shape's own translation
    has_normals = [bool(d.meshes[m].normals) for m in range(4)]
    assert has_normals.count(True) >= 1 and has_normals.count(False) >= 1   # the ascii file carries none


def test_glass_cube_transform_order():
    """scale 0.5 -> rotate y 45 deg -> translate (0,1,0), each pre-multiplied (parser/mod.rs:666-695)."""
    sc = pt.LoadedScene("cornell-box")
    d = sc.pointer().contents
    cube = d.meshes[6]
    P = np.ctypeslib.as_array(cube.points, shape=(cube.n_points, 3))
    assert np.allclose(P[:, 1].min(), 0.5) and np.allclose(P[:, 1].max(), 1.5)
    assert np.allclose(np.abs(P[:, [0, 2]]).max(), 0.5 * 2 ** 0.5, atol=1e-6)


# ------------------------------------------------------------------ SD-tree handle (the checker of tests/test_gpu_sdtree.py)
def test_oracle_sdtree_handle_waves_equal_whole_render_and_roundtrip():
    """Tracing wave by wave on an orc_sdtree handle is PathGuidingIntegrator::render cut at its wave boundaries: same tree
    sizes and image as the one-call render; export -> import -> export is the identity and both copies answer probes alike."""
    import oracle_bind as ob
    sc = pt.LoadedScene("cornell-box-guided", resx=32, resy=32, max_depth=6)
    o = ob.OracleScene(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.spatial_threshold = 100.0
    T = ob.OracleSdTree.fresh(o, 100.0)
    first, film = 0, None
    for w, n in enumerate([1, 2, 4, 8]):
        if w > 0:
            T.new_iteration(w - 1)
        film, _ = T.wave(o, sc.camera, integ, w, first, n, n_threads=1)
        first += n
    whole, scale, st = o.render(sc.camera, integ, 15, n_threads=1)
    e = T.export()
    assert len(e["spatial"]) == st["stree_nodes"] > 3 and len(e["s_rad"]) == st["dtree_nodes"]
    assert np.array_equal(film, whole) and scale == 1.0 / 8.0    # single-threaded: the same additions in the same order
    T2 = ob.OracleSdTree.from_arrays(e, 100.0)
    e2 = T2.export()
    for k in ob.TREE_KEYS:
        assert np.array_equal(e[k], e2[k]), k
    rng = np.random.default_rng(0)
    pts = rng.uniform(e["bounds"][:3], e["bounds"][3:], (2000, 3)).astype(np.float32)
    q = rng.random((2000, 2)).astype(np.float32)
    a, b = T.probe(pts, q, 3), T2.probe(pts, q, 3)
    assert np.array_equal(a, b)
    assert (e["leaf"][a[:, 0], 4] == 0).mean() > 0.5       # trained leaves
    # records with replayed draws: weights count them, sums add up
    rec = np.concatenate([pts[:500], np.tile([0, 0, 1], (500, 1)), np.ones((500, 1)), rng.random((500, 3))], axis=1).astype(np.float32)
    rec[0, 6] = 0.0
    rec[1, 6] = np.inf
    before = T.export()
    T.record(rec)
    after = T.export()
    assert after["c_weight"].sum() - before["c_weight"].sum() == 498
    leaf_cells = after["c_child"] == 0
    assert abs((after["c_rad"][leaf_cells].sum() - before["c_rad"][leaf_cells].sum()) - 498.0) < 0.05
