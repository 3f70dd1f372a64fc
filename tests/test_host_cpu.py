"""Host side above the hot path (libpthost.so), CPU only: Film::save_image (camera.rs:197-237) against a numpy restatement,
the PFM dump, and the loader's transform composition / winding rules (parser/mod.rs:231-235,666-695,1258-1303) against
matrices and index orders derived by hand."""
import numpy as np
import pytest
from PIL import Image

import path_tracer_amd as pt


# ------------------------------------------------------------------ Film::save_image
def reference_png_bytes(tile, splat, scale):
    """camera.rs:197-237 in numpy: splats with gamma 1/2.2 first, tile pixels ADDED with gamma 1/2.4, u8 wrapping, alpha 255."""
    def to_u8(film, inv_gamma):
        with np.errstate(invalid="ignore", over="ignore"):
            v = np.power(film * scale, inv_gamma)       # negative base -> NaN, like f64::powf
            v = np.clip(v, 0.0, 1.0) * 255.999            # clamp keeps NaN
        return np.where(np.isnan(v), 0, np.nan_to_num(v, nan=0.0)).astype(np.uint8)   # `as u8`: NaN -> 0, truncation
    h, w = (tile if tile is not None else splat).shape[:2]
    zero = np.zeros((h, w, 3))
    rgb = (to_u8(splat if splat is not None else zero, 1 / 2.2).astype(np.uint16) + to_u8(tile if tile is not None else zero, 1 / 2.4)) % 256
    return np.concatenate([rgb.astype(np.uint8), np.full((h, w, 1), 255, dtype=np.uint8)], axis=2)


def tricky_film(rng, h, w):
    f = rng.random((h, w, 3)) * 1.3                      # some values above 1 (clamped)
    f[0, 0] = [0.0, 1.0, 1.0 - 1e-12]
    f[0, 1] = [-0.5, np.nan, np.inf]                     # powf of a negative base, NaN, inf
    f[0, 2] = [1e-9, 0.5, 0.2176376]                     # near-zero; values around a truncation step
    f[1, :, 0] = np.linspace(0, 1, w)                    # a ramp: every 8-bit step boundary of the transfer curve
    return f


def test_save_image_matches_film_save_image_restatement(tmp_path):
    rng = np.random.default_rng(5)
    h, w = 9, 37                                          # not square, not a multiple of anything
    tile, splat = tricky_film(rng, h, w), tricky_film(rng, h, w)[::-1].copy()
    cases = {"tiles.png": (tile, None, 0.7), "splats.png": (None, splat, 1.9), "both.png": (tile * 0.6, splat * 0.6, 1.0)}
    for name, (t, s, scale) in cases.items():
        pt.save_image_film(t, s, scale, tmp_path / "zout", name)
        img = np.array(Image.open(tmp_path / "zout" / name))
        assert img.shape == (h, w, 4) and img.dtype == np.uint8          # RGBA8, row-major, top row first
        want = reference_png_bytes(t, s, scale)
        assert np.array_equal(img, want), (name, np.argwhere(img != want)[:5])
    both = np.array(Image.open(tmp_path / "zout" / "both.png"))
    a, b = reference_png_bytes(tile * 0.6, None, 1.0)[..., :3].astype(int), reference_png_bytes(None, splat * 0.6, 1.0)[..., :3].astype(int)
    assert ((a + b) > 255).any() and np.array_equal(both[..., :3], (a + b) % 256)   # the u8 `+=` really wrapped somewhere
    # the two transfer curves differ: the same film saved as tiles and as splats
    pt.save_image(tile, 0.7, False, tmp_path / "zout", "g24.png")
    pt.save_image(tile, 0.7, True, tmp_path / "zout", "g22.png")
    g24, g22 = np.array(Image.open(tmp_path / "zout" / "g24.png")), np.array(Image.open(tmp_path / "zout" / "g22.png"))
    assert np.array_equal(g24, reference_png_bytes(tile, None, 0.7)) and np.array_equal(g22, reference_png_bytes(None, tile, 0.7))
    assert (g24[..., :3].astype(int) >= g22[..., :3].astype(int)).all() and (g24 != g22).any()   # 1/2.4 lifts mid-tones more than 1/2.2


def test_pfm_round_trip(tmp_path):
    rng = np.random.default_rng(6)
    film = rng.random((5, 7, 3)) * 3.0
    pt.save_pfm(film, 0.25, tmp_path / "f.pfm")
    raw = (tmp_path / "f.pfm").read_bytes()
    magic, dims, endian, body = raw.split(b"\n", 3)
    assert magic == b"PF" and dims == b"7 5" and float(endian) < 0          # colour, little endian
    data = np.frombuffer(body, dtype="<f4").reshape(5, 7, 3)[::-1]           # PFM rows run bottom-up
    assert np.array_equal(data, (film * 0.25).astype(np.float32))


# ------------------------------------------------------------------ loader: transforms and winding
def load_with_shapes(tmp_path, shapes_xml, sensor_transform='<matrix value="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1"/>', files=()):
    scene = tmp_path / "t"
    scene.mkdir(exist_ok=True)
    for name, text in files:
        (scene / name).write_text(text)
    (scene / "scene_v3.xml").write_text(
        '<scene version="3.0.0"><integrator type="path"><integer name="max_depth" value="4"/></integrator>'
        f'<sensor type="perspective"><float name="fov" value="40"/><transform name="to_world">{sensor_transform}</transform>'
        '<sampler type="independent"><integer name="sample_count" value="4"/></sampler>'
        '<film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>'
        '<bsdf type="diffuse" id="M"><rgb name="reflectance" value="0.5, 0.5, 0.5"/></bsdf>' + shapes_xml + "</scene>")
    return pt.LoadedScene("t", collection=tmp_path)


def mesh_arrays(sc, i):
    m = sc.pointer().contents.meshes[i]
    return (np.ctypeslib.as_array(m.points, shape=(m.n_points, 3)).copy(), np.ctypeslib.as_array(m.triangles, shape=(m.n_triangles, 3)).copy())


RECT = np.array([[-1, -1, 0], [1, -1, 0], [1, 1, 0], [-1, 1, 0]], dtype=np.float64)   # parser/mod.rs:545-556
RECT_TRIS = [[2, 1, 0], [0, 3, 2]]


def test_transform_children_compose_in_document_order_each_premultiplied(tmp_path):
    """parse_transform (parser/mod.rs:666-695): every child is multiplied from the LEFT onto what came before, <matrix>
    replaces. Expected points derived by hand from that rule."""
    rect = lambda t: f'<shape type="rectangle"><transform name="to_world">{t}</transform><ref id="M"/></shape>'  # noqa: E731
    s2 = np.sqrt(0.5)
    cases = [
        ('<translate x="1" y="2" z="3"/><scale value="2"/>', 2.0 * (RECT + [1, 2, 3])),                      # translate first, then scale everything
        ('<scale value="2"/><translate x="1" y="2" z="3"/>', 2.0 * RECT + [1, 2, 3]),
        ('<rotate z="1" angle="90"/>', RECT[:, [1, 0, 2]] * [-1, 1, 1]),                                    # (x, y) -> (-y, x): counter-clockwise about +z
        ('<rotate y="1" angle="45"/><translate x="0" y="0" z="1"/>', np.stack([RECT[:, 0] * s2, RECT[:, 1], -RECT[:, 0] * s2 + 1], axis=1)),
        ('<scale x="2" y="3"/>', RECT * [2, 3, 1]),                                                          # missing component defaults to 1
        ('<translate y="5"/>', RECT + [0, 5, 0]),                                                            # missing components default to 0
        ('<scale value="7"/><matrix value="1 0 0 4 0 1 0 5 0 0 1 6 0 0 0 1"/>', RECT + [4, 5, 6]),          # <matrix> replaces what came before
        # lookat origin (0,0,5) -> target (0,0,0), up +y: fwd = -z, left = up x fwd = -x, up' = +y; columns left|up'|fwd|origin
        ('<lookat origin="0, 0, 5" target="0, 0, 0" up="0, 1, 0"/>', RECT * [-1, 1, 1] + [0, 0, 5]),
        ('<scale value="2"/><lookat origin="1, 2, 3" target="1, 2, 4" up="0, 1, 0"/>', 2.0 * RECT + [1, 2, 3]),  # looking down +z: identity rotation
    ]
    sc = load_with_shapes(tmp_path, "".join(rect(t) for t, _ in cases))
    for i, (t, want) in enumerate(cases):
        P, T = mesh_arrays(sc, i)
        assert np.allclose(P, want, atol=1e-6), (t, P, want)
        assert T.tolist() == RECT_TRIS, t   # all proper (det > 0) transforms, the look-at frame (-x, +y, -z) included: winding untouched


def test_negative_determinant_reverses_the_winding_and_keeps_the_normal_side(tmp_path):
    """parser/mod.rs:231-235: a mirroring to_world reverses every triangle, so that the geometric normal
    normalize((p0-p1) x (p2-p1)) (geom.rs:19-37) is the mirror image of the unmirrored mesh's normal, not its opposite."""
    sc = load_with_shapes(tmp_path, '<shape type="rectangle"><ref id="M"/></shape>'
                                    '<shape type="rectangle"><transform name="to_world"><scale x="-1" y="1" z="1"/></transform><ref id="M"/></shape>'
                                    '<shape type="rectangle"><transform name="to_world"><scale value="-2"/></transform><ref id="M"/></shape>')

    def normals(i):
        P, T = mesh_arrays(sc, i)
        return np.array([np.cross(P[a] - P[b], P[c] - P[b]) for a, b, c in T]), T
    n0, t0 = normals(0)
    n1, t1 = normals(1)
    n2, t2 = normals(2)
    assert t0.tolist() == RECT_TRIS and t1.tolist() == [r[::-1] for r in RECT_TRIS] and t2.tolist() == t1.tolist()
    assert (np.sign(n0[:, 2]) == np.sign(n1[:, 2])).all()        # mirror in x: the rectangle's normal stays on its side of the plane
    assert (np.sign(n2[:, 2]) == -np.sign(n0[:, 2])).all()       # point reflection: the normal is reflected with the geometry


def test_camera_lookat_becomes_world_to_camera(tmp_path):
    """The sensor's to_world is inverted into CameraOptions::world_to_camera (parser/mod.rs:85-101)."""
    sc = load_with_shapes(tmp_path, '<shape type="rectangle"><ref id="M"/></shape>', '<lookat origin="0, 0, 5" target="0, 0, 0" up="0, 1, 0"/>')
    w2c = np.array(list(sc.camera.world_to_camera)).reshape(4, 4)
    c2w = np.array([[-1, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 5], [0, 0, 0, 1.0]])
    assert np.allclose(w2c @ c2w, np.eye(4), atol=1e-6)
    assert np.allclose(w2c @ [0, 0, 0, 1], [0, 0, 5, 1])           # the world origin lies 5 units in front of the camera (+z is forward)


def test_obj_faces_are_wound_against_their_vertex_normals(tmp_path):
    """parse_obj (parser/mod.rs:1267-1288): a face whose (p0-p1) x (p2-p1) points against the sum of its vertex normals is
    reversed. For a counter-clockwise face that cross product is -z: with normals +z the face is reversed, with -z kept."""
    obj_up = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvn 0 0 1\nvn 0 0 1\nf 1//1 2//2 3//3\n"
    obj_down = obj_up.replace("vn 0 0 1", "vn 0 0 -1")
    obj_plain = "v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n"
    shape = lambda f, extra="": f'<shape type="obj"><string name="filename" value="{f}"/>{extra}<ref id="M"/></shape>'  # noqa: E731
    sc = load_with_shapes(tmp_path, shape("up.obj") + shape("down.obj") + shape("plain.obj") + shape("up.obj", '<boolean name="face_normals" value="true"/>'),
                          files=[("up.obj", obj_up), ("down.obj", obj_down), ("plain.obj", obj_plain)])
    tris = [mesh_arrays(sc, i)[1].tolist() for i in range(4)]
    assert tris[0] == [[2, 1, 0]] and tris[1] == [[0, 1, 2]] and tris[2] == [[0, 1, 2]]
    assert tris[3] == [[2, 1, 0]]                                    # face_normals only drops the normals afterwards (parser/mod.rs:617-621)
    d = sc.pointer().contents
    assert bool(d.meshes[0].normals) and not bool(d.meshes[2].normals) and not bool(d.meshes[3].normals)


# ------------------------------------------------------------------ rgb2spec (spectrum/rgb2spec.rs)
REPO = __import__("pathlib").Path(__file__).resolve().parent.parent


def _find_interval_rs(values, x):
    """rgb2spec.rs:56-71, line by line."""
    last = len(values) - 2
    left, size = 0, last
    while size > 0:
        half = size >> 1
        middle = left + half
        if values[middle] <= x:
            left = middle
            size -= half + 1
        else:
            size = half
    return min(left, last)


def test_find_interval_is_the_references_bisection():
    """rgb2spec::find_interval (rgb2spec.rs:56-71). QUIRK kept: its bisection continues from `middle`, not `middle + 1`, so it
    is not the textbook "last index with values[i] <= x" (it disagrees with that for about half of all x on the table's own
    scale axis, e.g. 35 instead of 37 at z = 0.7) — the loader restates it as written. For the only z a lookup ever asks
    for, 0.5 (TristimulusSpectrum::new divides by twice the largest component), both give slice 31 of 64."""
    rng = np.random.default_rng(3)
    for n in (2, 3, 4, 5, 64, 65):
        v = np.sort(rng.random(n)).astype(np.float32)
        xs = np.concatenate([v, np.nextafter(v, -1), np.nextafter(v, 2), [-1.0, 0.0, 2.0], rng.random(50)]).astype(np.float32)
        for x in xs:
            got = pt.rgb2spec_find_interval(v, x)
            assert got == _find_interval_rs(v, x) and 0 <= got <= n - 2, (n, float(x))
    t = np.arange(64) / 63.0
    ss = lambda x: x * x * (3 - 2 * x)  # noqa: E731
    scale = ss(ss(t)).astype(np.float32)
    assert pt.rgb2spec_find_interval(scale, 0.5) == 31 == int(np.searchsorted(scale, np.float32(0.5), side="right")) - 1
    assert pt.rgb2spec_find_interval(scale, 0.7) == 35 and int(np.searchsorted(scale, np.float32(0.7), side="right")) - 1 == 37


def _sigmoid(s, lam):
    x = (np.float64(s.c0) * lam + np.float64(s.c1)) * lam + np.float64(s.c2)
    return 0.5 + x / (2.0 * np.sqrt(1.0 + x * x))


def test_scene_colours_come_from_the_coefficient_table():
    """the reference's colours are fetch_coeffs lookups (rgb2spec.rs:8-54) in a res-64 table. The shipped
    excerpt of the regenerated table gives the ten scene colours the committed coefficients bit for bit, the loader hands
    exactly those to the GPU and the oracle, and they differ from a direct fit of the same colour (what round 2 used) by
    the table's interpolation error only: below 1e-3 in reflectance (measured: at most 7.6e-4, the Cornell box's red)."""
    import json
    import struct
    pt.rgb2spec_use(None)
    fx = json.loads((REPO / "tests" / "golden" / "rgb2spec_scene_colours.json").read_text())
    assert len(fx) == 10
    lam = np.arange(360.0, 831.0)
    bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]  # noqa: E731
    worst = 0.0
    for e in fx:
        s = pt.rgb_to_spectrum(e["rgb"])
        assert [bits(s.c0), bits(s.c1), bits(s.c2), bits(s.scale)] == e["c0_c1_c2_scale_bits"], e["rgb"]
        m = 2.0 * max(e["rgb"])
        c = pt.rgb2spec_fetch([v / np.float32(m) for v in np.float32(e["rgb"])])
        assert [bits(v) for v in c] == e["c0_c1_c2_scale_bits"][:3]
        f = pt.rgb_to_spectrum(e["rgb"], fit=True)
        assert f.scale == s.scale
        worst = max(worst, float(np.abs(_sigmoid(s, lam) - _sigmoid(f, lam)).max()))
    assert 1e-5 < worst < 1e-3, worst
    # the loader uses them: the Cornell box's red wall
    sc = pt.LoadedScene("cornell-box", resx=16, resy=16)
    d = sc.pointer().contents
    red = next(e for e in fx if e["rgb"] == [0.63, 0.065, 0.05])
    got = {tuple(bits(getattr(d.materials[i].s0, k)) for k in ("c0", "c1", "c2", "scale")) for i in range(d.n_materials)}
    assert tuple(red["c0_c1_c2_scale_bits"]) in got


def test_rgb2spec_opt_writes_the_references_table_layout(tmp_path):
    """tools/rgb2spec_opt.cpp regenerates srgb.coeffs in the layout rgb2spec.rs:9-10 slices (2 header words, `res` scale
    floats, 3 * res^3 * 3 coefficients): at res 64 the shipped excerpt is that table's slices 31-32 and answers every lookup
    the full table answers the same way, bit for bit; a table of another resolution loads and interpolates too."""
    import subprocess
    tool = REPO / "path-tracer_amd" / "lib" / "rgb2spec_opt"
    full, slab = tmp_path / "srgb.coeffs", tmp_path / "z05.coeffs"
    subprocess.run([str(tool), "--full", str(full), "--slab", str(slab)], check=True, capture_output=True)
    raw = full.read_bytes()
    assert raw[:4] == b"SPEC" and len(raw) == 4 * (2 + 64 + 3 * 64 ** 3 * 3) == 9437448
    words = np.frombuffer(raw, dtype=np.float32)
    assert np.frombuffer(raw[4:8], dtype=np.uint32)[0] == 64
    t = np.arange(64) / 63.0
    ss = lambda x: x * x * (3 - 2 * x)  # noqa: E731
    assert np.array_equal(words[2:66], ss(ss(t)).astype(np.float32))
    assert np.isfinite(words[66:]).all()
    assert slab.read_bytes() == (REPO / "path-tracer_amd" / "data" / "srgb_z05.coeffs").read_bytes()   # what is shipped is what the tool writes
    rng = np.random.default_rng(11)
    cols = rng.random((300, 3)).astype(np.float32)
    cols *= np.float32(0.5) / cols.max(axis=1, keepdims=True)                  # as TristimulusSpectrum::new normalises them
    cols[:4] = [[0.5, 0.5, 0.25], [0.5, 0.0, 0.0], [0.0, 0.5, 0.5], [0.25, 0.5, 0.5]]   # equal maxima (max_by takes the last), gamut corners
    try:
        pt.rgb2spec_use(slab)
        a = np.array([pt.rgb2spec_fetch(c) for c in cols])
        pt.rgb2spec_use(full)
        b = np.array([pt.rgb2spec_fetch(c) for c in cols])
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
        # the data words rgb2spec.rs reads for one cell corner, straight from the file: max component blue (i = 2), z slice 31
        l, k, j, i = 2, 31, 10, 20
        cell = words[66 + (((l * 64 + k) * 64 + j) * 64 + i) * 3:][:3]
        raw_slab = np.frombuffer(slab.read_bytes(), dtype=np.float32)
        assert np.array_equal(cell, raw_slab[4 + 64 + (((l * 2 + 0) * 64 + j) * 64 + i) * 3:][:3])
        small = tmp_path / "res16.coeffs"
        subprocess.run([str(tool), "--res", "16", "--full", str(small)], check=True, capture_output=True)
        pt.rgb2spec_use(small)
        lam = np.arange(360.0, 831.0)
        s16, fit = pt.rgb_to_spectrum([0.63, 0.065, 0.05]), pt.rgb_to_spectrum([0.63, 0.065, 0.05], fit=True)
        assert np.abs(_sigmoid(s16, lam) - _sigmoid(fit, lam)).max() < 0.02        # a 16^3 table interpolates more coarsely
        with pytest.raises(pt.PtError):
            pt.rgb2spec_use(tmp_path / "missing.coeffs")
    finally:
        pt.rgb2spec_use(None)
