#!/usr/bin/env python3
"""Writes scenes-test/<name>/scene_v3.xml for the reference's eight furnace fixtures (pt/scenes-test/*/scene.json).

Those JSON files are in a retired scene format that no code of the reference reads any more (its parser only opens
scene_v3.xml, parser/mod.rs:1168-1171), so each is restated here in the format the loader does read, field by field:
  plymeshes[].filename icosphere_6.ply + scale s  -> <shape type="sphere"> radius s (the embedded icosphere_6.ply, parser/mod.rs:580-608)
  translation                                       -> <point name="center">
  reverse_normals: true                             -> <shape type="ply"> of scenes-test/icosphere_6_reversed.ply: the embedded mesh with every
                                                       face's vertex order reversed (written by this script), scaled by to_world. The format has
                                                       no flag for it, and a negative-determinant to_world does NOT do it: the loader reverses the
                                                       winding of a mirrored mesh precisely so that its normals keep their side
                                                       (parser/mod.rs:231-235). One-sided emission (lighting.rs:80-91) follows
                                                       normalize((p0-p1) x (p2-p1)) (geom.rs:19-37), which points outward for the PLY as shipped.
  skip: true                                        -> a shape without a <ref> (material None: pass-through surface)
  Dielectric eta Constant 1                         -> <bsdf type="dielectric"> int_ior 1 / ext_ior 1
  SimplePathIntegrator{mode} / LightPathIntegrator  -> <integrator type="path"|"lightpath"> + <string name="mode">
The inner spheres (glass, skip) get a negative-determinant to_world (mirror in x), to exercise that winding fix-up: a
sphere centred on the mirror plane stays where it was, so the image must not change.
Expected image of every scene: RGB [1, 1, 1] (README.md:6)."""
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent / "scenes-test"
SCENES = {
    # name: (integrator, mode, fov, inner)
    "simple_pt_white_sphere": ("path", "PT", 19.5, None),
    "simple_nee_white_sphere": ("path", "NEE", 19.5, None),
    "simple_mis_white_sphere": ("path", "MIS", 19.5, None),
    "simple_pt_white_sphere_glass": ("path", "PT", 19.5, "glass"),
    "simple_nee_white_sphere_glass": ("path", "NEE", 19.5, "glass"),
    "simple_mis_white_sphere_glass": ("path", "MIS", 19.5, "glass"),
    "simple_mis_white_sphere_skip": ("path", "MIS", 19.5, "skip"),
    "lt_white_sphere": ("lightpath", None, 79.5, None),
}
MIRROR = '\t\t<transform name="to_world">\n\t\t\t<scale x="-1" y="1" z="1" />\n\t\t</transform>\n'


def xml(name, integrator, mode, fov, inner):
    mode_line = f'\t\t<string name="mode" value="{mode}" />\n' if mode else ""
    glass_bsdf = ('\t<bsdf type="dielectric" id="glass">\n\t\t<float name="int_ior" value="1" />\n\t\t<float name="ext_ior" value="1" />\n\t</bsdf>\n'
                  if inner == "glass" else "")
    inner_shape = ""
    if inner == "glass":
        inner_shape = '\t<shape type="sphere">\n\t\t<float name="radius" value="5" />\n' + MIRROR + '\t\t<ref id="glass" />\n\t</shape>\n'
    elif inner == "skip":
        inner_shape = ('\t<shape type="sphere">\n\t\t<float name="radius" value="2.5" />\n\t\t<point name="center" x="0" y="0" z="5" />\n' + MIRROR +
                       '\t</shape>\n')
    return f'''<scene version="3.0.0" output="{name}.png">
	<default name="integrator" value="{integrator}" />
	<default name="spp" value="64" />
	<default name="resy" value="16" />
	<default name="resx" value="16" />
	<default name="max_depth" value="16" />
	<integrator type="$integrator">
		<integer name="max_depth" value="$max_depth" />
{mode_line}	</integrator>
	<sensor type="perspective">
		<float name="fov" value="{fov}" />
		<transform name="to_world">
			<matrix value="1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1" />
		</transform>
		<sampler type="independent">
			<integer name="sample_count" value="$spp" />
		</sampler>
		<film type="hdrfilm">
			<integer name="width" value="$resx" />
			<integer name="height" value="$resy" />
		</film>
	</sensor>
	<bsdf type="diffuse" id="white">
		<rgb name="reflectance" value="0.5, 0.5, 0.5" />
	</bsdf>
{glass_bsdf}	<shape type="ply">
		<string name="filename" value="../icosphere_6_reversed.ply" />
		<transform name="to_world">
			<scale value="10" />
		</transform>
		<ref id="white" />
		<emitter type="area">
			<rgb name="radiance" value="0.5, 0.5, 0.5" />
		</emitter>
	</shape>
{inner_shape}</scene>
'''


def write_reversed_ply():
    import numpy as np
    src = (ROOT.parent / "path-tracer_amd" / "data" / "icosphere_6.ply").read_bytes()
    end = src.index(b"end_header\n") + len(b"end_header\n")
    header = src[:end].decode()
    assert "format binary_little_endian 1.0" in header and "property list uchar uint vertex_indices" in header
    nv = int(header.split("element vertex ")[1].split()[0])
    nf = int(header.split("element face ")[1].split()[0])
    verts = src[end:end + 12 * nv]
    faces = np.frombuffer(src[end + 12 * nv:end + 12 * nv + 13 * nf], dtype=np.dtype([("n", "u1"), ("i", "<u4", 3)])).copy()
    assert (faces["n"] == 3).all()
    faces["i"] = faces["i"][:, ::-1]
    (ROOT / "icosphere_6_reversed.ply").write_bytes(src[:end] + verts + faces.tobytes())


if __name__ == "__main__":
    ROOT.mkdir(exist_ok=True)
    write_reversed_ply()
    for name, (integrator, mode, fov, inner) in SCENES.items():
        d = ROOT / name
        d.mkdir(parents=True, exist_ok=True)
        (d / "scene_v3.xml").write_text(xml(name, integrator, mode, fov, inner))
        print("wrote", d / "scene_v3.xml")
