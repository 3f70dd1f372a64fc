#!/usr/bin/env python3
"""Writes tests/golden/rgb2spec_scene_colours.json: the ten non-grey colours of the shipped scenes and the
spectrum TristimulusSpectrum::new gives each (spectrum/mod.rs:272-302) — scale and the three coefficients
rgb2spec::fetch_coeffs (rgb2spec.rs:8-54) reads out of the coefficient table, as float bit patterns. The table is this
repo's regeneration of the reference's missing srgb.coeffs (tools/rgb2spec_opt.cpp; `make rgb2spec-table`); the shipped
excerpt path-tracer_amd/data/srgb_z05.coeffs must reproduce these values bit for bit (tests/test_host_cpu.py). Run from the
repo root after `make`."""
import glob
import json
import re
import struct
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import path_tracer_amd as pt  # noqa: E402

bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]  # noqa: E731
colours = {}
for f in sorted(glob.glob(str(REPO / "scenes" / "*" / "scene_v3.xml"))):
    for m in re.finditer(r'<rgb[^>]*value="([^"]+)"', Path(f).read_text()):
        v = tuple(float(x) for x in m.group(1).split(","))
        if len(v) == 3 and not v[0] == v[1] == v[2]:
            colours.setdefault(v, []).append(Path(f).parent.name)
out = []
for rgb, scenes in sorted(colours.items()):
    s = pt.rgb_to_spectrum(rgb)
    out.append({"rgb": list(rgb), "scenes": sorted(set(scenes)), "c0_c1_c2_scale_bits": [bits(s.c0), bits(s.c1), bits(s.c2), bits(s.scale)],
                "c0_c1_c2_scale": [s.c0, s.c1, s.c2, s.scale]})
(Path(__file__).resolve().parent / "rgb2spec_scene_colours.json").write_text(json.dumps(out, indent=1) + "\n")
print(len(out), "colours")
