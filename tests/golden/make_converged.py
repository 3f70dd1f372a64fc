#!/usr/bin/env python3
"""Writes tests/golden/converged_cornell_64.npz: the CPU oracle's converged images of the Cornell box at 64 x 64 for the
three integrators (16 384 spp; guided 16 383 = waves 1..8192) and, as the yardstick for "relMSE vs converged reference"
(BASELINE.json metric), the oracle's OWN relMSE at 1 024 spp (guided 1 023) against them. The GPU tests and
bench.py compare the HIP path's 1 024-spp image with the same references. Run from the repo root (~10 min on 8 cores)."""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import oracle_bind as ob  # noqa: E402
import path_tracer_amd as pt  # noqa: E402

RES, DEPTH = 64, 16
out = {}
for name, kind, scene in (("path", pt.INTEGRATOR_PATH, "cornell-box"), ("lightpath", pt.INTEGRATOR_LIGHTPATH, "cornell-box"),
                          ("guided", pt.INTEGRATOR_GUIDED, "cornell-box-guided")):
    sc = pt.LoadedScene(scene, resx=RES, resy=RES, max_depth=DEPTH)
    o = ob.OracleScene(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    integ.kind = kind
    big, small = (16383, 1023) if kind == pt.INTEGRATOR_GUIDED else (16384, 1024)
    t0 = time.time()
    rf, rs, _ = o.render(sc.camera, integ, big)
    cf, cs, _ = o.render(sc.camera, integ, small)
    ref, own = rf * rs, cf * cs
    out[f"{name}_ref"] = ref
    out[f"{name}_ref_spp"] = big
    out[f"{name}_spp"] = small
    out[f"{name}_own"] = own          # the oracle's image at the test spp (path / lightpath: the GPU walks the very same paths)
    out[f"{name}_oracle_relmse"] = ob.rel_mse(own, ref)
    terms = np.sort(((own - ref) ** 2 / (ref ** 2 + 1e-2)).mean(axis=2).ravel())
    out[f"{name}_oracle_relmse_trimmed"] = float(terms[: int(len(terms) * 0.99)].mean())  # without the 1 % worst pixels (fireflies behind the glass cube)
    print(name, "ref spp", big, "oracle relMSE at", small, "spp:", out[f"{name}_oracle_relmse"], "trimmed", out[f"{name}_oracle_relmse_trimmed"], f"({time.time() - t0:.0f} s)", flush=True)
np.savez_compressed(Path(__file__).resolve().parent / "converged_cornell_64.npz", **out)
