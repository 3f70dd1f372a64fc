# Scene set-up time with the BVH built on the host and on the device (same tree): shipped scenes and large soups.
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import path_tracer_amd as pt, scenes_util as su
ctx = pt.Context(0)
cases = [("veach-mis", pt.LoadedScene("veach-mis")), ("veach-bidir", pt.LoadedScene("veach-bidir"))]
for n in (200000, 1000000):
    cases.append((f"soup {n}", su.random_soup(np.random.default_rng(1), n, n_meshes=3, extent=40.0)))
for name, sc in cases:
    out = []
    for mode in (0, 1):
        ctx.set_bvh_builder(mode)
        best = None
        for _ in range(2):
            t = time.time(); ctx.set_scene(sc); dt = time.time() - t
            best = dt if best is None else min(best, dt)
        out.append(best)
    print(f"{name}: scene_create with the host builder {out[0]*1e3:.1f} ms, with the device builder {out[1]*1e3:.1f} ms (on device: {ctx.bvh_built_on_device()})", flush=True)
