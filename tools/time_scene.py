# One shipped scene, one integrator, best of 3 (A/B of environment switches): python3 tools/time_scene.py <scene> <path|lightpath|guided> <spp>
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
name, kind, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
ctx = pt.Context(0)
sc = pt.LoadedScene(name)
ctx.load(sc)
integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
integ.kind = {"path": pt.INTEGRATOR_PATH, "lightpath": pt.INTEGRATOR_LIGHTPATH, "guided": pt.INTEGRATOR_GUIDED}[kind]
ctx.set_profiling(True)
best = None
for _ in range(3):
    t = time.time(); f, s, st = ctx.render(integ, spp); dt = time.time() - t
    if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
print(f"{name} {kind} {spp} spp wide={ctx.traversal()}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
