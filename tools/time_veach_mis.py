# veach-mis path only, best of 4 (for A/B of walker builds).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("veach-mis", resx=1280, resy=720); ctx.load(sc)
ctx.set_profiling(True)
best = None
for _ in range(4):
    t = time.time(); f, s, st = ctx.render(sc.integrator, 128); dt = time.time() - t
    if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
print(f"veach-mis 128: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
