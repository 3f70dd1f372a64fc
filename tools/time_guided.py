import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
best = None
for _ in range(3):
    t = time.time(); f, s, st = ctx.render(sc.integrator, 512); dt = time.time() - t
    best = dt if best is None else min(best, dt)
print(f"guided 512: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", flush=True)
