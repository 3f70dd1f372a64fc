# Guided headline workload, best of 3, with per-stage kernel times (HIP events).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.set_profiling(True)
best = None
for _ in range(3):
    t = time.time(); f, s, st = ctx.render(sc.integrator, 512); dt = time.time() - t
    if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
print(f"guided 512: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s mean={(f*s).mean():.4f} stree={st['stree_nodes']} dtree={st['dtree_nodes']}", kt, flush=True)
