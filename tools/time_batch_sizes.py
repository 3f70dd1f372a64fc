# Guided headline workload at several batch sizes (paths in flight): stage times vs footprint of the per-path buffers.
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.set_profiling(True)
for bp in [int(a) for a in sys.argv[1:]] or [0, 1 << 22, 1 << 21, 1 << 20]:
    ctx.set_batch_paths(bp)
    best = None
    for _ in range(2):
        t = time.time(); f, s, st = ctx.render(sc.integrator, 512); dt = time.time() - t
        if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
    print(f"batch {bp}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
