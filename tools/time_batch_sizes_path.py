# path / lightpath workloads at several batch sizes (paths in flight). Usage: time_batch_sizes_path.py [sizes...]
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
ctx.set_profiling(True)
for name, spp, ov in (("veach-mis", 256, dict(resx=1280, resy=720)), ("cornell-box", 64, dict(max_depth=8)), ("cornell-box", 256, dict(integrator="lightpath"))):
    sc = pt.LoadedScene(name, **ov); ctx.load(sc)
    for bp in [int(a) for a in sys.argv[1:]] or [0, 1 << 23, 1 << 24]:
        ctx.set_batch_paths(bp)
        best = None
        for _ in range(3):
            t = time.time(); f, s, st = ctx.render(sc.integrator, spp); dt = time.time() - t
            if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
        print(f"{name} {ov} batch {bp}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
    ctx.set_batch_paths(0)
