# guided on the three scenes (headline; veach-mis and veach-bidir at depth 32 = configs[4] shapes), stage times
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
ctx.set_profiling(True)
def run(name, spp, **ov):
    sc = pt.LoadedScene(name, integrator="guided", **ov); ctx.load(sc)
    best = None
    for _ in range(2):
        t = time.time(); f, s, st = ctx.render(sc.integrator, spp); dt = time.time() - t
        if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
    print(f"{name} {ov} spp={spp}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
run("cornell-box-guided", 512)
run("veach-mis", 64, resx=1280, resy=720, max_depth=32)
run("veach-bidir", 64, resx=512, resy=512, max_depth=32)
