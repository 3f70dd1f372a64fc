# veach-mis (path) and veach-mis guided at reduced size, best of 3: for A/B of walker builds (PTGPU_LIB).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
def run(name, spp, reps=3, **ov):
    sc = pt.LoadedScene(name, **ov); ctx.load(sc)
    ctx.set_profiling(True)
    best = None
    for _ in range(reps):
        t = time.time(); f, s, st = ctx.render(sc.integrator, spp); dt = time.time() - t
        if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
    print(f"{name} {ov} spp={spp}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, flush=True)
run("veach-mis", 128, resx=1280, resy=720)
run("veach-bidir", 64, resx=512, resy=512, max_depth=8)
run("veach-mis", 32, resx=640, resy=360, integrator="guided", max_depth=32)
