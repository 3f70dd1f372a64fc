# Walk-heavy workloads only (traversal experiments).
import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import path_tracer_amd as pt
ctx = pt.Context(0)
def run(name, spp, **ov):
    sc = pt.LoadedScene(name, **ov); ctx.load(sc)
    ctx.set_profiling(True)
    best=None
    for _ in range(2):
        t=time.time(); f, s, st = ctx.render(sc.integrator, spp); dt=time.time()-t
        best = dt if best is None else min(best,dt)
    print(f"{name} {ov} spp={spp}: {best:.3f}s {st['samples']/best*1e-6:.1f} Msamples/s", {k:round(v[0],1) for k,v in ctx.kernel_times().items() if v[1]})
run("cornell-box", 64, max_depth=8)
run("veach-mis", 64, resx=1280, resy=720)
run("veach-bidir", 64, resx=512, resy=512, max_depth=8)
