import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, path_tracer_amd as pt
ctx = pt.Context(0)
def run(name, spp, prof, **ov):
    sc = pt.LoadedScene(name, **ov); ctx.load(sc)
    ctx.set_profiling(prof)
    t=time.time(); f, s, st = ctx.render(sc.integrator, spp); dt=time.time()-t
    print(f"{name} {ov} spp={spp} prof={prof}: {dt:.3f}s {st['samples']/dt*1e-6:.1f} Msamples/s {(st['extend_rays']+st['shadow_rays'])/dt*1e-6:.1f} Mrays/s launches={st['kernel_launches']} stree={st['stree_nodes']} dtree={st['dtree_nodes']} mean={(f*s).mean():.4f}")
    if prof: print("   ", {k:(round(v[0],2),v[1]) for k,v in ctx.kernel_times().items()})
run("cornell-box", 64, False, max_depth=8)
run("cornell-box", 64, False, max_depth=8)
run("cornell-box", 64, True, max_depth=8)
run("cornell-box-guided", 512, False)
run("cornell-box-guided", 512, True)
run("cornell-box", 64, True, integrator="lightpath", max_depth=8)
run("veach-mis", 64, True, resx=1280, resy=720)
