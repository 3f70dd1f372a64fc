import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    best = 1e9
    for _ in range(3):
        t=time.time(); f, s, st = ctx.render(sc.integrator, 512); best=min(best, time.time()-t)
    print(f"profiling={prof}: {best:.4f}s {st['samples']/best*1e-6:.1f} Msamples/s launches={st['kernel_launches']}")
