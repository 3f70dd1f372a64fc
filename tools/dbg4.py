import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.set_profiling(True)
for bp in (1<<21, 1<<22, 1<<23, 1<<24):
    ctx.set_batch_paths(bp)
    for _ in range(2):
        t=time.time(); f, s, st = ctx.render(sc.integrator, 512); dt=time.time()-t
    print(f"batch {bp>>20}Mi: {dt:.3f}s {st['samples']/dt*1e-6:.1f} Msamples/s", {k:round(v[0],1) for k,v in ctx.kernel_times().items() if v[1]})
