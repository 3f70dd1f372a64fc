import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.set_profiling(True)
for (spp, idx, n) in ((512, 0, 1), (4096, 0, 8), (4096, 5, 8)):
    for _ in range(2):
        t=time.time(); f, s, st = ctx.render(sc.integrator, spp, idx, n); dt=time.time()-t
    print(f"spp={spp} shard {idx}/{n}: {dt:.3f}s own samples {st['samples']} -> {st['samples']/dt*1e-6:.1f} Msamples/s stree={st['stree_nodes']}", {k:round(v[0],1) for k,v in ctx.kernel_times().items() if v[1]})
