import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("lt_white_sphere", collection="/root/repo/scenes-test"); ctx.load(sc)
for spp in (1<<14, 1<<16, 1<<18, 1<<20, 1<<22):
    f,s,st = ctx.render(sc.integrator, spp); im = f*s
    print("lt", spp, im.mean(axis=(0,1))-1, "edge rows/cols mean", im[0].mean()-1, im[-1].mean()-1, im[:,0].mean()-1, im[:,-1].mean()-1, "interior", im[1:-1,1:-1].mean()-1, round(st["render_ms"]), flush=True)
for name in ("simple_mis_white_sphere_glass","simple_mis_white_sphere"):
    for d in (16, 64):
        sc = pt.LoadedScene(name, collection="/root/repo/scenes-test", max_depth=d); ctx.load(sc)
        f,s,st = ctx.render(sc.integrator, 16384); print(name, d, (f*s).mean(axis=(0,1))-1, ctx.guard_counts(), flush=True)
