# Guided throughput against the sample count (tree size effect on per-sample cost; what one rank of an N-GPU weak-scaling run sees).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.render(sc.integrator, 64)
for spp in (512, 1024, 2048, 4096):
    t = time.time(); f, s, st = ctx.render(sc.integrator, spp); dt = time.time() - t
    print(f"spp={spp}: {dt:.3f}s {st['samples'] / dt * 1e-6:.1f} Msamples/s stree={st['stree_nodes']} dtree={st['dtree_nodes']}", flush=True)
