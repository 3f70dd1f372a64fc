# Guided renders of the headline workload (for profilers). Usage: render_once.py [spp] [reps]
import sys
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 1):
    f, s, st = ctx.render(sc.integrator, int(sys.argv[1]) if len(sys.argv) > 1 else 512)
print(st['samples'], (f * s).mean())
