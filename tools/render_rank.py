# One rank of an N-GPU strong-scaling guided render (shard 0 of N), statistics of the other ranks stood in for by scaling
# this rank's own: for profilers. Usage: render_rank.py [world] [reps]
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    n = ctx.guided_begin(sc.integrator, 512, 0, world)
    t0 = time.time(); per = []
    for w in range(n):
        t = time.time(); ctx.guided_wave(w); per.append(round((time.time() - t) * 1e3, 2))
        if w + 1 < n:
            ctx.guided_stats_set(ctx.guided_stats_host() * float(world))
    f, s, st = ctx.guided_end()
    print(f"rep {rep}: {1e3 * (time.time() - t0):.1f} ms, waves {per}", flush=True)
