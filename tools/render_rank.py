# One rank of an N-GPU strong-scaling guided render (shard 0 of N), statistics of the other ranks stood in for by scaling
# this rank's own: for profilers. Usage: render_rank.py [world] [reps] [paths per batch, 0 = default]
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
if len(sys.argv) > 3: ctx.set_batch_paths(int(sys.argv[3]))
for rep in range(int(sys.argv[2]) if len(sys.argv) > 2 else 2):
    n = ctx.guided_begin(sc.integrator, 512, 0, world)
    t0 = time.time(); per = []; swap = 0.0
    for w in range(n):
        t = time.time(); ctx.guided_wave(w); per.append(round((time.time() - t) * 1e3, 2))
        if w + 1 < n:
            t = time.time(); ctx.guided_stats_set(ctx.guided_stats_host() * float(world)); swap += time.time() - t
    f, s, st = ctx.guided_end()
    total = time.time() - t0
    print(f"rep {rep}: {1e3 * total:.1f} ms, of which {1e3 * swap:.1f} ms the host-side stand-in for the all-reduce; render alone {1e3 * (total - swap):.1f} ms; waves {per}", flush=True)
