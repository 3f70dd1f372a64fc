import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.render(sc.integrator, 64)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rep in range(2):
    n = ctx.guided_begin(sc.integrator, 512, 0, world)
    for w in range(n):
        t = time.time(); ctx.guided_wave(w); dt = time.time() - t
        print(f"== wave {w}: {dt*1e3:.2f} ms", file=sys.stderr, flush=True)
        if w + 1 < n:
            ctx.guided_stats_set(ctx.guided_stats_host() * float(world))
    ctx.guided_end()
