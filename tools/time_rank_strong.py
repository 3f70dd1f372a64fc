# What ONE rank of an N-GPU STRONG-scaling run does (512 spp total dealt over N ranks, shard 0), timed on one GPU: the other
# ranks' statistics are stood in for by scaling this rank's own by N between waves (so the tree grows as it would), the
# collectives themselves are not included. Ideal time = single-GPU step / N.
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
ctx.render(sc.integrator, 64)
t = time.time(); f, s, st = ctx.render(sc.integrator, 512); one = time.time() - t
print(f"N=1: 512 spp in {one:.3f} s", flush=True)
ctx.set_profiling(True)
for world in (2, 4, 8, 8):
    n = ctx.guided_begin(sc.integrator, 512, 0, world)
    wave_s = 0.0; per = []
    for w in range(n):
        t = time.time(); ctx.guided_wave(w); dt = time.time() - t; wave_s += dt; per.append(round(dt * 1e3, 2))
        if w + 1 < n:
            ctx.guided_stats_set(ctx.guided_stats_host() * float(world))
    t = time.time(); f, s, st = ctx.guided_end(); wave_s += time.time() - t
    print('   ms per wave:', per)
    print('   stage ms:', {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]})
    print(f"rank 0 of {world} (strong): {st['samples']} samples in {wave_s*1e3:.1f} ms; ideal {one/world*1e3:.1f} ms -> efficiency {one / world / wave_s:.3f} before collectives", flush=True)
