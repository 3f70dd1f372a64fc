# One rank of an N-GPU guided render of the headline scene (shard 0 of N), timed on one GPU: the statistics of the other
# ranks are stood in for by scaling this rank's own by N between waves (so the tree grows as it would); the collectives
# themselves are not included. An EMULATION: no multi-GPU run is behind these numbers.
#   render_rank.py [world=8] [reps=2] [paths per batch, 0 = default] [--weak]
# strong (default): 512 spp dealt over `world` ranks, ideal = single-GPU step / world. --weak: 512 * world spp, ideal = the
# single-GPU step. Also what the profilers run (tools/gpu_ktrace.sh).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
args = [a for a in sys.argv[1:] if not a.startswith("--")]
weak = "--weak" in sys.argv
world = int(args[0]) if len(args) > 0 else 8
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
if len(args) > 2: ctx.set_batch_paths(int(args[2]))
ctx.render(sc.integrator, 512)  # (buffers grow to the job's size here, not in the timed render)
t = time.time(); ctx.render(sc.integrator, 512); one = time.time() - t
print(f"N=1: 512 spp in {1e3 * one:.1f} ms", flush=True)
for rep in range(int(args[1]) if len(args) > 1 else 2):
    n = ctx.guided_begin(sc.integrator, 512 * world if weak else 512, 0, world)
    t0 = time.time(); per = []; swap = 0.0
    for w in range(n):
        t = time.time(); ctx.guided_wave(w); per.append(round((time.time() - t) * 1e3, 2))
        if w + 1 < n:
            t = time.time(); ctx.guided_stats_set(ctx.guided_stats_host() * float(world)); swap += time.time() - t
    f, s, st = ctx.guided_end()
    mine = time.time() - t0 - swap
    ideal = one if weak else one / world
    print(f"rep {rep}: rank 0 of {world} ({'weak' if weak else 'strong'}): {1e3 * mine:.1f} ms (+ {1e3 * swap:.1f} ms host-side stand-in for the all-reduce, not counted); "
          f"ideal {1e3 * ideal:.1f} ms -> {ideal / mine:.3f} before collectives; waves {per}; stree {st['stree_nodes']} dtree {st['dtree_nodes']}", flush=True)
