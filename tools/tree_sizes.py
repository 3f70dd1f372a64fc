# Sizes of the per-leaf collector quadtrees wave by wave (how many exceed what k_guide_splat_bins / k_push_lazy /
# the refine kernels stage in LDS). Usage: tree_sizes.py [world]
import sys, time
sys.path.insert(0, '.')
import numpy as np
import path_tracer_amd as pt
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
n = ctx.guided_begin(sc.integrator, 512, 0, world)
for w in range(n):
    ctx.guided_wave(w)
    t = ctx.guided_tree()
    cs = t["leaf"][:, 3]
    print(f"wave {w}: leaves {len(cs)} collector nodes per leaf: mean {cs.mean():.0f} p50 {np.percentile(cs, 50):.0f} p99 {np.percentile(cs, 99):.0f} max {cs.max()}"
          f"  >192: {(cs > 192).sum()} (nodes {cs[cs > 192].sum()} of {cs.sum()})  >256: {(cs > 256).sum()}", flush=True)
    if w + 1 < n and world > 1:
        ctx.guided_stats_set(ctx.guided_stats_host() * float(world))
ctx.guided_end()
