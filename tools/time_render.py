# Times renders through the C ABI on the GPU box: best of --reps, optional per-stage kernel times (HIP events).
#   time_render.py                                   the six standard workloads (one line each), as every A/B compares them
#   time_render.py --scene veach-mis --integrator path --spp 64 --res 1280x720 [--depth D] [--stages]
#   time_render.py --spp 512,1024,2048               one line per sample count (tree size against per-sample cost)
#   time_render.py --batch 0,4194304,1048576         one line per batch size (paths in flight; 0 = default)
# Environment switches (PTGPU_LIB, PTGPU_NO_OVERLAP, PTGPU_BVH_WIDE, PTGPU_STREAM_WALK, PTGPU_WALKS, PTGPU_DIRECT_SPLAT,
# PTGPU_VERTEX_GB) are read by the library: A/B them from the shell (tools/gpu_variants.sh does it for library builds).
import argparse
import sys
import time
sys.path.insert(0, '.')
import path_tracer_amd as pt

STANDARD = [("cornell-box-guided", None, 512, {}), ("cornell-box", None, 64, dict(max_depth=8)), ("veach-mis", None, 64, dict(resx=1280, resy=720)),
            ("cornell-box", "lightpath", 128, {}), ("veach-bidir", None, 64, dict(resx=512, resy=512, max_depth=8)),
            ("veach-mis", "guided", 16, dict(resx=640, resy=360, max_depth=32))]
KINDS = {"path": pt.INTEGRATOR_PATH, "lightpath": pt.INTEGRATOR_LIGHTPATH, "guided": pt.INTEGRATOR_GUIDED}


def run(ctx, scene, integrator, spp, ov, reps, stages, batch=None):
    sc = pt.LoadedScene(scene, **ov)
    ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
    if integrator:
        integ.kind = KINDS[integrator]
    if batch is not None:
        ctx.set_batch_paths(batch)
    ctx.set_profiling(stages)
    best, kt = None, {}
    for _ in range(reps):
        t = time.time(); f, s, st = ctx.render(integ, spp); dt = time.time() - t
        if best is None or dt < best:
            best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]} if stages else {}
    what = f"{scene} {integrator or 'xml'} {ov} spp={spp}" + (f" batch={batch}" if batch is not None else "")
    print(f"{what}: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s mean={(f * s).mean():.4f} wide={ctx.traversal()} stree={st['stree_nodes']} dtree={st['dtree_nodes']}", kt, flush=True)


ap = argparse.ArgumentParser()
ap.add_argument("--scene")
ap.add_argument("--integrator", choices=sorted(KINDS))
ap.add_argument("--spp", default="")
ap.add_argument("--res", default="")
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--batch", default="")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--stages", action="store_true")
a = ap.parse_args()
ctx = pt.Context(0)
if not a.scene and not a.spp and not a.batch:
    for scene, integ, spp, ov in STANDARD:
        run(ctx, scene, integ, spp, ov, 2, True)
else:
    ov = {}
    if a.res:
        ov["resx"], ov["resy"] = (int(v) for v in a.res.split("x"))
    if a.depth:
        ov["max_depth"] = a.depth
    scene = a.scene or "cornell-box-guided"
    for spp in [int(v) for v in a.spp.split(",")] if a.spp else [512]:
        for batch in [int(v) for v in a.batch.split(",")] if a.batch else [None]:
            run(ctx, scene, a.integrator, spp, ov, a.reps, a.stages, batch)
