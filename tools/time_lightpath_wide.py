# lightpath on veach-mis (four-wide walk), best of 3: A/B of the streamed walkers (PTGPU_STREAM_WALK=0|1).
import sys, time
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("veach-mis", resx=1280, resy=720)
ctx.load(sc)
integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
integ.kind = pt.INTEGRATOR_LIGHTPATH
ctx.set_profiling(True)
best = None
for _ in range(3):
    t = time.time(); f, s, st = ctx.render(integ, 64); dt = time.time() - t
    if best is None or dt < best: best, kt = dt, {k: round(v[0], 1) for k, v in ctx.kernel_times().items() if v[1]}
print(f"veach-mis lightpath 64 spp: {best:.3f}s {st['samples'] / best * 1e-6:.1f} Msamples/s", kt, st["extend_rays"], st["shadow_rays"], float(f.sum()), flush=True)
