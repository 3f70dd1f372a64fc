# Exercises the smaller-batch fallback: a bystander allocation leaves the device ~30 GB, then the headline workloads run.
# Usage: oom_fallback.py [free_gb]
import sys, time
sys.path.insert(0, '.')
import torch
import path_tracer_amd as pt
free_gb = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
free, total = torch.cuda.mem_get_info(0)
hog = torch.empty(int(max(0, free - free_gb * 2**30)), dtype=torch.uint8, device="cuda:0")
print(f"device {total / 2**30:.0f} GB, left free {torch.cuda.mem_get_info(0)[0] / 2**30:.1f} GB", flush=True)
ctx = pt.Context(0)
for name, spp, ov in (("cornell-box-guided", 512, {}), ("veach-mis", 128, dict(resx=1280, resy=720))):
    sc = pt.LoadedScene(name, **ov); ctx.load(sc)
    t = time.time(); f, s, st = ctx.render(sc.integrator, spp); dt = time.time() - t
    print(f"{name}: {dt:.3f}s {st['samples'] / dt * 1e-6:.1f} Msamples/s mean={(f * s).mean():.4f} samples={st['samples']}", flush=True)
