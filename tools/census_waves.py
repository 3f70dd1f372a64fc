# Census of the walk (nodes visited / triangles tested): per ray with the in-tree library, per wave-trip with a
# -DPT_CENSUS_WAVE build (PTGPU_LIB). Usage: census_waves.py [scene] [spp]
import sys
sys.path.insert(0, '.')
import path_tracer_amd as pt
name = sys.argv[1] if len(sys.argv) > 1 else "cornell-box-guided"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = pt.Context(0)
sc = pt.LoadedScene(name); ctx.load(sc)
ctx.set_census(True)
f, s, st = ctx.render(sc.integrator, spp)
print(name, spp, "census", ctx.census(), {k: st[k] for k in ("extend_rays", "shadow_rays", "samples") if k in st}, flush=True)
