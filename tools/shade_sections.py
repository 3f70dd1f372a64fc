# Section shares of k_shade_guided (needs a -DPT_DBG_TIMING build named by PTGPU_LIB). Prints from ptgpu_guided_end.
import sys
sys.path.insert(0, '.')
import path_tracer_amd as pt
ctx = pt.Context(0)
sc = pt.LoadedScene("cornell-box-guided"); ctx.load(sc)
for _ in range(2):
    ctx.render(sc.integrator, 512)
