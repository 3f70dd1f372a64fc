import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_bind as ob, path_tracer_amd as pt
ctx = pt.Context(0)
for name, res in (("veach-bidir",(64,64)),("cornell-box-guided",(64,64)),("veach-mis",(80,48))):
    for depth in (4, 8, 32):
        sc = pt.LoadedScene(name, resx=res[0], resy=res[1], max_depth=depth, integrator="guided")
        o = ob.OracleScene(sc); ctx.load(sc)
        integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator)
        ctx.guided_begin(integ, 1); ctx.guided_wave(0); t = ctx.guided_tree(); gf, gs, gst = ctx.guided_end()
        O = ob.OracleSdTree.fresh(o, 1000.0); fo, sto = O.wave(o, sc.camera, integ, 0, 0, 1); e = O.export()
        rel = np.abs(gf - fo).sum(axis=2) / (np.abs(fo).sum(axis=2) + 1e-6)
        print(name, depth, "weights", t["c_weight"].sum(), e["c_weight"].sum(), "rays", [gst[k] for k in ("extend_rays","shadow_rays","segments")],
              [sto[k] for k in ("extend_rays","shadow_rays","segments")], "pixels differing >1e-3:", int((rel > 1e-3).sum()), "sum rad", t["c_rad"].sum(), e["c_rad"].sum())
