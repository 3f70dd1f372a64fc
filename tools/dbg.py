import sys, time
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, path_tracer_amd as pt, scenes_util as su, oracle_bind as ob
ctx = pt.Context(0)
for name, kind, res, spp, ov in [("cornell-box", pt.INTEGRATOR_PATH, (128,128), 64, dict(max_depth=8)),
                                 ("veach-mis", pt.INTEGRATOR_PATH, (160,96), 64, {}),
                                 ("veach-bidir", pt.INTEGRATOR_PATH, (128,128), 32, {}),
                                 ("cornell-box", pt.INTEGRATOR_LIGHTPATH, (128,128), 64, dict(max_depth=8))]:
    sc = pt.LoadedScene(name, resx=res[0], resy=res[1], **ov)
    o = ob.OracleScene(sc); ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator); integ.kind = kind
    t=time.time(); gf, gs, gst = ctx.render(integ, spp); tg=time.time()-t
    t=time.time(); cf, cs, cst = o.render(sc.camera, integ, spp); tc=time.time()-t
    g, c = gf*gs, cf*cs
    rel = np.abs(g-c).sum(axis=2)/(np.abs(c).sum(axis=2)+1e-6)
    print(name, kind, "gpu %.3fs cpu %.3fs"%(tg,tc), "means", g.mean(), c.mean(), "median rel", np.median(rel), "p90", np.percentile(rel,90), "p99", np.percentile(rel,99), "frac<1e-3", (rel<1e-3).mean(),
          "rays", gst["extend_rays"], cst["extend_rays"], gst["shadow_rays"], cst["shadow_rays"], gst["segments"], cst["segments"])
