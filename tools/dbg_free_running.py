import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_bind as ob, path_tracer_amd as pt
ctx = pt.Context(0)
for name, res, spp, depth in (("cornell-box-guided",(64,64),63,8),("veach-bidir",(64,64),31,8),("cornell-box-guided",(64,64),127,8),("veach-bidir",(64,64),63,8)):
    sc = pt.LoadedScene(name, resx=res[0], resy=res[1], max_depth=depth, integrator="guided")
    o = ob.OracleScene(sc); ctx.load(sc)
    integ = pt.IntegratorDesc.from_buffer_copy(sc.integrator); integ.spatial_threshold = 200.0
    gf, gs, gst = ctx.render(integ, spp); cf, cs, cst = o.render(sc.camera, integ, spp)
    g, c = gf*gs, cf*cs; k=8
    gb = g.reshape(res[1]//k,k,res[0]//k,k,3).mean(axis=(1,3)); cb = c.reshape(res[1]//k,k,res[0]//k,k,3).mean(axis=(1,3))
    print(name, spp, "stree", gst["stree_nodes"], cst["stree_nodes"], "dtree", gst["dtree_nodes"], cst["dtree_nodes"], "rays", gst["extend_rays"]/cst["extend_rays"]-1,
          "median", np.median(gb)/np.median(cb)-1, "mean", g.mean()/c.mean()-1, "relmse blocks", ob.rel_mse(np.minimum(gb,5),np.minimum(cb,5)))
