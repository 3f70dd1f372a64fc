"""Ray counts of renders in the 59-level chain scene (tests/scenes_util.py: deep_chain_scene), binary and wide walk, and
the oracle's: python3 tools/deep_chain_diag.py  (PTGPU_STREAM_WALK=0 for the non-streamed wide extend)."""
import sys, pathlib
import numpy as np
ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import path_tracer_amd as pt
import scenes_util as su
import oracle_bind as ob

sc, cen, size = su.deep_chain_scene(180, 1.6, lit=True)
ctx = pt.Context(0)
ctx.set_scene(sc)
cam = pt.make_camera(su.IDENTITY, 120.0, 96, 96)
ctx.set_camera(cam)
o = ob.OracleScene(sc)
for kind, depth in ((pt.INTEGRATOR_PATH, 1), (pt.INTEGRATOR_PATH, 2), (pt.INTEGRATOR_PATH, 4), (pt.INTEGRATOR_LIGHTPATH, 1), (pt.INTEGRATOR_LIGHTPATH, 4)):
    integ = pt.make_integrator(kind, depth, pt.MODE_MIS)
    _, _, cst = o.render(cam, integ, 2)
    row = {"oracle": (cst["extend_rays"], cst["shadow_rays"])}
    for wide in (False, True):
        ctx.set_traversal(wide)
        _, _, st = ctx.render(integ, 2)
        row["wide" if wide else "binary"] = (st["extend_rays"], st["shadow_rays"])
    print(kind, depth, row, flush=True)
# rays like the renders' bounce rays: from any triangle of the chain, in any direction
rng = np.random.default_rng(1)
k = rng.integers(0, 180, 200000)
origin = cen[k] + rng.normal(size=(len(k), 3)) * size[k, None] * 0.2
d = rng.normal(size=(len(k), 3))
d /= np.linalg.norm(d, axis=1, keepdims=True)
towards = rng.random(len(k)) < 0.5   # half of them at another triangle
k2 = rng.integers(0, 180, len(k))
d[towards] = (cen[k2] - origin)[towards]
d[towards] /= np.linalg.norm(d[towards], axis=1, keepdims=True)
rays = np.concatenate([origin, d, np.full((len(k), 1), np.inf)], axis=1).astype(np.float32)
c = o.trace(rays)
for wide in (False, True):
    ctx.set_traversal(wide)
    g = ctx.trace(rays)
    bad = np.nonzero((g["hit"] != c["hit"]) | (g["triangle_index"] != c["triangle_index"]) | (g["t"].view(np.uint32) != c["t"].view(np.uint32)))[0]
    print("wide" if wide else "binary", "trace mismatches vs oracle:", len(bad), "of", len(rays), "hits", int(c["hit"].sum()), flush=True)
    for i in bad[:5]:
        print("  ray", rays[i], "gpu", g[i], "oracle", c[i], flush=True)
