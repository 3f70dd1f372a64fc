# Experiment: how much faster does the closest-hit walk run on bounce rays that are grouped by (direction octant, origin cell)?
# Run under rocprofv3 --kernel-trace and compare the two k_trace_batch dispatches.
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, path_tracer_amd as pt
name = sys.argv[1] if len(sys.argv) > 1 else "veach-bidir"
ctx = pt.Context(0)
sc = pt.LoadedScene(name); ctx.load(sc)
rng = np.random.default_rng(3)
d = sc.pointer().contents
pts = np.concatenate([np.ctypeslib.as_array(d.meshes[m].points, shape=(d.meshes[m].n_points, 3)) for m in range(d.n_meshes)])
lo, hi = pts.min(axis=0), pts.max(axis=0)
n = 1 << 21
c = (lo + hi) / 2
o = c + rng.uniform(-0.3, 0.3, size=(n, 3)) * (hi - lo)
dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
rays = np.concatenate([o, dirs, np.full((n, 1), np.inf)], axis=1).astype(np.float32)
h = ctx.trace(rays)
ok = h["hit"] != 0
p = rays[ok, :3] + rays[ok, 3:6] * h["t"][ok, None]
m = len(p)
nd = rng.normal(size=(m, 3)); nd /= np.linalg.norm(nd, axis=1, keepdims=True)
bounce = np.concatenate([p - 1e-3 * rays[ok, 3:6], nd, np.full((m, 1), np.inf)], axis=1).astype(np.float32)
print("bounce rays", m, flush=True)
ctx.trace(bounce)                                   # dispatch A: as they come (incoherent)
cell = np.clip(((bounce[:, :3] - lo) / (hi - lo + 1e-9) * 8).astype(np.int64), 0, 7)
octant = (bounce[:, 3] < 0) * 1 + (bounce[:, 4] < 0) * 2 + (bounce[:, 5] < 0) * 4
key = (octant << 9) | (cell[:, 0] << 6) | (cell[:, 1] << 3) | cell[:, 2]
srt = bounce[np.argsort(key, kind="stable")]
ctx.trace(srt)                                      # dispatch B: grouped

# dispatch C: only what a 256-thread workgroup could do while it writes its rays: octant order within chunks of 256
idx = np.arange(m)
local = np.lexsort((octant, idx // 256))
ctx.trace(bounce[local])
# dispatch D: the same within chunks of 1024
ctx.trace(bounce[np.lexsort((octant, idx // 1024))])
