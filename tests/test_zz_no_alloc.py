"""ptgpu_trace_batch_noalloc (include/ptgpu.h): the entry point that allocates nothing must return the very hit records
of ptgpu_trace_batch, on two calls in a row that share the library's static scratch array, the second one smaller — and
must leave the device's free memory where it found it."""
import ctypes as C

import numpy as np
import pytest

import path_tracer_amd as pt
import scenes_util as su

pytestmark = pytest.mark.gpu


def test_noalloc_trace_equals_trace_batch_twice():
    ctx = pt.Context(0)
    try:
        sc = pt.LoadedScene("veach-bidir")
        ctx.set_scene(sc)
        b = ctx.scene_info()["bounds"]
        rng = np.random.default_rng(21)
        big = su.random_rays(rng, 200000, b[:3], b[3:])
        small = su.random_rays(rng, 1237, b[:3], b[3:], finite_tmax=True)
        for rays in (big, small):          # second call: fewer rays over the first call's leftovers in the scratch
            want = ctx.trace(rays)
            got = ctx.trace_noalloc(rays)
            assert want["hit"].mean() > 0.2
            assert got.tobytes() == want.tobytes()
        # the property the entry point exists for: no device allocation (a per-call hipMalloc would show as a dip in free memory
        # while the call runs, and a leaked one afterwards; the runtime's own pools are warm after the calls above)
        hip = C.CDLL("libamdhip64.so")
        free_b, total_b = C.c_size_t(), C.c_size_t()

        def free_now():
            assert hip.hipMemGetInfo(C.byref(free_b), C.byref(total_b)) == 0
            return free_b.value

        before = free_now()
        for _ in range(3):
            ctx.trace_noalloc(small)
            assert free_now() == before
        ctx.trace(small)            # the allocating twin gives its buffers back as well, but through hipMalloc / hipFree
        assert len(ctx.trace_noalloc(np.zeros((0, 7), dtype=np.float32))) == 0
        with pytest.raises(pt.PtError, match="at most"):
            ctx.trace_noalloc(np.zeros((pt.NOALLOC_MAX_RAYS + 1, 7), dtype=np.float32))
        # the limit itself is accepted
        edge = su.random_rays(rng, pt.NOALLOC_MAX_RAYS, b[:3], b[3:])
        assert ctx.trace_noalloc(edge).tobytes() == ctx.trace(edge).tobytes()
    finally:
        ctx.close()
