"""Sharded render driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The reference is a single-process CPU renderer (tiles over rayon, integrator/mod.rs:34-46); this is how its per-sample
loop is dealt over N GPUs without changing any sample:
  * path / lightpath: rank r renders sample indices i = r, r + N, ... of every pixel (ptgpu_render_shard);
  * guided: in every wave, tile t of pass k belongs to rank (t + k) % N (pt_guided.h: k_raygen_guided); after each wave
    but the last the collector statistics are summed IN PLACE on the library's buffer with one all_reduce, then every
    rank refines the same tree;
  * the film sums are reduced onto rank 0 once at the end, from the library's device buffer (ptgpu_film_device): no
    host round trip.
launch_ranks() starts the N rank processes of a one-node job (python -m torch.distributed.run) from a parent that has
not touched the GPU.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import numpy as np


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks: int, script: str, argv, env=None, timeout=None) -> int:
    """Runs `script argv...` as n_ranks processes of one node under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set for each; rendezvous on 127.0.0.1) and returns its exit code. The children inherit stdout, so the one
    JSON line rank 0 prints is this process's output. Must be called BEFORE the caller makes any HIP / torch.cuda call:
    a process that has initialised the GPU must not start or become another program on this pool."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script, *argv]
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL over xGMI needs it on this pool
    return subprocess.run(cmd, env=e, timeout=timeout).returncode


def samples_of_shard(spp: int, rank: int, world: int) -> range:
    """Sample indices rank `rank` renders for every pixel (ptgpu.hip: render_path_like)."""
    return range(rank, spp, world)


def shard_of_pair(tile: int, pass_index: int, world: int) -> int:
    """Owner of (pass, tile) in a guided wave (pt_guided.h: k_raygen_guided)."""
    return (tile + pass_index) % world


class DeviceArray:
    """Raw device pointer -> torch tensor (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, count, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr, "data": (ptr, False), "version": 2}


def device_stats_tensor(torch, dev):
    """wrap(ptr, count[, typestr]) for a libptgpu context: a tensor aliasing a device buffer of the library (float32
    statistics by default, "<f8" for the film)."""
    return lambda ptr, count, typestr="<f4": torch.as_tensor(DeviceArray(ptr, count, typestr), device=dev)


def _accepts_typestr(fn) -> bool:
    import inspect
    try:
        params = list(inspect.signature(fn).parameters.values())
    except (TypeError, ValueError):
        return False
    return len(params) >= 3 or any(p.kind == p.VAR_POSITIONAL for p in params)


def render_sharded(ctx, integ, spp, rank, world, guided, dist=None, wrap_stats=None, film_buf=None, sync=lambda: None, collectives=None):
    """One render of the whole job on this rank's shard. Returns (film sums [complete on rank 0], film_scale, stats).

    ctx       : object with render(integ, spp, rank, world) and guided_{begin,wave,stats,end} (path_tracer_amd.Context)
    dist      : torch.distributed (initialised) when world > 1
    wrap_stats: (ptr_or_buffer, count[, typestr]) -> tensor aliasing a buffer of the context: the float32 statistics by
                default, the float64 film with typestr "<f8" (device_stats_tensor). A callable without the third
                parameter (or None, for path / lightpath) sends the film through the host copy and `film_buf` instead
    film_buf  : preallocated float64 tensor (H, W, 3) on the collective's device
    sync      : waits for the collective's stream (torch.cuda.synchronize on GPU)
    collectives: run the collectives even when world == 1 (rehearsal of the RCCL path on a one-GPU box); default world > 1
    """
    if collectives is None:
        collectives = world > 1
    # reduce straight from the library's film buffer where the caller's wrapper can alias it
    on_device = collectives and hasattr(ctx, "film_device") and wrap_stats is not None and _accepts_typestr(wrap_stats)
    if guided:
        n_waves = ctx.guided_begin(integ, spp, rank, world)
        for w in range(n_waves):
            ctx.guided_wave(w)
            if collectives and w + 1 < n_waves:
                ptr, count = ctx.guided_stats()
                t = wrap_stats(ptr, count)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                sync()
        film, scale, st = ctx.guided_end(host_film=False) if on_device else ctx.guided_end()
    else:
        film, scale, st = ctx.render(integ, spp, rank, world, host_film=False) if on_device else ctx.render(integ, spp, rank, world)
    if on_device:
        ptr, count = ctx.film_device()
        t = wrap_stats(ptr, count, "<f8")
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        sync()
        if rank == 0:
            film = t.cpu().numpy().reshape(film_buf.shape).copy()
    elif collectives:
        import torch
        film_buf.copy_(torch.from_numpy(np.ascontiguousarray(film)))
        dist.reduce(film_buf, dst=0, op=dist.ReduceOp.SUM)
        sync()
        if rank == 0:
            film = film_buf.cpu().numpy().copy()
    return film, scale, st
