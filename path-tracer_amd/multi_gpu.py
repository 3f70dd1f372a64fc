"""Sharded render driver: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).

The reference is a single-process CPU renderer (tiles over rayon, integrator/mod.rs:34-46); this is how its per-sample
loop is dealt over N GPUs without changing any sample:
  * path / lightpath: rank r renders sample indices i = r, r + N, ... of every pixel (ptgpu_render_shard);
  * guided: in every wave, tile t of pass k belongs to rank (t + k) % N (pt_guided.h: k_raygen_guided); after each wave
    but the last the collector statistics are summed IN PLACE on the library's buffer with one all_reduce, then every
    rank refines the same tree;
  * the film sums are reduced onto rank 0 once at the end.
"""
from __future__ import annotations

import numpy as np


def samples_of_shard(spp: int, rank: int, world: int) -> range:
    """Sample indices rank `rank` renders for every pixel (ptgpu.hip: render_path_like)."""
    return range(rank, spp, world)


def shard_of_pair(tile: int, pass_index: int, world: int) -> int:
    """Owner of (pass, tile) in a guided wave (pt_guided.h: k_raygen_guided)."""
    return (tile + pass_index) % world


class DeviceArray:
    """Raw device pointer -> torch tensor (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def device_stats_tensor(torch, dev):
    """wrap(ptr, count) for a libptgpu context: a float32 tensor aliasing the library's statistics buffer."""
    return lambda ptr, count: torch.as_tensor(DeviceArray(ptr, count), device=dev)


def render_sharded(ctx, integ, spp, rank, world, guided, dist=None, wrap_stats=None, film_buf=None, sync=lambda: None, collectives=None):
    """One render of the whole job on this rank's shard. Returns (film sums [complete on rank 0], film_scale, stats).

    ctx       : object with render(integ, spp, rank, world) and guided_{begin,wave,stats,end} (path_tracer_amd.Context)
    dist      : torch.distributed (initialised) when world > 1
    wrap_stats: (ptr_or_buffer, count) -> tensor aliasing the statistics buffer
    film_buf  : preallocated float64 tensor (H, W, 3) on the collective's device
    sync      : waits for the collective's stream (torch.cuda.synchronize on GPU)
    collectives: run the collectives even when world == 1 (rehearsal of the RCCL path on a one-GPU box); default world > 1
    """
    if collectives is None:
        collectives = world > 1
    if guided:
        n_waves = ctx.guided_begin(integ, spp, rank, world)
        for w in range(n_waves):
            ctx.guided_wave(w)
            if collectives and w + 1 < n_waves:
                ptr, count = ctx.guided_stats()
                t = wrap_stats(ptr, count)
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                sync()
        film, scale, st = ctx.guided_end()
    else:
        film, scale, st = ctx.render(integ, spp, rank, world)
    if collectives:
        import torch
        film_buf.copy_(torch.from_numpy(np.ascontiguousarray(film)))
        dist.reduce(film_buf, dst=0, op=dist.ReduceOp.SUM)
        sync()
        if rank == 0:
            film = film_buf.cpu().numpy().copy()
    return film, scale, st
