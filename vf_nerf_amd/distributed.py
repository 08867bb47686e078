"""One process per GPU over RCCL: ray sharding and the single gradient all-reduce.

Replaces the reference's per-module ``nn.DataParallel`` (models/nerf/vector_field_nerf.py:70-75), which
re-broadcasts weights every forward, scatters points per module call and composites on GPU 0.  Rays are
independent end to end (sampling, window, scan, argmax and sort are per ray; BatchNorm runs in eval mode, Q8),
so each rank renders a contiguous slice of the ray batch with replicated weights and NO data-path collective;
training adds exactly one all-reduce per step over one flat fp32 bucket holding the 805 780 unique parameters'
gradients (3.2 MB — latency-bound on xGMI), issued after backward and before the trainer's clip_grad_norm_ so
that every replica clips identically.  The VF parameters appear twice in ``model.parameters()`` (Q4); the bucket
is built from ``unique_parameters()`` so the alias does not double it.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """-> (rank, world_size, local_rank).  ``backend`` defaults to nccl (= RCCL on ROCm) with a GPU, gloo without."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        use_gpu = torch.cuda.is_available()
        if use_gpu:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"))
    return rank, world, local_rank


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n items for this rank (first n % world ranks get one extra)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def seed_rank_streams(model, rank: int, base_seed: int = 0) -> None:
    """Give this rank its own device random streams: the sampler's Philox stream (``model.rng_seed``) and the supervision
    sampler's (``supervision.manual_seed``).  With the defaults every rank would draw the same jitter and the same supervision
    points for its shard of the batch — correct, but the shards' draws would be copies of one another."""
    from . import supervision
    model.rng_seed = (int(base_seed) << 16) + int(rank)
    model._rng_offset = 0
    supervision.manual_seed(0x5eed + 7919 * (int(rank) + 1) + int(base_seed))


def shard_rays(pose: torch.Tensor, pixels: torch.Tensor, intrinsics: torch.Tensor, rank: int, world: int):
    lo, hi = shard_bounds(pixels.shape[0], rank, world)
    return pose[lo:hi], pixels[lo:hi], intrinsics[lo:hi]


class GradientBucket:
    """Flat fp32 gradient storage for the model's unique parameters; every ``param.grad`` is a view into it, so
    backward accumulates straight into the bucket and the all-reduce needs no packing copies."""

    def __init__(self, model) -> None:
        # the facade's optimizer (optim.FlatAdam) already keeps every gradient in one flat buffer: the bucket IS that buffer
        opt = getattr(model, "optimizer", None)
        f = opt.flat() if hasattr(opt, "flat") else None
        if f is not None and all(p.requires_grad for p, _, _, _ in f["entries"]):
            self.params: List[torch.nn.Parameter] = [p for p, _, _, _ in f["entries"]]
            self.flat = f["grad"]
            opt._rebind_grads(f)
            return
        self.params = [p for p in model.unique_parameters() if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def numel(self) -> int:
        return self.flat.numel()

    def zero(self) -> None:
        """Use instead of optimizer.zero_grad(set_to_none=True), which would detach the views."""
        self.flat.zero_()

    def rebind(self) -> None:
        """Re-attach the views if something replaced ``param.grad`` (e.g. zero_grad(set_to_none=True))."""
        off = 0
        for p in self.params:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
                p.grad = view
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
                p.grad = view
            off += n

    def all_reduce_mean(self, group=None, async_op: bool = False):
        """One collective per step: sum over ranks, then divide by the world size."""
        if not dist.is_initialized() or dist.get_world_size(group) == 1:
            return None
        self.rebind()
        world = dist.get_world_size(group)
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            return work, world
        self.flat.div_(world)
        return None


def broadcast_parameters(model, src: int = 0, group=None) -> None:
    """Make every replica start from rank ``src``'s weights (parameters and BatchNorm running statistics)."""
    if not dist.is_initialized():
        return
    for mod in (model.vector_field_network, model.rendering_network, model.density):
        for t in list(mod.parameters()) + list(mod.buffers()):
            dist.broadcast(t.data, src=src, group=group)


def gather_rows(t: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """Concatenate per-rank row shards produced with ``shard_bounds`` (inference outputs; optional)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    max_rows = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
