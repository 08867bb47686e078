"""Dense-grid queries of the vector field (marching-cubes input).

``get_set_predictions`` keeps the signature of evaluation/utils/mc_utils.py:88-104 (decoder, samples, max_batch,
device) so the reference's mesh-extraction code can call it unchanged, but: only the 3 vector columns of the last
Linear are computed (``vector_only``: 12.5 % fewer MACs than ``decoder(x)[:, :3]``), host<->device copies go through
pinned buffers on a side stream so chunk k+1 uploads while chunk k computes, and with ``world_size > 1`` the chunks
are dealt round-robin to the ranks (no collective: each rank fills its own rows of the host buffer; the caller
combines them, e.g. ``torch.distributed.all_reduce`` of the zero-initialised CPU buffer over gloo or a file merge).
"""
from __future__ import annotations

import torch


@torch.no_grad()
def get_set_predictions(decoder, samples: torch.Tensor, max_batch: int, device, rank: int = 0,
                        world_size: int = 1) -> torch.Tensor:
    samples.requires_grad = False
    n = samples.shape[0]
    out = torch.zeros_like(samples[:, :3])
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"
    if on_gpu and not samples.is_cuda:
        out = out.pin_memory()
    copy_stream = torch.cuda.Stream(device=dev) if on_gpu else None
    chunks = [(h, min(h + max_batch, n)) for h in range(0, n, max_batch)][rank::world_size]
    pending = None
    for lo, hi in chunks:
        sub = samples[lo:hi, :3].contiguous().float()
        if on_gpu and not sub.is_cuda:
            sub = sub.pin_memory().to(dev, non_blocking=True)
        else:
            sub = sub.to(dev)
        vec = decoder(sub, vector_only=True) if _accepts_vector_only(decoder) else decoder(sub)[:, :3]
        if pending is not None:
            _drain(pending)
        if on_gpu and not out.is_cuda:
            copy_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(copy_stream):
                out[lo:hi].copy_(vec, non_blocking=True)
            vec.record_stream(copy_stream)
            pending = copy_stream
        else:
            out[lo:hi] = vec.to(out.device)
    if pending is not None:
        _drain(pending)
    return out


def _drain(stream) -> None:
    stream.synchronize()


def _accepts_vector_only(decoder) -> bool:
    from .networks import VectorFieldNetwork
    return isinstance(decoder, VectorFieldNetwork)
