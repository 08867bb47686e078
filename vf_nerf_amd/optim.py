"""Optimizer-side glue of the training step: the reference's Adam and gradient clipping over a parameter list that
names the vector-field parameters twice (models/nerf/vector_field_nerf.py:57-63), without the ~1000
per-parameter kernel launches of the sequential PyTorch loops.

The reference semantics are those of ``torch.optim.Adam`` / ``torch.nn.utils.clip_grad_norm_`` iterating the list one
entry after the other (the PyTorch 1.x it was written for; ``foreach=False`` today): a parameter listed twice is
clipped twice and receives two consecutive Adam updates per ``step()`` from the same gradient, with its step counter
advancing by two.  PyTorch's multi-tensor ("foreach") kernels are only unsafe when the SAME tensor occurs twice inside
one call; so here every pass runs them on lists of distinct tensors — pass 1 over the unique parameters, pass 2 over
the ones listed twice — which performs, element for element, the same floating-point operations in the same order as
the sequential loop (``tests/test_hip_backward.py::test_one_adam_step_matches_oracle_step``).
"""
from __future__ import annotations

from typing import Iterable, List, Tuple

import torch
from torch.optim.adam import adam as _functional_adam


def _passes(params: Iterable[torch.nn.Parameter]) -> List[List[torch.nn.Parameter]]:
    """[params seen >= 1 times, params seen >= 2 times, ...] in first-occurrence order."""
    count, order = {}, []
    for p in params:
        if id(p) not in count:
            count[id(p)] = 0
            order.append(p)
        count[id(p)] += 1
    depth = max(count.values(), default=0)
    return [[p for p in order if count[id(p)] > k] for k in range(depth)]


class SequentialAdam(torch.optim.Adam):
    """``torch.optim.Adam(params, foreach=False)`` — same state, same ``state_dict`` — whose ``step`` runs the
    multi-tensor kernels once per multiplicity pass instead of once per list entry."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0) -> None:
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
                raise NotImplementedError("SequentialAdam covers the reference's plain Adam configuration")
            beta1, beta2 = group["betas"]
            for plist in _passes(group["params"]):
                params, grads, exp_avgs, exp_avg_sqs, steps = [], [], [], [], []
                for p in plist:
                    if p.grad is None:
                        continue
                    if p.grad.is_sparse:
                        raise RuntimeError("Adam does not support sparse gradients")
                    state = self.state[p]
                    if len(state) == 0:      # same lazy initialisation as torch.optim.Adam._init_group
                        state["step"] = torch.tensor(0.0, dtype=torch.float32)
                        state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                        state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    params.append(p)
                    grads.append(p.grad)
                    exp_avgs.append(state["exp_avg"])
                    exp_avg_sqs.append(state["exp_avg_sq"])
                    steps.append(state["step"])
                if not params:
                    continue
                use_foreach = all(p.is_cuda for p in params)
                _functional_adam(params, grads, exp_avgs, exp_avg_sqs, [], steps, foreach=use_foreach, capturable=False,
                                 differentiable=False, fused=False, grad_scale=None, found_inf=None, has_complex=False,
                                 amsgrad=False, beta1=beta1, beta2=beta2, lr=group["lr"], weight_decay=group["weight_decay"],
                                 eps=group["eps"], maximize=False)
        return loss


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm: float, norm_type: float = 2.0) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_(parameters, max_norm, foreach=False)`` for a list that may name a parameter more
    than once: the total norm counts such a gradient once per occurrence and the gradient is scaled once per occurrence
    (train/vector_field_nerf_train.py:254-255 over models/nerf/vector_field_nerf.py:57-63).  Multi-tensor kernels over
    distinct tensors per pass."""
    if norm_type != 2.0:
        raise NotImplementedError("only the 2-norm (the reference's default) is implemented")
    plist = [p for p in (parameters if not isinstance(parameters, torch.Tensor) else [parameters])]
    passes = [[p for p in ps if p.grad is not None] for ps in _passes(plist)]
    passes = [ps for ps in passes if ps]
    if not passes:
        return torch.tensor(0.0)
    dev = passes[0][0].grad.device
    # one norm per unique gradient; the stacked vector lists each one as often as the parameter occurs (same values the
    # sequential function stacks, in a different order: the 2-norm of the stack is order-independent up to rounding)
    uniq = passes[0]
    norms = torch._foreach_norm([p.grad for p in uniq], 2.0) if dev.type == "cuda" else [torch.linalg.vector_norm(p.grad, 2.0) for p in uniq]
    by_id = {id(p): n for p, n in zip(uniq, norms)}
    stacked = torch.stack([by_id[id(p)].to(dev) for ps in passes for p in ps])
    total_norm = torch.linalg.vector_norm(stacked, 2.0)
    clip_coef = torch.clamp(max_norm / (total_norm + 1e-6), max=1.0)
    for ps in passes:
        grads = [p.grad for p in ps]
        if dev.type == "cuda":
            torch._foreach_mul_(grads, clip_coef)
        else:
            for g in grads:
                g.mul_(clip_coef)
    return total_norm
