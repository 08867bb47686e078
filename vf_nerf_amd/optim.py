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


# Which FlatAdam owns a parameter: kept OUTSIDE the tensor (an attribute on the Parameter would land in its __dict__ and make the
# Parameter, and every module holding it, unpicklable: torch.save(module), spawn arguments).  id(p) -> (weakref to p, weakref to the
# optimizer); an entry whose parameter has died (its id may be reused) is recognised by the identity check and dropped.
_OWNERS = {}

# dropin.install(patch_clip=False): torch.nn.utils.clip_grad_norm_ stays PyTorch's own function.  A step session's backward then leaves
# every ``param.grad`` None (the gradient sits in the flat buffer only), so the trainer's clip call (train/vector_field_nerf_train.py:
# 254-255) finds nothing to scale, and FlatAdam.step() — the next call the trainer makes — all-reduces (more than one rank), clips with
# the model's configured clip_norm (the value the trainer passes: config.vf_nerf_config.scheduler_config.clip_norm) and updates: the same
# launches in the same order as the wrapped clip_grad_norm_ followed by step().
CLIP_INSIDE_STEP = False


def _register_owner(p: torch.nn.Parameter, opt) -> None:
    import weakref
    _OWNERS[id(p)] = (weakref.ref(p), weakref.ref(opt))
    if len(_OWNERS) > 4096:                       # forget entries of parameters that no longer exist
        for k in [k for k, (rp, ro) in _OWNERS.items() if rp() is None or ro() is None]:
            del _OWNERS[k]


def owner_of(p: torch.nn.Parameter):
    """The FlatAdam that holds ``p`` in its flat buffers, or None."""
    hit = _OWNERS.get(id(p))
    if hit is None:
        return None
    rp, ro = hit
    return ro() if rp() is p else None


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


class FlatAdam(SequentialAdam):
    """The same optimizer with everything in ONE flat fp32 buffer on the device (csrc/vfn_adam.hip): the unique parameters
    are re-pointed to views of ``flat_param`` (sorted by multiplicity: the parameters listed twice first), their gradients
    to views of ``flat_grad``, the Adam moments to views of ``flat_exp_avg`` / ``flat_exp_avg_sq``; ``step()`` is one launch,
    ``zero_grad()`` one memset, and ``clip_grad_norm_`` (below) two launches instead of ~15.  ``state`` / ``state_dict()`` /
    ``load_state_dict()`` keep torch.optim.Adam's layout (the per-parameter moments are the views), so checkpoints written by
    either load in the other.  Parameters on the CPU fall back to SequentialAdam's path.

    The update runs through raw device pointers, which does not advance ``tensor._version``; ``after_step`` (a callable) lets the
    owner invalidate what it caches on parameter versions (the facade's weight packs)."""

    def __init__(self, params, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0) -> None:
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._flat = None
        self.after_step = None
        self.step_engine = None            # weakref to the owner's stepengine.StepEngine (set by it), or None
        for p in self.param_groups[0]["params"]:           # lets clip_grad_norm_ find the flat gradient buffer of a parameter list
            _register_owner(p, self)

    # -- flat storage -----------------------------------------------------------------------------
    def _layout(self):
        """unique parameters sorted by multiplicity (descending, stable) -> [(param, offset, numel, mult)], regions"""
        plist = self.param_groups[0]["params"]
        count, order = {}, []
        for p in plist:
            if id(p) not in count:
                count[id(p)] = 0
                order.append(p)
            count[id(p)] += 1
        order.sort(key=lambda p: -count[id(p)])
        entries, regions, off = [], [], 0
        for p in order:
            m = count[id(p)]
            if m > 2:
                raise NotImplementedError("a parameter listed more than twice")
            if regions and regions[-1][2] == m:
                regions[-1][1] = off + p.numel()
            else:
                regions.append([off, off + p.numel(), m])
            entries.append((p, off, p.numel(), m))
            off += p.numel()
        return entries, [tuple(r) for r in regions], off

    def _bound(self) -> bool:
        f = self._flat
        if f is None:
            return False
        pairs = f.get("bound_pairs")
        if pairs is None:
            base = f["param"].data_ptr()
            pairs = f["bound_pairs"] = [(p, base + 4 * off) for p, off, _, _ in f["entries"]]
        for p, want in pairs:
            if p.data_ptr() != want or not p.requires_grad:
                return False
        return True

    def flat(self):
        """Build (or re-build, when something replaced a parameter's storage: .to(), .cuda()) the flat buffers; returns the
        dict {param, grad, exp_avg, exp_avg_sq, entries, regions} or None when the parameters are not on one CUDA device."""
        if len(self.param_groups) != 1:
            return None
        if self._bound():
            return self._flat
        entries, regions, total = self._layout()
        if any(not p.requires_grad for p, _, _, _ in entries):
            # A frozen parameter (requires_grad_(False)) has no gradient: torch.optim.Adam skips it — no update, no step count, no
            # weight decay.  The flat kernels update every element of the buffer, so such a list takes SequentialAdam's per-parameter
            # path (which skips ``grad is None``); a gradient view this optimizer attached earlier must not pose as a gradient.
            old = self._flat
            if old is not None:
                views = {id(v) for v in old.get("grad_views", [])}
                for p, _, _, _ in old["entries"]:
                    if not p.requires_grad and (id(p.grad) in views or (p.grad is not None and p.grad.data_ptr() >= old["grad"].data_ptr() and
                                                                          p.grad.data_ptr() < old["grad"].data_ptr() + 4 * old["grad"].numel())):
                        p.grad = None
            self._flat = None
            return None
        devs = {p.device for p, _, _, _ in entries}
        if len(devs) != 1 or next(iter(devs)).type != "cuda" or any(p.dtype != torch.float32 for p, _, _, _ in entries) or len(regions) > 4:
            self._flat = None
            return None
        dev = next(iter(devs))
        # One bias correction per region and update (step()): sound only while every parameter of a region has taken the same number
        # of updates.  A parameter that sat out some steps frozen (SequentialAdam's path skipped it) has a smaller count than its
        # neighbours: such a list stays on the per-parameter path, whose corrections are per parameter like torch.optim.Adam's.
        for m in {m for _, _, _, m in entries}:
            counts = {float(self.state[p]["step"]) if len(self.state.get(p, {})) else 0.0 for p, _, _, mm in entries if mm == m}
            if len(counts) > 1:
                self._flat = None
                return None
        from . import lib
        old = self._flat
        f = dict(param=torch.empty(total, device=dev), grad=torch.zeros(total, device=dev), exp_avg=torch.zeros(total, device=dev),
                 exp_avg_sq=torch.zeros(total, device=dev), entries=entries, regions=regions, out2=torch.zeros(2, device=dev),
                 workspace=lib.flat_clip_workspace(dev), steps=torch.zeros(len(entries), dtype=torch.float32))
        with torch.no_grad():
            for i, (p, off, n, m) in enumerate(entries):
                f["param"][off:off + n].copy_(p.detach().reshape(-1))
                if p.grad is not None:
                    f["grad"][off:off + n].copy_(p.grad.detach().reshape(-1))
                st = self.state.get(p, {})
                if len(st):                      # moments that exist already (a loaded checkpoint, or a re-build after .to())
                    f["exp_avg"][off:off + n].copy_(st["exp_avg"].detach().reshape(-1).to(dev))
                    f["exp_avg_sq"][off:off + n].copy_(st["exp_avg_sq"].detach().reshape(-1).to(dev))
                    f["steps"][i] = float(st["step"])
                p.data = f["param"][off:off + n].view(p.shape)
                p.grad = f["grad"][off:off + n].view(p.shape)
                self.state[p] = {"step": f["steps"][i], "exp_avg": f["exp_avg"][off:off + n].view(p.shape),
                                 "exp_avg_sq": f["exp_avg_sq"][off:off + n].view(p.shape)}
        del old
        self._flat = f
        return f

    def _grad_views(self, f):
        views = f.get("grad_views")
        if views is None:
            views = f["grad_views"] = [f["grad"][off:off + n].view(p.shape) for p, off, n, _ in f["entries"]]
        return views

    def _rebind_grads(self, f) -> None:
        """Every ``param.grad`` is (again) its view of the flat gradient buffer; a gradient that autograd or the caller put
        somewhere else is copied in.  The views are made once: the common case is fifty identity checks."""
        parked = "parked_max_norm" in f           # CLIP_INSIDE_STEP: a None gradient is the PARKED one (the buffer holds it), not a zero one
        for (p, off, n, _), view in zip(f["entries"], self._grad_views(f)):
            g = p.grad
            if g is view:
                continue
            if g is None:
                if not parked:
                    view.zero_()              # (None IS a zero gradient: the view must not bring back what an earlier step left in the buffer)
                p.grad = view
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
                p.grad = view
            else:
                p.grad = view

    # -- torch.optim.Optimizer interface ------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True) -> None:
        f = self.flat()
        if f is None:
            return super().zero_grad(set_to_none=set_to_none)
        f["grad"].zero_()
        f.pop("parked_max_norm", None)
        for (p, off, n, _), view in zip(f["entries"], self._grad_views(f)):     # (keeps the views: setting .grad to None would detach them from the buffer)
            if p.grad is not view:
                p.grad = view

    def step_scalars(self, f):
        """(beta1, beta2, step_size[2 per region], bc2_sqrt[2 per region]) of the NEXT update, evaluated on the host in double
        precision as torch.optim.Adam does — what vfn_flat_adam_step (and vfn_train_step) take."""
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            raise NotImplementedError("FlatAdam covers the reference's plain Adam configuration")
        beta1, beta2 = group["betas"]
        lr = float(group["lr"])
        step_size, bc2_sqrt = [], []
        first_of_region = {}
        for i, (p, off, n, m) in enumerate(f["entries"]):
            first_of_region.setdefault(m, i)
        for (start, end, m) in f["regions"]:
            t0 = float(f["steps"][first_of_region[m]])      # every parameter of a region has taken the same number of updates (flat() checks)
            for k in range(2):
                t = t0 + k + 1
                step_size.append(lr / (1 - beta1 ** t))
                bc2_sqrt.append((1 - beta2 ** t) ** 0.5)
        return beta1, beta2, step_size, bc2_sqrt

    def finish_step(self, f) -> None:
        """Bookkeeping after the update kernels ran (here or inside vfn_train_step): step counters, the owner's pack invalidation."""
        inc = f.get("step_increments")
        if inc is None:
            inc = f["step_increments"] = torch.tensor([float(m) for _, _, _, m in f["entries"]])
        f["steps"] += inc
        if self.after_step is not None:
            self.after_step()

    @torch.no_grad()
    def step(self, closure=None):
        f = self.flat()
        if f is None:
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import lib
        parked = f.pop("parked_max_norm", None)
        if parked is not None:
            # CLIP_INSIDE_STEP: the session's backward parked the gradient in the flat buffer with every param.grad None — here a parked
            # gradient, NOT the zero gradient _rebind_grads takes a None for.  All-reduce | clip | update, as the wrapped clip + step() do.
            for (p, off, n, _), view in zip(f["entries"], self._grad_views(f)):
                if p.grad is None:
                    p.grad = view
                elif p.grad is not view:              # the caller put a gradient of its own there after the backward: it joins the parked one
                    view.add_(p.grad)
                    p.grad = view
            from . import dropin
            if dropin.data_parallel:
                dropin.all_reduce_flat(f)
            lib.flat_clip_grad_norm(f["grad"], f["regions"], float(parked), f["workspace"], f["out2"])
        else:
            self._rebind_grads(f)
            if CLIP_INSIDE_STEP:
                from . import dropin
                if dropin.data_parallel and dropin.world_size() > 1:
                    raise RuntimeError("dropin.install(patch_clip=False) with more than one rank: the gradient all-reduce rides on the step session's "
                                       "optimizer.step(); this step did not go through a session (stepengine.StepEngine.of(model).why_not) — "
                                       "use install(patch_clip=True) or all-reduce model.optimizer's flat gradient yourself")
        # the owning model's step engine (stepengine.py) applies the update and re-packs the weight packs in one C call when its structs
        # describe these buffers (after a training render of the shipped regime they do)
        eng = self.step_engine() if self.step_engine is not None else None
        if eng is not None and eng.adam_step(f):
            return loss
        group = self.param_groups[0]
        beta1, beta2, step_size, bc2_sqrt = self.step_scalars(f)
        lib.flat_adam_step(f["param"], f["grad"], f["exp_avg"], f["exp_avg_sq"], f["regions"], step_size, bc2_sqrt, beta1, beta2,
                           group["eps"], group["weight_decay"])
        self.finish_step(f)
        return loss

    def load_state_dict(self, state_dict) -> None:
        super().load_state_dict(state_dict)          # per-parameter tensors (clones on the parameters' device)
        self._flat = None                            # re-built, with these moments, on next use

    def regions_for(self, plist) -> bool:
        """True when ``plist`` names exactly this optimizer's parameters with the same multiplicities (the fused clip applies)."""
        f = self.flat()
        if f is None:
            return False
        ids = tuple(map(id, plist))
        if f.get("regions_ok_ids") == ids:          # (the trainer passes the same list every step: one tuple comparison)
            return True
        count = {}
        for i in ids:
            count[i] = count.get(i, 0) + 1
        ok = len(count) == len(f["entries"]) and all(count.get(id(p), 0) == m for p, _, _, m in f["entries"])
        if ok:
            f["regions_ok_ids"] = ids
        return ok


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm: float, norm_type: float = 2.0) -> torch.Tensor:
    """``torch.nn.utils.clip_grad_norm_(parameters, max_norm, foreach=False)`` for a list that may name a parameter more
    than once: the total norm counts such a gradient once per occurrence and the gradient is scaled once per occurrence
    (train/vector_field_nerf_train.py:254-255 over models/nerf/vector_field_nerf.py:57-63).  Multi-tensor kernels over
    distinct tensors per pass."""
    if norm_type != 2.0:
        raise NotImplementedError("only the 2-norm (the reference's default) is implemented")
    plist = [p for p in (parameters if not isinstance(parameters, torch.Tensor) else [parameters])]
    owner = owner_of(plist[0]) if plist else None
    if owner is not None and owner.regions_for(plist):           # the gradients are one flat buffer: two launches
        from . import lib
        f = owner.flat()
        owner._rebind_grads(f)
        lib.flat_clip_grad_norm(f["grad"], f["regions"], float(max_norm), f["workspace"], f["out2"])
        return f["out2"][0].clone()
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
