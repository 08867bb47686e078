"""``VFLoss`` with the interface of the reference's ``models/losses/vf_loss.py:13-87``: same constructor
arguments (a config with ``depth_loss_clamp`` / ``norm_smaller_than_one_start`` / ``directional_derivatives_start`` and a
weights record), same ``forward(pred, gt, epoch) -> (loss, {name: float})``, same terms.  The reference reads its six log
scalars back with six ``.item()`` calls — six device synchronisations in the MIDDLE of every training step (between the loss
and ``backward()``, with the device idle while the host catches up); here they are stacked on the device, copied to pinned
host memory asynchronously, and the returned dict fetches them the first time a value is READ (``_LazyTerms``) — in the
reference trainer that is after ``optimizer.step()`` has been queued (train/vector_field_nerf_train.py:262-275), and a loop
that never looks at the terms never synchronises.  ``vf_nerf_amd.dropin`` installs it as ``models.losses.vf_loss.VFLoss``.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import nn

_NAMES = ("rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss",
          "directional_derivatives_loss")


DEFERRED_SCALARS = True        # loss.item() and losses_dict[key] as deferred scalars (deferred.py): the reference loop's running sums without a sync


class _LazyTerms(dict):
    """{name: float} whose values arrive from the device on first read.  Keys, length and iteration order are available at
    once.  Indexing a value that has not arrived gives a ``deferred.DeferredScalar`` (a number that stays on the device while it is only
    added to other such numbers — the reference trainer's running sums, train.py:262-275 — and is the float on any other use); every
    other way of reading (get, values, items, pop, copy, repr, comparison) waits for the device and holds plain floats from then on."""

    def __init__(self, names, stacked: torch.Tensor, holder=None, first: int = 0) -> None:
        super().__init__((n, None) for n in names)
        self._pending = None
        if stacked is not None and not stacked.is_cuda:
            dict.update(self, zip(names, stacked.tolist()))
            return
        from .deferred import DeviceScalars
        # ``holder``: a DeviceScalars whose entries first .. first + len(names) - 1 are these terms (the loss's total sits in the same vector)
        self._pending = (holder if holder is not None else DeviceScalars(stacked.detach().float()), tuple(names), int(first))

    def _fetch(self) -> None:
        if self._pending is not None:
            holder, names, first = self._pending
            self._pending = None
            host = holder.values()
            for j, n in enumerate(names):
                if dict.__getitem__(self, n) is None:          # (a value the caller has overwritten meanwhile stays)
                    dict.__setitem__(self, n, host[first + j])

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None and self._pending is not None:
            if DEFERRED_SCALARS and k in self._pending[1]:
                from .deferred import DeferredScalar
                return DeferredScalar(self._pending[0], self._pending[2] + self._pending[1].index(k))
            self._fetch()
            return dict.__getitem__(self, k)
        return v

    def __iter__(self):          # (an overridden __iter__ also keeps dict(terms) / {**terms} off CPython's raw-copy fast path)
        return dict.__iter__(self)

    def get(self, k, default=None):
        self._fetch()
        return dict.get(self, k, default)

    def values(self):
        self._fetch()
        return dict.values(self)

    def items(self):
        self._fetch()
        return dict.items(self)

    def pop(self, *a):
        self._fetch()
        return dict.pop(self, *a)

    def copy(self):
        self._fetch()
        return dict(dict.items(self))

    def __repr__(self) -> str:
        self._fetch()
        return dict.__repr__(self)

    def __eq__(self, other) -> bool:
        self._fetch()
        return dict.__eq__(self, other)

    def __ne__(self, other) -> bool:
        return not self.__eq__(other)

    def __reduce__(self):
        self._fetch()
        return (dict, (dict(dict.items(self)),))


class _FusedLoss(torch.autograd.Function):
    """The five tensor terms of VFLoss in csrc/vfn_loss.hip: forward = one reduction launch + a one-workgroup finish, backward = one
    elementwise launch (instead of ~40 tensor-op launches over [N*S_t, 3] operands).  Inputs after the fixed ones: the supervision
    segments as pred0, gt0, pred1, gt1, ... (at most three pairs)."""

    @staticmethod
    def forward(ctx, lp, dn_buffer, rgb, rgb_gt, depth, depth_gt, normals, points, *segs):
        from . import lib
        ws = lib.vf_loss_workspace(rgb.device)
        preds, gts = list(segs[0::2]), list(segs[1::2])
        out = lib.vf_loss_fwd(lp, rgb, rgb_gt, depth, depth_gt, normals, points, preds, gts, ws)
        ctx.lp, ctx.ws, ctx.n_seg = lp, ws, len(preds)
        # (an open training step lends the rows its backward reads the normals' gradient from: written in place, no copy; stepengine.py)
        ctx.dn_buffer = dn_buffer
        ctx.save_for_backward(rgb, rgb_gt, depth, depth_gt, normals, points, *segs)
        ctx.mark_non_differentiable(out)
        return out[6].clone(), out

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        from . import lib
        rgb, rgb_gt, depth, depth_gt, normals, points, *segs = ctx.saved_tensors
        preds, gts = list(segs[0::2]), list(segs[1::2])
        need = ctx.needs_input_grad            # (lp, dn_buffer, rgb, rgb_gt, depth, depth_gt, normals, points, pred0, gt0, ...)
        d_rgb = torch.empty_like(rgb) if need[2] else None
        d_depth = torch.empty_like(depth) if (need[4] and depth is not None) else None
        d_normals = None
        if need[6]:
            buf = ctx.dn_buffer
            d_normals = buf if (buf is not None and buf.shape == normals.shape and buf.device == normals.device) else torch.empty_like(normals)
        d_sup = [torch.empty_like(p) if need[8 + 2 * k] else None for k, p in enumerate(preds)]
        lib.vf_loss_bwd(ctx.lp, rgb, rgb_gt, depth, depth_gt, normals, points, preds, gts, ctx.ws, g_total.reshape(1).float().contiguous(),
                        d_rgb, d_depth, d_normals, d_sup)
        grads = [None, None, d_rgb, None, d_depth, None, d_normals, None]
        for k in range(ctx.n_seg):
            grads += [d_sup[k], None]
        return tuple(grads)


class VFLoss(nn.Module):
    def __init__(self, config, weights) -> None:
        super().__init__()
        self.config = config
        self.weights = weights
        # device tensors: the terms come from the fused kernels of csrc/vfn_loss.hip (False: the tensor-op formulation below, which
        # is also what CPU tensors get)
        self.fused = True

    def _fused_forward(self, pred, gt, epoch: int):
        from . import lib
        w, rgb = self.weights, pred["rgb"]

        def f32(t):
            return t.float().contiguous()

        normals = f32(pred["normals"].reshape(-1, 3))
        segments = pred.get("supervised_segments")
        if segments is None:
            segments = [(pred["supervised_normals"], gt["supervised_normals"])] if pred["supervised_normals"].nelement() > 0 else []
        segments = [(f32(a.reshape(-1, 3)), f32(b.reshape(-1, 3))) for a, b in segments if a.nelement() > 0]
        if len(segments) > 3:
            segments = segments[:2] + [(torch.cat([a for a, _ in segments[2:]]), torch.cat([b for _, b in segments[2:]]))]
        lp = lib.LossParams()
        lp.n_rays, lp.n_normals = rgb.shape[0], normals.shape[0]
        for k, (a, _) in enumerate(segments):
            lp.n_sup[k] = a.shape[0]
        has_depth = gt["depth"].nelement() > 0
        lp.has_depth, lp.smaller_on = int(has_depth), int(epoch >= self.config.norm_smaller_than_one_start)
        lp.w_rgb, lp.w_depth, lp.w_unit, lp.w_sup, lp.w_smaller = float(w.rgb), float(w.depth), float(w.unit_norm), float(w.supervision), \
            float(w.norm_smaller_than_one)
        lp.depth_clamp = float(self.config.depth_loss_clamp)
        points = None
        rc = pred.get("ray_center")                # (points[N,S,3] | [M,3], centroid as three Python floats, radius): trainer.TrainStep
        dn_buffer = None
        if normals.requires_grad:
            from .stepengine import current_session, find_marker
            session = current_session()
            if session is not None and normals.data_ptr() == session.normals.data_ptr() and normals.numel() == session.m * 3:
                dn_buffer = session.dn
            # the centre-ball rows supervision.get_center_indices_and_gt deferred to this loss (the reference trainer's call sequence)
            marked = find_marker(pred["supervised_normals"]) if rc is None else None
            if marked is not None:
                info = marked.ray_centre
                if marked is not session or dn_buffer is None or info is None:
                    raise ValueError("VFLoss: the deferred centre-ball rows belong to another render() than pred['normals']")
                rc = (marked.points, info["centroid"], info["radius"])
                info["consumed"] = True
        if rc is not None:
            points = f32(rc[0].reshape(-1, 3))
            lp.ray_center, lp.radius = 1, float(rc[2])
            for i in range(3):
                lp.centroid[i] = float(rc[1][i])
        flat = [t for pair in segments for t in pair]
        total, out = _FusedLoss.apply(lp, dn_buffer, f32(rgb.reshape(-1, 3)), f32(gt["rgb"].reshape(-1, 3)),
                                      f32(pred["depth"].reshape(-1)) if has_depth else None,
                                      f32(gt["depth"].reshape(-1)) if has_depth else None, normals, points, *flat)
        terms = out[:6]
        dd = pred.get("directional_derivatives")
        if dd is not None and epoch >= self.config.directional_derivatives_start:
            dd_loss = dd.mean()
            total = total + w.directional_derivatives * dd_loss
            terms = terms.clone()
            terms[5] = dd_loss.detach()
            return total, _LazyTerms(_NAMES, terms)
        # the total (entry 6) and the six terms are ONE device vector: loss.item() and losses_dict[key] of a step add to the trainer's
        # running sums as one vector addition, with no synchronisation (deferred.py)
        from .deferred import DeviceScalars, as_loss
        holder = DeviceScalars(out)
        return (as_loss(total, holder, 6) if DEFERRED_SCALARS else total), _LazyTerms(_NAMES, None, holder=holder)

    @staticmethod
    def _fused_shapes_ok(pred, gt) -> Optional[str]:
        """The fused kernels read raw pointers with sizes taken from pred["rgb"] / pred["normals"] alone: every other operand must
        have exactly the matching extent (the tensor-op formulation below would raise or broadcast on a mismatch; a kernel would
        read out of bounds).  Returns what does not match, or None."""
        n = pred["rgb"].reshape(-1, 3).shape[0] if pred["rgb"].numel() % 3 == 0 else -1
        if n < 0 or gt["rgb"].numel() != 3 * n:
            return f"rgb {tuple(pred['rgb'].shape)} against ground truth {tuple(gt['rgb'].shape)}"
        if gt["depth"].nelement() > 0 and (pred["depth"] is None or pred["depth"].numel() != n or gt["depth"].numel() != n):
            return f"depth {None if pred['depth'] is None else tuple(pred['depth'].shape)} / ground truth {tuple(gt['depth'].shape)} for {n} rays"
        if pred["normals"].numel() % 3:
            return f"normals {tuple(pred['normals'].shape)}"
        rc = pred.get("ray_center")
        if rc is not None and rc[0].numel() != pred["normals"].numel():
            return f"ray_center points {tuple(rc[0].shape)} against normals {tuple(pred['normals'].shape)}"
        segments = pred.get("supervised_segments")
        if segments is None:
            segments = [(pred["supervised_normals"], gt["supervised_normals"])] if pred["supervised_normals"].nelement() > 0 else []
        for k, (a, b) in enumerate(segments):
            if a.numel() % 3 or a.numel() != b.numel():
                return f"supervised segment {k}: prediction {tuple(a.shape)} against ground truth {tuple(b.shape)}"
        return None

    def forward(self, pred: Dict[str, torch.Tensor], gt: Dict[str, torch.Tensor], epoch: int
                ) -> Tuple[torch.Tensor, Dict[str, float]]:
        rgb = pred["rgb"]
        if self.fused and rgb.is_cuda and rgb.dtype == torch.float32 and pred["normals"].is_cuda:
            bad = self._fused_shapes_ok(pred, gt)
            if bad is None:
                return self._fused_forward(pred, gt, epoch)
            if pred.get("supervised_segments") is not None or pred.get("ray_center") is not None:
                raise ValueError("VFLoss: operand sizes do not match (" + bad + ")")
            # plain reference-style operands of unequal extent: the tensor-op formulation decides (it raises or broadcasts as torch does)
        if pred.get("supervised_segments") is not None or pred.get("ray_center") is not None:
            raise ValueError("supervised_segments / ray_center are inputs of the fused device loss (CUDA tensors, VFLoss.fused = True)")
        zero = torch.zeros((), device=rgb.device, dtype=rgb.dtype)
        w = self.weights
        rgb_loss = (rgb - gt["rgb"]).abs().mean()
        depth_loss = zero
        if gt["depth"].nelement() > 0:
            depth_loss = (pred["depth"] - gt["depth"]).abs().clamp(max=self.config.depth_loss_clamp).mean()
        norms = torch.linalg.vector_norm(pred["normals"], dim=1)
        unit_norm_loss = ((norms - 1) ** 2).mean()
        supervision_loss = zero
        if pred["supervised_normals"].nelement() > 0:
            rows = pred.get("supervised_rows")
            if rows is None:
                supervision_loss = ((pred["supervised_normals"] - gt["supervised_normals"]) ** 2).mean()
            else:
                # extension (trainer.TrainStep): some rows are placeholders with prediction = ground truth = 0; the mean runs over
                # the `rows` real ones (a device scalar), so no row count has to travel to the host
                supervision_loss = ((pred["supervised_normals"] - gt["supervised_normals"]) ** 2).sum() / (3.0 * rows)
        smaller_loss = zero
        if epoch >= self.config.norm_smaller_than_one_start:
            smaller_loss = (torch.relu(norms - 1) ** 2).mean()
        dd_loss = zero
        dd = pred.get("directional_derivatives")
        if dd is not None and epoch >= self.config.directional_derivatives_start:
            dd_loss = dd.mean()
        terms = (rgb_loss, depth_loss, unit_norm_loss, supervision_loss, smaller_loss, dd_loss)
        loss = w.rgb * rgb_loss + w.depth * depth_loss + w.unit_norm * unit_norm_loss + w.supervision * supervision_loss + \
            w.norm_smaller_than_one * smaller_loss + w.directional_derivatives * dd_loss
        # one asynchronous read-back instead of six synchronising ones; the values are waited for when first read
        return loss, _LazyTerms(_NAMES, torch.stack([t.detach() for t in terms]))
