"""Dispatch of the MLP forwards to the HIP kernels (and, when gradients are required, to the
``torch.autograd.Function`` wrappers of the backward kernels)."""
from __future__ import annotations

import torch

from . import lib


def _wants_grad(net, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in net.parameters())


def _flat3(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1]).contiguous().float()


def vf_forward(net, points: torch.Tensor, vector_only: bool = False) -> torch.Tensor:
    if not net.supports_fused():       # a geometry the fused kernels are not specialised for: layer-at-a-time row kernels
        from .batchstat import vf_forward_eval_rows
        out = vf_forward_eval_rows(net, points)
        return out[:, :3].contiguous() if vector_only else out
    if _wants_grad(net, points):
        from .backward import vf_forward_autograd
        # the trainer's supervision batches (train/vector_field_nerf_train.py:191,203,215) run the f16x3 training forward too: their
        # operands are under the range guard's watch like render()'s (a report switches the model to the exact-fp32 kernels)
        guard = getattr(net, "_range_guard", None)
        if guard is not None and guard.active() and points.is_cuda and getattr(net, "precision", "fp32") == "f16x3" and net.supports_f16x3():
            guard.poll()
            if getattr(net, "precision", "fp32") == "f16x3":
                with guard.watch(points.device) as w:
                    out = vf_forward_autograd(net, points, vector_only)
                if not w.flagged:
                    return out
                # strict mode and flagged: the guard has switched the net to the exact-fp32 kernels; this forward is repeated on them,
                # so that no call returns values the clamp touched (the abandoned graph of the first attempt is simply dropped)
        return vf_forward_autograd(net, points, vector_only)
    pts = _flat3(points)
    if vector_only and getattr(net, "precision", "fp32") == "f16x3" and net.supports_f16x3():
        guard = getattr(net, "_range_guard", None)        # the owning model's range guard (guard.py), if any
        if guard is None or not guard.active() or not pts.is_cuda:
            return lib.vf_mlp16_fwd(net.geometry(), net.packed16_weights(), pts)
        guard.poll()
        if getattr(net, "precision", "fp32") == "f16x3":
            with guard.watch(pts.device) as w:
                out = lib.vf_mlp16_fwd(net.geometry(), net.packed16_weights(), pts)
            if not w.flagged:
                return out                                 # (strict mode and flagged: fall through to the fp32 kernel)
    cols = 3 if vector_only else 3 + net._feature_dims()
    return lib.vf_mlp_fwd(net.geometry(), net.packed_weights(), pts, cols)


def render_forward(net, points, normals, view_dirs, feats) -> torch.Tensor:
    if _wants_grad(net, points, normals, view_dirs, feats) or not net.supports_fused():
        from .backward import render_forward_autograd
        return render_forward_autograd(net, points, normals, view_dirs, feats)
    return lib.render_mlp_fwd(net.geometry(), net.packed_weights(), _flat3(points), _flat3(normals),
                              _flat3(view_dirs), _flat3(feats))


def fine_pass(model, pts: torch.Tensor, z: torch.Tensor, ray_dirs: torch.Tensor):
    """Steps (7)-(11) of render(): fused VF + rendering MLPs, then density / weights / composite per ray.
    Returns (normals[M,3], colors[M,3], rgb[N,3], depth[N,1], weights[N,S_t])."""
    vf, rn = model.vector_field_network, model.rendering_network
    n, s_t = z.shape
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.unique_parameters()):
        from .backward import fine_pass_autograd
        return fine_pass_autograd(model, pts, z, ray_dirs)
    f16 = model.uses_f16x3()
    vf_w, rn_w = (vf.packed16_weights(), rn.packed16_weights()) if f16 else (vf.packed_weights(), rn.packed_weights())
    with model._timed("fused16" if f16 else "fused32"):   # bench.py: HIP events around the dominant kernel
        if f16:
            normals, colors = lib.vf_render_fused16_fwd(vf.geometry(), vf_w, rn.geometry(), rn_w, pts.view(-1, 3), ray_dirs, s_t,
                                                            colour_products=model.colour_products)
        else:
            normals, colors, _ = lib.vf_render_fused_fwd(vf.geometry(), vf_w, rn.geometry(), rn_w, pts.view(-1, 3), ray_dirs, s_t)
    _, weights, _, rgb, depth = lib.ray_density_weights(model._density_params(), normals, ray_dirs, z,
                                                        model.density.raw_scalars(), colors=colors, want_sigma=False)
    return normals, colors, rgb, depth, weights
