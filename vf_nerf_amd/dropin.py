"""``import vf_nerf_amd.dropin`` makes the reference's own import paths resolve to this implementation:

    import vf_nerf_amd.dropin                       # once, before the reference trainer/evaluator imports
    from models.nerf.vector_field_nerf import VectorFieldNerf      # -> vf_nerf_amd.nerf.VectorFieldNerf

so ``train/vector_field_nerf_train.py`` and ``evaluation/*`` run unchanged on the HIP path.  Only the modules
on the hot path are aliased; everything else (datasets, loss, config parser, mesh extraction)
keeps coming from the reference tree.  If the reference package is importable, its ``models`` / ``utils``
packages stay in place and only the leaf modules are replaced."""
from __future__ import annotations

import importlib
import sys
import types

from . import density, evaluator, grid, loss, nerf, networks, render_output, samplers, supervision

_ALIASES = {
    "models.nerf.vector_field_nerf": nerf,
    "models.nerf.output": render_output,
    "models.vector_field.vector_field_network": networks,
    "models.vector_field.rendering_network": networks,
    "models.samplers.ray_sampler": samplers,
    "models.helpers.density_functions": density,
}


def _ensure_package(name: str) -> None:
    if name in sys.modules:
        return
    try:
        importlib.import_module(name)
    except Exception:
        pkg = types.ModuleType(name)
        pkg.__path__ = []  # namespace-like placeholder
        sys.modules[name] = pkg


def install() -> None:
    for dotted, module in _ALIASES.items():
        parts = dotted.split(".")
        for i in range(1, len(parts)):
            _ensure_package(".".join(parts[:i]))
        sys.modules[dotted] = module
        parent = sys.modules[".".join(parts[:-1])]
        setattr(parent, parts[-1], module)
    # evaluation/utils/mc_utils.get_set_predictions -> pinned, vector-only, rank-sharded version
    try:
        mc = importlib.import_module("evaluation.utils.mc_utils")
        mc.get_set_predictions = grid.get_set_predictions
        # the conv3d / gather stages after the queries, as device kernels (they expect the grid on the GPU)
        mc.extract_divergence = grid.extract_divergence
        mc.unify_direction = grid.unify_direction
        mc.make_comb_format = grid.make_comb_format
        importlib.import_module("evaluation.utils.guassian_smoothing").smooth_vf = grid.smooth_vf
    except Exception:
        pass
    # the evaluator's image loop (evaluation/methods.py:472-545: one upload, one render() and four .cpu() synchronisations per
    # chunk) -> chunks on alternating streams, one download per image; same dataset, same files
    try:
        importlib.import_module("evaluation.methods").render_images = evaluator.render_images
    except Exception:
        pass
    # models.helpers.functions stays the reference's module (the trainer uses more of it); only the two host-side numpy
    # samplers the trainer calls every step are replaced by the device-side ones
    try:
        fn = importlib.import_module("models.helpers.functions")
        fn.sample_border_points = supervision.sample_border_points
        fn.sample_center_points = supervision.sample_center_points
    except Exception:
        pass
    # the loss with one device read-back per step instead of six
    try:
        importlib.import_module("models.losses.vf_loss").VFLoss = loss.VFLoss
    except Exception:
        pass


    _patch_clip_grad_norm()


_torch_clip = None


def _patch_clip_grad_norm() -> None:
    """The trainer calls ``torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip_norm)``
    (train/vector_field_nerf_train.py:254-255) on a list that names every VF parameter twice (Q4).  The sequential loop the
    reference was written against clips such a gradient twice; PyTorch's multi-tensor implementation, which newer versions
    pick on GPUs, scales duplicated tensors concurrently.  So that the trainer stays UNCHANGED, the function is wrapped: a
    list of device parameters with duplicates goes to ``optim.clip_grad_norm_`` (same semantics as the sequential loop,
    multi-tensor kernels over distinct tensors per pass); every other call reaches PyTorch's own function untouched."""
    global _torch_clip
    import torch
    from . import optim
    if _torch_clip is not None:
        return
    _torch_clip = torch.nn.utils.clip_grad_norm_

    def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
        plist = [parameters] if isinstance(parameters, torch.Tensor) else list(parameters)
        ids = [id(p) for p in plist]
        if len(set(ids)) != len(ids) and float(norm_type) == 2.0 and not error_if_nonfinite and \
                all(p.is_cuda for p in plist if p.grad is not None):
            return optim.clip_grad_norm_(plist, max_norm, 2.0)
        return _torch_clip(plist, max_norm, norm_type, error_if_nonfinite, foreach)

    clip_grad_norm_.__wrapped__ = _torch_clip
    torch.nn.utils.clip_grad_norm_ = clip_grad_norm_


def uninstall_clip_grad_norm() -> None:
    """Put PyTorch's own clip_grad_norm_ back (tests)."""
    global _torch_clip
    import torch
    if _torch_clip is not None:
        torch.nn.utils.clip_grad_norm_ = _torch_clip
        _torch_clip = None


install()
