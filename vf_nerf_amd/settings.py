"""Configuration records accepted by the facade.

Record names, field names, field order, defaults and validation follow the interface of the reference's
``config_parser/vf_nerf_config.py:10-132`` so a ``VFNerfConfig`` built for the reference constructs this
implementation unchanged (the facade only reads attributes; instances of the reference's own dataclasses work too).
Every record is generated from the ``_SPEC`` table: ``name -> [(field, type, default | _REQ | factory)]``.
``shipped_config()`` returns the values of ``confs/vf_nerf.conf`` without needing pyhocon.
"""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Any, Callable, Dict, List, Optional

import torch

_REQ = object()          # marks a field without default


def _fresh(value) -> Callable[[], Any]:
    """default_factory returning a new copy of a list / dict default."""
    return lambda: type(value)(value)


_SPEC: Dict[str, list] = {
    "DensityConfig": [("beta_bounds", List[float], _fresh([1e-4, 1e9])), ("mean_bounds", List[float], _fresh([0.6, 1.0])),
                      ("scale_min", float, 0.1),
                      ("params_init", Dict[str, float], _fresh(dict(beta=0.5, mean=0.7, scale=100.0))),
                      ("cutoff", float, -0.5)],
    "VFNetConfig": [("input_dims", int, _REQ), ("output_dims", int, _REQ), ("dimensions", List[int], _REQ),
                    ("feature_vector_dims", int, 0), ("embedder_multires", int, 0), ("weight_norm", bool, True),
                    ("batch_norm", bool, True), ("skip_connection_in", Optional[List[int]], None),
                    ("bias_init", float, 0.0), ("dropout", bool, True), ("dropout_probability", float, 0.0),
                    ("xavier_init", bool, True), ("init", str, "center")],
    "RenderingNetConfig": [("output_dims", int, _REQ), ("dimensions", List[int], _REQ), ("feature_vector_dims", int, 0),
                           ("weight_norm", bool, False), ("batch_norm", bool, True), ("mode", str, "idr"),
                           ("embedder_multires", int, 0), ("detach_normals", bool, False)],
    "RaySamplerConfig": [("n_samples", int, 64), ("n_importance", int, 64), ("rays_per_batch", int, 1024),
                         ("perturb", bool, True), ("near", float, 0.0), ("far", float, 1.0), ("fine_range", float, 0.5),
                         ("increase_every", int, 100), ("max_samples", int, 100)],
    "CudaConfig": [("device", torch.device, torch.device('cuda')), ("num_gpus", int, 1)],
    "SchedulerConfig": [("lr", float, 1e-3), ("lr_decay_factor", float, 0.5), ("lr_decay_steps", int, 50000),
                        ("clip_norm", float, 0.5), ("weight_decay", float, 0.0)],
}
_SPEC["VFNerfConfig"] = [(n, Any, _REQ) for n in (
    "vf_net_config", "rendering_net_config", "ray_sampler_config", "cuda_config", "scheduler_config", "density_config",
    "cos_sim_weights", "cos_sim_weights_anneal", "anneal_start", "anneal_end", "rendering", "normalize_rendering")] + [
    ("dir_to_normal_th", float, -2.0), ("numerical_jacobian", bool, False), ("border_supervision", bool, True),
    ("center_supervision", bool, True)]


def _density_todict(self) -> Dict[str, Any]:
    # cutoff is deliberately absent: the reference never forwards it to the density.
    return {k: getattr(self, k) for k in ("beta_bounds", "mean_bounds", "scale_min", "params_init")}


def _validate_top(self) -> None:
    if self.cos_sim_weights_anneal not in ("none", "hard", "soft"):
        raise ValueError(f"Invalid cos_sim_weights_anneal: {self.cos_sim_weights_anneal}")
    if self.rendering not in ("nerf", "volsdf"):
        raise ValueError(f"Invalid rendering: {self.rendering}")
    self.cos_sim_weights = torch.as_tensor(self.cos_sim_weights).float().to(self.cuda_config.device)


_METHODS: Dict[str, Dict[str, Callable]] = {
    "DensityConfig": dict(todict=_density_todict),
    "RaySamplerConfig": dict(fine_sampling=lambda self: self.n_importance > 0),
    "VFNerfConfig": dict(__post_init__=_validate_top,
                         cos_sim_weights_dict=lambda self: {f"w_{i}": w.item()
                                                            for i, w in enumerate(self.cos_sim_weights)}),
}


def _make(name: str):
    cols = []
    for fname, ftype, default in _SPEC[name]:
        if default is _REQ:
            cols.append((fname, ftype))
        elif callable(default):
            cols.append((fname, ftype, field(default_factory=default)))
        else:
            cols.append((fname, ftype, field(default=default)))
    return make_dataclass(name, cols, namespace=dict(_METHODS.get(name, {}), __module__=__name__))


DensityConfig = _make("DensityConfig")
VFNetConfig = _make("VFNetConfig")
RenderingNetConfig = _make("RenderingNetConfig")
RaySamplerConfig = _make("RaySamplerConfig")
CudaConfig = _make("CudaConfig")
SchedulerConfig = _make("SchedulerConfig")
VFNerfConfig = _make("VFNerfConfig")


def shipped_config(device: torch.device, n_samples: int = 100, n_importance: int = 30, perturb: bool = True,
                   near: float = 0.0, far: float = 1.0, fine_range: float = 0.3, max_samples: int = 100,
                   dir_to_normal_th: float = -2.0, n_window: int = 11, anneal: str = "hard",
                   num_gpus: int = 1) -> "VFNerfConfig":
    """The network / density / scheduler values of ``confs/vf_nerf.conf`` with the sampler sizes
    overridable (benchmarks and tests use 64+64, 32+32 ...)."""
    return VFNerfConfig(
        vf_net_config=VFNetConfig(input_dims=3, output_dims=3, dimensions=[256] * 8, feature_vector_dims=256,
                                  embedder_multires=6, weight_norm=False, batch_norm=True,
                                  skip_connection_in=[4], bias_init=0.0, dropout=False,
                                  dropout_probability=0.2, xavier_init=False, init=""),
        rendering_net_config=RenderingNetConfig(output_dims=3, dimensions=[256] * 4, feature_vector_dims=256,
                                                weight_norm=False, batch_norm=True, mode="idr",
                                                embedder_multires=4, detach_normals=True),
        ray_sampler_config=RaySamplerConfig(n_samples=n_samples, n_importance=n_importance, rays_per_batch=1024,
                                            perturb=perturb, near=near, far=far, fine_range=fine_range,
                                            increase_every=50, max_samples=max_samples),
        cuda_config=CudaConfig(device=device, num_gpus=num_gpus),
        scheduler_config=SchedulerConfig(lr=5e-4, lr_decay_factor=0.1, clip_norm=0.5, weight_decay=0.0),
        density_config=DensityConfig(beta_bounds=[1e-4, 1e9], mean_bounds=[0.6, 1.0], scale_min=1.0,
                                     params_init=dict(beta=0.5, scale=100.0, mean=0.7), cutoff=-2.0),
        cos_sim_weights=[0.09] * n_window, cos_sim_weights_anneal=anneal, anneal_start=700, anneal_end=1400,
        rendering="volsdf", normalize_rendering=True, dir_to_normal_th=dir_to_normal_th,
        numerical_jacobian=False, border_supervision=True, center_supervision=True)
