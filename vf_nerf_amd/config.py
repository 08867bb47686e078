"""Configuration dataclasses accepted by the facade.

Field names, defaults and validation mirror the reference's ``config_parser/vf_nerf_config.py:10-132``
so a ``VFNerfConfig`` built for the reference constructs this implementation unchanged (the
facade only reads attributes; instances of the reference's own dataclasses work too).
``shipped_config()`` returns the values of ``confs/vf_nerf.conf`` without needing pyhocon.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional

import torch


@dataclass
class DensityConfig:
    beta_bounds: List[float] = field(default_factory=lambda: [1e-4, 1e9])
    mean_bounds: List[float] = field(default_factory=lambda: [0.6, 1.0])
    scale_min: float = 0.1
    params_init: Dict[str, float] = field(default_factory=lambda: dict(beta=0.5, mean=0.7, scale=100.0))
    cutoff: float = -0.5

    def todict(self) -> Dict[str, Any]:
        # cutoff is deliberately absent: the reference never forwards it (Q5).
        return dict(beta_bounds=self.beta_bounds, mean_bounds=self.mean_bounds,
                    scale_min=self.scale_min, params_init=self.params_init)


@dataclass
class VFNetConfig:
    input_dims: int
    output_dims: int
    dimensions: List[int]
    feature_vector_dims: int = 0
    embedder_multires: int = 0
    weight_norm: bool = True
    batch_norm: bool = True
    skip_connection_in: Optional[List[int]] = None
    bias_init: float = 0.0
    dropout: bool = True
    dropout_probability: float = 0.0
    xavier_init: bool = True
    init: str = "center"


@dataclass
class RenderingNetConfig:
    output_dims: int
    dimensions: List[int]
    feature_vector_dims: int = 0
    weight_norm: bool = False
    batch_norm: bool = True
    mode: str = "idr"
    embedder_multires: int = 0
    detach_normals: bool = False


@dataclass
class RaySamplerConfig:
    n_samples: int = 64
    n_importance: int = 64
    rays_per_batch: int = 1024
    perturb: bool = True
    near: float = 0.0
    far: float = 1.0
    fine_range: float = 0.5
    increase_every: int = 100
    max_samples: int = 100

    def fine_sampling(self) -> bool:
        return self.n_importance > 0


@dataclass
class CudaConfig:
    device: torch.device = torch.device('cuda')
    num_gpus: int = 1


@dataclass
class SchedulerConfig:
    lr: float = 1e-3
    lr_decay_factor: float = 0.5
    lr_decay_steps: int = 50000
    clip_norm: float = 0.5
    weight_decay: float = 0.0


@dataclass
class VFNerfConfig:
    vf_net_config: VFNetConfig
    rendering_net_config: RenderingNetConfig
    ray_sampler_config: RaySamplerConfig
    cuda_config: CudaConfig
    scheduler_config: SchedulerConfig
    density_config: DensityConfig

    cos_sim_weights: Any
    cos_sim_weights_anneal: str
    anneal_start: int
    anneal_end: int

    rendering: str
    normalize_rendering: bool
    dir_to_normal_th: float = -2.0
    numerical_jacobian: bool = False
    border_supervision: bool = True
    center_supervision: bool = True

    def __post_init__(self) -> None:
        if self.cos_sim_weights_anneal not in ("none", "hard", "soft"):
            raise ValueError(f"Invalid cos_sim_weights_anneal: {self.cos_sim_weights_anneal}")
        if self.rendering not in ("nerf", "volsdf"):
            raise ValueError(f"Invalid rendering: {self.rendering}")
        self.cos_sim_weights = torch.as_tensor(self.cos_sim_weights).float().to(self.cuda_config.device)

    def cos_sim_weights_dict(self) -> Dict[str, float]:
        return {f"w_{i}": self.cos_sim_weights[i].item() for i in range(len(self.cos_sim_weights))}


def shipped_config(device: torch.device, n_samples: int = 100, n_importance: int = 30, perturb: bool = True,
                   near: float = 0.0, far: float = 1.0, fine_range: float = 0.3, max_samples: int = 100,
                   dir_to_normal_th: float = -2.0, n_window: int = 11, anneal: str = "hard",
                   num_gpus: int = 1) -> VFNerfConfig:
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
