"""``LaplaceDensity`` parameter container (reference: models/helpers/density_functions.py:111-204).

Holds the three learnable scalars under the reference's names (state_dict keys ``beta``/``mean``/``scale``)
and the clamp bounds; the density itself is evaluated per ray inside ``vfn_ray_density_weights``
(csrc/vfn_rays.hip), which applies the same clamps.  ``forward`` is provided for callers that use the module
directly on a tensor of (negated) cosine values."""
from __future__ import annotations

from typing import Dict, Sequence

import torch
from torch import nn


class LaplaceDensity(nn.Module):
    def __init__(self, params_init: Dict[str, float] = None, beta_bounds: Sequence[float] = (1e-6, 0.0006),
                 scale_min: float = 1.0, mean_bounds: Sequence[float] = (0.5, 1.0)) -> None:
        super().__init__()
        for name, value in (params_init or {}).items():
            setattr(self, name, nn.Parameter(torch.tensor(value)))
        self.beta_bounds = torch.tensor(list(beta_bounds))
        self.scale_min = torch.tensor(scale_min)
        self.mean_bounds = torch.tensor(list(mean_bounds))
        # Density.forward drops the configured cutoff (density_functions.py:34)
        self.cutoff = -0.5

    def get_beta(self) -> torch.Tensor:
        return torch.clamp(self.beta, self.beta_bounds[0].to(self.mean.device), self.beta_bounds[1].to(self.mean.device))

    def set_beta(self, beta: torch.Tensor) -> None:
        self.beta.data = beta

    def get_scale(self) -> torch.Tensor:
        if hasattr(self, 'scale'):
            return torch.max(self.scale.abs(), self.scale_min.to(self.scale.device))
        return 1 / self.get_beta()

    def get_mean(self) -> torch.Tensor:
        return torch.clamp(self.mean, self.mean_bounds[0].to(self.mean.device), self.mean_bounds[1].to(self.mean.device))

    def raw_scalars(self) -> torch.Tensor:
        """[beta, mean, scale] as one device tensor (the kernel clamps them itself).  Cached on the parameters' (address,
        version); an optimizer that writes through raw pointers calls ``_invalidate_scalars`` (VectorFieldNerf._invalidate_packs)."""
        scale = self.scale if hasattr(self, 'scale') else 1 / self.get_beta()
        key = tuple((t.data_ptr(), t._version) for t in (self.beta, self.mean, scale)) + (getattr(self, "_scalars_epoch", 0),)
        cache = getattr(self, "_scalars_cache", None)
        if cache is None or cache[0] != key or not hasattr(self, 'scale'):
            cache = (key, torch.stack([self.beta.detach(), self.mean.detach(), scale.detach()]).float().contiguous())
            self._scalars_cache = cache
        return cache[1]

    def _invalidate_scalars(self) -> None:
        self._scalars_epoch = getattr(self, "_scalars_epoch", 0) + 1

    def forward(self, input: torch.Tensor, beta=None, scale=None, mean=None, cutoff: float = -0.5) -> torch.Tensor:
        """Element-wise density of a [M,1] tensor: a thin tensor-op path kept for API parity (the hot path
        never calls it; ``cutoff`` is ignored exactly as the reference ignores it)."""
        beta = self.get_beta() if beta is None else beta
        scale = self.get_scale() if scale is None else scale
        mean = self.get_mean() if mean is None else mean

        def cdf(x):
            return scale * (0.5 + 0.5 * torch.sign(x - mean) * (1 - torch.exp(-torch.abs(x - mean) / beta)))
        return torch.relu(cdf(input) - cdf(torch.tensor([self.cutoff], device=input.device)))
