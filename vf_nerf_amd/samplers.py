"""Sampler objects of the facade (reference: models/samplers/ray_sampler.py:11-142, 240-302).

They carry the attributes the trainer / evaluator mutate (``near``, ``far``, ``N_samples``,
``max_samples``; train/vector_field_nerf_train.py:43-45,128-131,146-147); the sampling arithmetic itself
runs in ``csrc/vfn_rays.hip`` (K1 ``vfn_raygen_uniform`` and K3b ``vfn_range_fine_sample``), driven by
``VectorFieldNerf.render``."""
from __future__ import annotations


class RaySampler:
    def __init__(self, near, far, N_samples: int) -> None:
        self.near = near
        self.far = far
        self._N_samples = N_samples

    @property
    def N_samples(self) -> int:
        return self._N_samples

    @N_samples.setter
    def N_samples(self, n: int) -> None:
        self._N_samples = n

    def active_sampler(self) -> bool:
        return self.N_samples > 0


class UniformSampler(RaySampler):
    def __init__(self, N_samples: int, near, far, deterministic: bool = False) -> None:
        super().__init__(near, far, N_samples)
        self.deterministic = deterministic


class RangeFineSampler(RaySampler):
    def __init__(self, N_samples: int, near, far, deterministic: bool = False, range: float = 0.5,
                 max_samples: int = 100, pytest: bool = False) -> None:
        super().__init__(near, far, N_samples)
        self.deterministic = deterministic
        self.pytest = pytest
        self.range = range
        self.max_samples = max_samples
