"""Sampler objects of the facade (reference: models/samplers/ray_sampler.py:11-142, 240-302).

They carry the attributes the trainer / evaluator mutate (``near``, ``far``, ``N_samples``, ``max_samples``;
train/vector_field_nerf_train.py:43-45,128-131,146-147) AND the reference's methods: ``sample()`` / ``get_z_vals()`` run the
same device kernels ``VectorFieldNerf.render`` drives (``csrc/vfn_rays.hip``: ``vfn_uniform_sample``,
``vfn_rows_argmax`` + ``vfn_range_fine_sample``), so a caller that samples through the objects gets the arithmetic of
``render()`` — depths bit-identical to the reference's CPU path on the same uniforms.

Random numbers: the reference calls ``torch.rand`` (ray_sampler.py:138,287,292).  Here each sampler draws from its own
counter-based Philox stream on the device (``rng_seed``), or — for replaying a CPU run — from tensors queued in
``replay`` (consumed in the reference's call order: UniformSampler one draw [N,S] when not deterministic;
RangeFineSampler the jitter [N,N_f] when not deterministic, then the uniform extras [N,N_f], always)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from . import lib

_EMPTY = torch.empty(0)


class RaySampler:
    def __init__(self, near, far, N_samples: int) -> None:
        self.near = near
        self.far = far
        self._N_samples = N_samples
        self.rng_seed = 0x5a17
        self._rng_offset = 0
        self.replay: List[torch.Tensor] = []
        self._t_vals = {}

    @property
    def N_samples(self) -> int:
        return self._N_samples

    @N_samples.setter
    def N_samples(self, n: int) -> None:
        self._N_samples = n

    def active_sampler(self) -> bool:
        return self.N_samples > 0

    # -- helpers --------------------------------------------------------------------------------
    def _draw(self, shape, device) -> torch.Tensor:
        if self.replay:
            u = self.replay.pop(0)
            if tuple(u.shape) != tuple(shape):
                raise ValueError(f"replayed draw has shape {tuple(u.shape)}, the sampler needs {tuple(shape)}")
            return u.to(device).float().contiguous()
        out = torch.empty(shape, device=device)
        lib.fill_uniform(out, self.rng_seed, self._rng_offset)
        self._rng_offset += (out.numel() + 3) // 4
        return out

    def _far_args(self, device):
        if isinstance(self.far, torch.Tensor):
            return 0.0, self.far.reshape(-1).float().contiguous().to(device)
        return float(self.far), None

    def _linspace(self, n: int, device) -> torch.Tensor:
        key = (n, str(device))
        if key not in self._t_vals:       # computed by torch on the host: bit-identical to ray_sampler.py:129
            self._t_vals[key] = torch.linspace(0., 1., steps=n).to(device)
        return self._t_vals[key]

    @staticmethod
    def _rays(ray_dirs: torch.Tensor, cam_loc: torch.Tensor):
        if not ray_dirs.is_cuda:
            raise lib.VfnError("the samplers run on the device (no CPU fallback)")
        return ray_dirs.reshape(-1, 3).float().contiguous(), cam_loc.reshape(-1, 3).float().contiguous()

    # -- reference interface ----------------------------------------------------------------------
    def sample(self, ray_dirs: torch.Tensor, cam_loc: torch.Tensor, additional_depths: torch.Tensor = _EMPTY,
               device: Optional[torch.device] = None, coarse_z_vals: torch.Tensor = _EMPTY,
               coarse_weights: torch.Tensor = _EMPTY) -> Tuple[torch.Tensor, torch.Tensor]:
        """-> (points[N,S,3], z_vals[N,S]) (ray_sampler.py:49-80).  ``ray_dirs`` are the un-normalised directions render()
        passes (Q7)."""
        dirs, cam = self._rays(ray_dirs, cam_loc)
        if additional_depths.shape[0] > 0:                      # :69-73: merge extra depths, sort, recompute the points
            z = self.get_z_vals(ray_dirs, cam_loc, device=device, coarse_z_vals=coarse_z_vals, coarse_weights=coarse_weights)
            z, pts = lib.merge_sort_depths(z.contiguous(), additional_depths.to(z.device).float().contiguous(), dirs, cam)
            return pts, z
        return self._sample(dirs, cam, coarse_z_vals, coarse_weights, want_points=True)

    def get_z_vals(self, ray_dirs: torch.Tensor, cam_loc: torch.Tensor, device: Optional[torch.device] = None,
                   coarse_z_vals: torch.Tensor = _EMPTY, coarse_weights: torch.Tensor = _EMPTY) -> torch.Tensor:
        dirs, cam = self._rays(ray_dirs, cam_loc)
        return self._sample(dirs, cam, coarse_z_vals, coarse_weights, want_points=False)[1]

    def _sample(self, dirs, cam, coarse_z_vals, coarse_weights, want_points: bool):
        raise NotImplementedError


class UniformSampler(RaySampler):
    def __init__(self, N_samples: int, near, far, deterministic: bool = False) -> None:
        super().__init__(near, far, N_samples)
        self.deterministic = deterministic

    def _sample(self, dirs, cam, coarse_z_vals, coarse_weights, want_points: bool):
        n, s, dev = dirs.shape[0], self.N_samples, dirs.device
        u = None if self.deterministic else self._draw((n, s), dev)
        far, far_t = self._far_args(dev)
        z, pts = lib.uniform_sample(dirs, cam, self._linspace(s, dev), s, float(self.near), far, far_t, u, want_points)
        return pts, z


class RangeFineSampler(RaySampler):
    def __init__(self, N_samples: int, near, far, deterministic: bool = False, range: float = 0.5,
                 max_samples: int = 100, pytest: bool = False) -> None:
        super().__init__(near, far, N_samples)
        self.deterministic = deterministic
        self.pytest = pytest
        self.range = range
        self.max_samples = max_samples

    def _sample(self, dirs, cam, coarse_z_vals, coarse_weights, want_points: bool):
        """ray_sampler.py:264-302: argmax of the proposal weights, a +-range window around its depth (or uniform extras where
        the argmax is 0, Q9), merged with the proposal depths and sorted."""
        if coarse_z_vals.numel() == 0 or coarse_weights.numel() == 0:
            raise ValueError("RangeFineSampler needs coarse_z_vals and coarse_weights")
        n, dev = dirs.shape[0], dirs.device
        n_f = min(self.max_samples, self.N_samples)
        imax = lib.rows_argmax(coarse_weights.float().contiguous())
        u_fine = None if self.deterministic else self._draw((n, n_f), dev)
        u_add = self._draw((n, n_f), dev)
        far, far_t = self._far_args(dev)
        z, pts = lib.range_fine_sample(coarse_z_vals.float().contiguous(), imax, dirs, cam, n_f, self.near, far, self.range,
                                       u_add, u_fine, far_t)
        return (pts if want_points else None), z
