"""``NerfOutput``: what ``VectorFieldNerf.render`` returns (interface of the reference's models/nerf/output.py:7-70).

The attribute names are the reference's, including its quirk that every ``*_coarse`` / ``coarse_*`` attribute holds
the result of the S_c+N_f ("fine") pass while the ``*_fine`` / ``fine_*`` attributes stay ``None``.
The class is generated from the two name tables below: four mandatory tensors, then the optional ones.
"""
from __future__ import annotations

from dataclasses import field, make_dataclass
from typing import Optional

import torch

_MANDATORY = "points_coarse coarse_normals coarse_rgb_values coarse_depth_map".split()
_OPTIONAL = ("mask z_vals points_fine fine_normals fine_rgb_values fine_depth_map fine_mask "
             "directional_derivtives ray_dirs coarse_colors").split()   # "derivtives": the reference's spelling
_DICT_KEYS = _MANDATORY + ["mask"] + [k for k in _OPTIONAL if k.startswith(("points_", "fine_"))]


def _adjacent(t: torch.Tensor, n_rays: int, per_ray: int):
    """[n_rays*per_ray, 3] -> (sample j, sample j+1) for j < per_ray-1, each flattened to [*, 3]."""
    t = t.reshape(n_rays, per_ray, 3)
    return t[:, :-1].reshape(-1, 3), t[:, 1:].reshape(-1, 3)


def _fine_active(self) -> bool:
    return self.fine_normals is not None


def _get_normals(self, N_rays: int, N_coarse: int, N_fine: int):
    """Adjacent-sample normal pairs per pass: (coarse_j, coarse_j+1, fine_j | None, fine_j+1 | None)."""
    first = _adjacent(self.coarse_normals, N_rays, N_coarse)
    second = _adjacent(self.fine_normals, N_rays, N_fine) if self.fine_active() else (None, None)
    return first + second


def _to_dict(self):
    return {k: getattr(self, k) for k in _DICT_KEYS}


class RepeatedRows:
    """``rows[N,3]`` standing for the reference's ``ray_dirs`` output, every ray's direction repeated for each of its ``times``
    samples ([N * times, 3], models/nerf/vector_field_nerf.py:239-249,336).  Nothing in the reference's trainer, evaluator or loss
    reads that field, and materialising it is a launch and 12 bytes per sample on every render: it is built on first access."""

    def __init__(self, rows: torch.Tensor, times: int) -> None:
        self.rows, self.times = rows, int(times)

    def materialise(self) -> torch.Tensor:
        n = self.rows.shape[0]
        return self.rows.reshape(n, 1, 3).expand(n, self.times, 3).reshape(-1, 3)


class LazyColours:
    """``coarse_colors`` of a render that ran the rendering net on the SELECTED samples only (the sparse colour branch of a training step's
    render, or ``model.sparse_colours``): ``sparse[M,3]`` holds the selected samples' colours and zeros elsewhere.  The reference returns the
    rendering net's colour of EVERY sample (models/nerf/vector_field_nerf.py:315-338) although nothing in its trainer, evaluator or loss reads
    the field; so the other rows are evaluated when — and only when — somebody does: ``fill()`` runs the dense gradient-free fused launch on
    the render's points with the weights the render saw and returns [M,3]; rows the render computed keep the render's values."""

    def __init__(self, sparse: torch.Tensor, fill) -> None:
        self.sparse, self._fill = sparse, fill

    def materialise(self) -> torch.Tensor:
        dense = self._fill()
        self._fill = None
        ran = (self.sparse != 0).any(dim=-1, keepdim=True)       # (a sigmoid is never exactly 0: a zero row was not evaluated)
        return torch.where(ran, self.sparse, dense)


def _getattribute(self, name):
    value = object.__getattribute__(self, name)
    if name == "ray_dirs" and isinstance(value, RepeatedRows):
        value = value.materialise()
        object.__setattr__(self, "ray_dirs", value)
    elif name == "coarse_colors" and isinstance(value, LazyColours):
        value = value.materialise()
        object.__setattr__(self, "coarse_colors", value)
    return value


NerfOutput = make_dataclass(
    "NerfOutput",
    [(n, torch.Tensor) for n in _MANDATORY] + [(n, Optional[torch.Tensor], field(default=None)) for n in _OPTIONAL],
    namespace=dict(fine_active=_fine_active, get_normals=_get_normals, to_dict=_to_dict, __getattribute__=_getattribute, __module__=__name__))
