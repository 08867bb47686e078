"""``NerfOutput``: what ``VectorFieldNerf.render`` returns (reference: models/nerf/output.py:7-70).

Naming follows the reference verbatim, including its quirk that every ``*_coarse`` field holds the result
of the S_c+N_f ("fine") pass and the ``*_fine`` fields stay ``None``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch


@dataclass
class NerfOutput:
    points_coarse: torch.Tensor
    coarse_normals: torch.Tensor
    coarse_rgb_values: torch.Tensor
    coarse_depth_map: torch.Tensor
    mask: Optional[torch.Tensor] = None
    z_vals: Optional[torch.Tensor] = None
    points_fine: Optional[torch.Tensor] = None
    fine_normals: Optional[torch.Tensor] = None
    fine_rgb_values: Optional[torch.Tensor] = None
    fine_depth_map: Optional[torch.Tensor] = None
    fine_mask: Optional[torch.Tensor] = None
    directional_derivtives: Optional[torch.Tensor] = None
    ray_dirs: Optional[torch.Tensor] = None
    coarse_colors: Optional[torch.Tensor] = None

    def fine_active(self) -> bool:
        return self.fine_normals is not None

    def get_normals(self, N_rays: int, N_coarse: int, N_fine: int
                    ) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """Adjacent-sample normal pairs (this, next) per pass, flattened to [*,3]."""
        def pairs(t: torch.Tensor, s: int):
            t = t.reshape(N_rays, s, 3)
            return t[:, :-1, :].reshape(-1, 3), t[:, 1:, :].reshape(-1, 3)
        c0, c1 = pairs(self.coarse_normals, N_coarse)
        if not self.fine_active():
            return c0, c1, None, None
        f0, f1 = pairs(self.fine_normals, N_fine)
        return c0, c1, f0, f1

    def to_dict(self) -> Dict[str, Optional[torch.Tensor]]:
        keys = ("points_coarse", "coarse_normals", "coarse_rgb_values", "coarse_depth_map", "mask", "points_fine",
                "fine_normals", "fine_rgb_values", "fine_depth_map", "fine_mask")
        return {k: getattr(self, k) for k in keys}
