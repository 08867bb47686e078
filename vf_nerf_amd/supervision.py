"""Device-side supervision points of the trainer.

The reference draws them with numpy on the host and uploads them every step
(train/vector_field_nerf_train.py:186-214 -> models/helpers/functions.py:100-135 -> models/samplers/sampler.py:160-193).
Same function names, arguments and return values here; the samples come from the library's Philox stream on the device
(csrc/vfn_rays.hip: vfn_sphere_shell_kernel), so there is no host work and no upload.  The random STREAM necessarily
differs from numpy's; the distribution and the ground truth are the reference's (tests replay explicit uniforms through
both).  ``vf_nerf_amd.dropin`` installs the two samplers into ``models.helpers.functions``.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.nn.functional as F

from . import lib

_seed = 0x5eed
_offset = 0
_replay = []          # explicit unit uniforms [n,3] consumed by the next sampler calls instead of the Philox stream


def manual_seed(seed: int) -> None:
    """Seed of the supervision stream (independent of the render() sampling stream)."""
    global _seed, _offset
    _seed, _offset = int(seed), 0


def replay_uniforms(*draws: torch.Tensor) -> None:
    """Queue explicit uniforms (each [n,3]: azimuth, cos(polar angle), radius draw, all in [0,1)) for the next sampler calls,
    in call order — replaying a host run of the reference's numpy sampler (models/samplers/sampler.py:176-183) through the
    device kernel.  An empty queue means the Philox stream."""
    _replay.extend(draws)


def _shell(r_min: float, r_max: float, num_samples: int, centroid: torch.Tensor, device, inward: bool):
    global _offset
    n = int(num_samples)
    u = None
    if _replay:
        u = _replay.pop(0)
        if tuple(u.shape) != (n, 3):
            raise ValueError(f"replayed uniforms have shape {tuple(u.shape)}, the sampler needs {(n, 3)}")
    # Inside an open training step (a grad-mode render() of the shipped regime, stepengine.StepSession) the points go straight into the
    # step workspace's supervision rows: the vector-field forward the trainer runs on them next (train/vector_field_nerf_train.py:201,213)
    # is then a vector-only saving forward on those rows, differentiated by the step's one chain.  Same kernel, same draws, same values.
    if torch.is_grad_enabled() and n > 0:
        from .stepengine import current_session
        session = current_session()
        want = torch.device(device)
        if session is not None and want.type == session.ws.device.type and want.index in (None, session.ws.device.index):
            got = session.sample(inward, float(r_min), float(r_max), centroid, n, None if u is None else u.to(session.ws.device).float().contiguous(),
                                 _seed, _offset)
            if got is not None:
                if u is None:
                    _offset += n
                return got
    c = torch.as_tensor(centroid, dtype=torch.float32, device=device).reshape(3).contiguous()
    if u is not None:
        return lib.sample_sphere_shell(n, float(r_min), float(r_max), c, inward, u=u.to(c.device).float().contiguous())
    pts, gt = lib.sample_sphere_shell(n, float(r_min), float(r_max), c, inward, _seed, _offset)
    _offset += n
    return pts, gt


def sample_border_points(r_min: float, r_max: float, num_samples: int, centroid: torch.Tensor,
                         device: torch.device = "cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """Points in the shell r_min..r_max around the centroid and unit vectors pointing at the centroid
    (functions.py:100-117)."""
    return _shell(r_min, r_max, num_samples, centroid, device, inward=True)


def sample_center_points(centroid: torch.Tensor, radius: float, num_samples: int,
                         device: torch.device = "cuda") -> Tuple[torch.Tensor, torch.Tensor]:
    """Points in the ball of the given radius around the centroid and unit vectors pointing away from it
    (functions.py:119-135)."""
    return _shell(0.0, radius, num_samples, centroid, device, inward=False)


def get_border_indices_and_gt(points: torch.Tensor, normals: torch.Tensor, far: float, radius: float,
                              centroid: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Normals of the ray samples farther than far/2 - radius from the centroid, and unit vectors from those samples to
    the centroid (functions.py:75-98).  points / normals: [N,S,3]."""
    keep = torch.linalg.vector_norm(points - centroid, dim=2) > (far / 2 - radius)
    return normals.reshape(points.shape)[keep], F.normalize(centroid - points[keep], dim=1)


def get_center_indices_and_gt(points: torch.Tensor, normals: torch.Tensor, centroid: torch.Tensor,
                              radius: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Normals of the ray samples closer than ``radius`` to the centroid, and unit vectors from the centroid to them
    (functions.py:137-157).

    Called on the outputs of an open training step (stepengine.StepSession) the rows are DEFERRED: both results are empty [0,3] tensors, the
    prediction one carrying a marker node that ``loss.VFLoss`` finds behind ``pred["supervised_normals"]``; its fused kernels then select,
    count and differentiate the same rows on the device (csrc/vfn_loss.hip: ray_center) — the same loss value and gradients without the
    boolean-mask indexing, i.e. without a device synchronisation in the middle of the step.  A loss other than ``loss.VFLoss`` would see no
    such rows: the step's backward refuses to run then (``model.defer_center_rows = False`` keeps the compaction)."""
    if torch.is_grad_enabled() and points.is_cuda:
        from .stepengine import centre_rows, current_session
        session = current_session()
        if session is not None and getattr(session.model, "defer_center_rows", True):
            got = centre_rows(session, points, normals, centroid, radius)
            if got is not None:
                return got
    keep = torch.linalg.vector_norm(points - centroid, dim=2) < radius
    return normals.reshape(points.shape)[keep], F.normalize(points[keep] - centroid, dim=1)


def center_rows_dense(points: torch.Tensor, normals: torch.Tensor, centroid: torch.Tensor, radius: float):
    """``get_center_indices_and_gt`` without the compaction: every ray sample keeps its row, rows outside the centre ball are
    ZERO in both the prediction and the ground truth (they add nothing to a sum of squared differences and receive no gradient),
    and the number of selected rows comes back as a device scalar -> (normals[N*S,3] masked, gt[N*S,3] masked, count).  No
    boolean-mask indexing, hence no host synchronisation (trainer.TrainStep uses it with VFLoss's ``supervised_rows``)."""
    diff = points - centroid
    keep = (torch.linalg.vector_norm(diff, dim=2) < radius).unsqueeze(-1)
    gt = F.normalize(diff, dim=2)
    zero = torch.zeros((), device=points.device, dtype=normals.dtype)
    return (torch.where(keep, normals.reshape(points.shape), zero).reshape(-1, 3), torch.where(keep, gt, zero).reshape(-1, 3),
            keep.sum().to(normals.dtype))
