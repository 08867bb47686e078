"""Synthetic scenes: cameras and random-weight networks that still composite non-trivially.

Neither datasets nor trained checkpoints are available offline (the init ``.pth`` files are
LFS pointers), and with default-initialised MLPs the vector field is so smooth that the Laplace density is
identically zero (Q5).  The recipe of SURVEY.md §8(d): default PyTorch initialisation under a fixed seed,
hidden Linear weights scaled by ``gain``, and the three vector rows of the last VF layer re-centred /
re-scaled so that the pre-tanh vector components have zero mean and unit variance over the frustum —
the field then flips sign often enough for rays to hit "surfaces".
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch


def pinhole_batch(n_rays: int, width: int, height: int, focal: float, seed: int, device="cpu",
                  pose: torch.Tensor = None, skew: float = 0.0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Random pixels of one pinhole view -> (uv[N,2] float, pose[N,4,4], intrinsics[N,4,4]), replicated per ray
    like the reference datasets emit them (datasets/normal_datasets/replica_dataset.py:146-212)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, width, (n_rays,), generator=g).float()
    v = torch.randint(0, height, (n_rays,), generator=g).float()
    uv = torch.stack([u, v], dim=1)
    K = torch.eye(4)
    K[0, 0] = focal
    K[1, 1] = focal
    K[0, 1] = skew
    K[0, 2] = (width - 1) / 2
    K[1, 2] = (height - 1) / 2
    P = torch.eye(4) if pose is None else pose.float()
    return (uv.to(device), P.repeat(n_rays, 1, 1).contiguous().to(device), K.repeat(n_rays, 1, 1).contiguous().to(device))


def pinhole_image(width: int, height: int, focal: float, device="cpu", pose: torch.Tensor = None
                  ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Every pixel of one pinhole view in row-major order -> (uv[W*H,2], pose[W*H,4,4], intrinsics[W*H,4,4]);
    the principal point is the image centre, so a 1/k-resolution image of the same camera uses focal/k."""
    v, u = torch.meshgrid(torch.arange(height).float(), torch.arange(width).float(), indexing="ij")
    uv = torch.stack([u.reshape(-1), v.reshape(-1)], dim=1)
    K = torch.eye(4)
    K[0, 0] = K[1, 1] = focal
    K[0, 2], K[1, 2] = (width - 1) / 2, (height - 1) / 2
    P = torch.eye(4) if pose is None else pose.float()
    n = width * height
    return uv.to(device), P.repeat(n, 1, 1).contiguous().to(device), K.repeat(n, 1, 1).contiguous().to(device)


def orbit_pose(azimuth_deg: float, elevation_deg: float, radius: float, target=(0.0, 0.0, 0.6)) -> torch.Tensor:
    """Camera-to-world 4x4 looking at ``target`` from a sphere around it (+z forward, OpenCV style)."""
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    t = torch.tensor(target)
    eye = t + radius * torch.tensor([math.cos(el) * math.sin(az), math.sin(el), -math.cos(el) * math.cos(az)])
    fwd = (t - eye) / (t - eye).norm()
    right = torch.linalg.cross(torch.tensor([0.0, 1.0, 0.0]), fwd)
    right = right / right.norm()
    up = torch.linalg.cross(fwd, right)
    P = torch.eye(4)
    P[:3, 0], P[:3, 1], P[:3, 2], P[:3, 3] = right, up, fwd, eye
    return P


@torch.no_grad()
def scale_hidden_weights(vf_net, rn_net, gain: float) -> None:
    """Multiply the hidden Linear weights (all but the last layer of each net) by ``gain``."""
    if gain == 1.0:
        return
    for net in (vf_net, rn_net):
        n = len(net.layers)
        for i in range(n - 1):
            layer = net.layers[i]
            lin = layer[0] if isinstance(layer, torch.nn.Sequential) else layer
            lin.weight.mul_(gain)


@torch.no_grad()
def recentre_vector_head(vf_net, pre_mean: torch.Tensor, pre_std: torch.Tensor) -> None:
    """Given mean / std [3] of the pre-tanh vector columns over the frustum, standardise them by editing
    rows 0:3 of the last Linear: W <- W / std, b <- (b - mean) / std."""
    last = vf_net.layers[len(vf_net.layers) - 1]
    m = pre_mean.to(last.weight.device)
    s = pre_std.to(last.weight.device)
    last.bias[:3] = (last.bias[:3] - m) / s
    last.weight[:3] = last.weight[:3] / s[:, None]


@torch.no_grad()
def vector_head_stats_from_tanh(out_vec: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """mean / std of the pre-activation recovered from tanh outputs (atanh), for use with the HIP forward."""
    pre = torch.atanh(out_vec.double().clamp(-1 + 1e-7, 1 - 1e-7))
    return pre.mean(0).float(), pre.std(0).float()


def frustum_points(n: int, seed: int, half_width: float = 0.6, near: float = 0.0, far: float = 1.0) -> torch.Tensor:
    """Uniform samples in the axis-aligned box around a z-forward frustum."""
    g = torch.Generator().manual_seed(seed)
    p = torch.rand(n, 3, generator=g)
    p[:, 0] = (p[:, 0] * 2 - 1) * half_width
    p[:, 1] = (p[:, 1] * 2 - 1) * half_width
    p[:, 2] = near + p[:, 2] * (far - near)
    return p


def weights_checksum(state_dicts: Dict[str, Dict[str, torch.Tensor]]) -> Dict[str, float]:
    """Order-independent fp64 fingerprints of a set of state dicts (stored beside golden vectors)."""
    tot, tot_abs, cnt = 0.0, 0.0, 0
    for sd in state_dicts.values():
        for k in sorted(sd):
            t = sd[k]
            if not torch.is_floating_point(t):
                continue
            t = t.detach().double().cpu()
            tot += float(t.sum())
            tot_abs += float(t.abs().sum())
            cnt += t.numel()
    return {"sum": tot, "abs_sum": tot_abs, "count": float(cnt)}
