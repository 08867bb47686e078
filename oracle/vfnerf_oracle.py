"""CPU oracle for the VF-NeRF volume-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain PyTorch-CPU (fp32) restatement of the
reference algorithm behind ``VectorFieldNerf.render`` so that the HIP path can be
checked on a box where ``/root/reference`` does not exist.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it;
the product package ``vf_nerf_amd`` never does.

Parity pin: the reference has no tests / golden vectors of its own, so
this oracle is pinned against outputs of the reference itself, captured in the build
container by ``tests/golden/make_golden.py`` (imports ``/root/reference`` read-only) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them.  The
trainer-side restatements (``vf_loss_terms``, ``sphere_shell_points_from_draws``, the
supervision selections, ``trainer_epoch``) are pinned the same way by
``tests/golden/make_train_golden.py``, which runs the reference's own ``train_epoch``,
``VFLoss`` and ``SphereSampler`` (``trainer_steps.npz``): three optimizer steps reproduced
bit for bit.

Every function cites the reference lines it restates (paths relative to the reference
root).  Weights are passed as state dicts using the reference's key names
(``layers.{i}.0.weight`` … ``layers.8.bias``) so real checkpoints can be fed in.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# a1: rays  (utils/rendering.py:12-60, utils/pinhole_model.py:9-63)
# --------------------------------------------------------------------------------------
def quaternion_pose_to_matrix(pose7: Tensor) -> Tensor:
    """[N,7] (qr,qi,qj,qk,tx,ty,tz) -> [N,4,4].  utils/pinhole_model.py:9-33 and
    utils/rendering.py:27-33 (the reference hard-codes .cuda() there, Q13; the math is
    restated device-free)."""
    q = F.normalize(pose7[:, :4], dim=1)
    r, i, j, k = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    n = pose7.shape[0]
    p = torch.eye(4, dtype=pose7.dtype).repeat(n, 1, 1)
    p[:, 0, 0] = 1 - 2 * (j ** 2 + k ** 2)
    p[:, 0, 1] = 2 * (j * i - k * r)
    p[:, 0, 2] = 2 * (i * k + r * j)
    p[:, 1, 0] = 2 * (j * i + k * r)
    p[:, 1, 1] = 1 - 2 * (i ** 2 + k ** 2)
    p[:, 1, 2] = 2 * (j * k - i * r)
    p[:, 2, 0] = 2 * (k * i - j * r)
    p[:, 2, 1] = 2 * (j * k + i * r)
    p[:, 2, 2] = 1 - 2 * (i ** 2 + j ** 2)
    p[:, :3, 3] = pose7[:, 4:]
    return p


def ray_directions(uv: Tensor, pose: Tensor, intrinsics: Tensor):
    """uv[N,2], pose[N,4,4]|[N,7], K[N,4,4] -> directions[N,3] (un-normalised, camera z=±1),
    ray_dirs[N,3] (unit), cam_loc[N,3].  utils/rendering.py:12-60 + pinhole_model.py:36-63.
    The z sign is read from ray 0's fy only (rendering.py:42)."""
    if pose.shape[1] == 7:
        pose = quaternion_pose_to_matrix(pose)
    cam_loc = pose[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0], intrinsics[:, 1, 1]
    cx, cy = intrinsics[:, 0, 2], intrinsics[:, 1, 2]
    sk = intrinsics[:, 0, 1]
    u, v = uv[:, 0], uv[:, 1]
    z = torch.ones(uv.shape[0], dtype=uv.dtype) * torch.sign(intrinsics[0, 1, 1])
    x = (u - cx + cy * sk / fy - sk * v / fy) / fx * z.abs()
    y = (v - cy) / fy * z.abs()
    cam_h = torch.stack([x, y, z, torch.ones_like(z)], dim=-1).unsqueeze(-1)  # [N,4,1]
    world = torch.bmm(pose, cam_h)[:, :3, 0]
    directions = world - cam_loc
    ray_dirs = F.normalize(directions, dim=1)
    return directions, ray_dirs, cam_loc


# --------------------------------------------------------------------------------------
# a2: coarse sampler  (models/samplers/ray_sampler.py:49-80, 113-142)
# --------------------------------------------------------------------------------------
def stratify(z: Tensor, u: Tensor) -> Tensor:
    """Stratified jitter inside the half-way intervals (ray_sampler.py:132-140, 282-289)."""
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * u


def uniform_z_vals(n_rays: int, n_samples: int, near: float, far, u: Optional[Tensor] = None) -> Tensor:
    """z[N,S_c].  ``far`` is a float or a per-ray [N,1] tensor (ray_sampler.py:126-127).
    ``u`` = the torch.rand(N,S_c) draw when perturb is on, None when deterministic."""
    near_t = near * torch.ones(n_rays, 1)
    far_t = far * torch.ones(n_rays, 1) if isinstance(far, float) else far
    t = torch.linspace(0.0, 1.0, steps=n_samples)
    z = near_t * (1.0 - t) + far_t * t
    if u is not None:
        z = stratify(z, u)
    return z


def points_along_rays(cam_loc: Tensor, directions: Tensor, z: Tensor) -> Tensor:
    """[N,S,3] = o + z * d with the UN-normalised d (ray_sampler.py:77-78; Q7)."""
    return cam_loc.unsqueeze(1) + z.unsqueeze(2) * directions.unsqueeze(1)


# --------------------------------------------------------------------------------------
# a3: positional encoding  (models/helpers/embedder.py:11-37, 40-52)
# --------------------------------------------------------------------------------------
def positional_encoding(x: Tensor, n_freqs: int) -> Tensor:
    """[M,3] -> [M, 3 + 6 L]: x, then per octave sin(2^k x), cos(2^k x) as 3-wide blocks."""
    out = [x]
    for k in range(n_freqs):
        f = float(2.0 ** k)
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    return torch.cat(out, dim=-1)


# --------------------------------------------------------------------------------------
# a4 / a10: the two MLPs, eval mode  (vector_field_network.py:177-208,
#           rendering_network.py:62-108)
# --------------------------------------------------------------------------------------
def _n_layers(sd: Dict[str, Tensor]) -> int:
    idx = {int(k.split('.')[1]) for k in sd if k.startswith('layers.')}
    return max(idx) + 1


def _layer_eval(sd: Dict[str, Tensor], i: int, x: Tensor, train: bool = False) -> Tensor:
    """Linear (+ BatchNorm1d, eps 1e-5) of layer i.  Eval mode normalises with the running statistics.  ``train`` is
    nn.BatchNorm1d in training mode (the state of the modules after ``VectorFieldNerf.train()``,
    vector_field_nerf.py:139-150): the batch mean and the biased batch variance over the rows of this call normalise, and
    the running statistics in ``sd`` are updated IN PLACE with momentum 0.1 (the unbiased variance goes into
    running_var) and ``num_batches_tracked`` advances, as torch does."""
    if f'layers.{i}.0.weight' in sd:  # nn.Sequential(Linear, BatchNorm1d)
        y = F.linear(x, sd[f'layers.{i}.0.weight'], sd[f'layers.{i}.0.bias'])
        if f'layers.{i}.1.running_mean' in sd:
            if train:      # y_hat = (y - mean_batch) / sqrt(var_batch_biased + eps); running stats <- 0.9 old + 0.1 batch (unbiased var)
                y = F.batch_norm(y, sd[f'layers.{i}.1.running_mean'], sd[f'layers.{i}.1.running_var'],
                                 sd[f'layers.{i}.1.weight'], sd[f'layers.{i}.1.bias'], True, 0.1, 1e-5)
                if f'layers.{i}.1.num_batches_tracked' in sd:
                    sd[f'layers.{i}.1.num_batches_tracked'] += 1
            else:
                y = F.batch_norm(y, sd[f'layers.{i}.1.running_mean'], sd[f'layers.{i}.1.running_var'],
                                 sd[f'layers.{i}.1.weight'], sd[f'layers.{i}.1.bias'], False, 0.0, 1e-5)
        return y
    return F.linear(x, sd[f'layers.{i}.weight'], sd[f'layers.{i}.bias'])


def _relu(x: Tensor, masks: Optional[list]) -> Tensor:
    """ReLU; with ``masks`` (a list consumed front to back) the unit is open where the given boolean mask says so
    instead of where x > 0.  Tests pass the masks of the implementation under test: a pre-activation that is ~1e-7 from
    zero may legitimately land on either side in two fp32 implementations, and with the masks pinned the gradients of
    both can be compared tightly."""
    if masks is None:
        return torch.relu(x)
    return x * masks.pop(0).to(x.dtype)


def vf_mlp(points: Tensor, sd: Dict[str, Tensor], multires: int = 6, skip_in=(4,), hidden: Optional[list] = None,
           masks: Optional[list] = None, train: bool = False) -> Tensor:
    """[M,3] -> [M, 3 + F] (cols 0:3 vector after tanh, 3: features after tanh).
    vector_field_network.py:177-208: skip layers see cat([x, pe]) / sqrt(2); ReLU between
    layers, tanh on the last.  ``hidden`` (a list) receives every post-ReLU activation (tests use it to
    detect ReLU-kink flips between two fp32 implementations); ``masks``: see _relu."""
    pe = positional_encoding(points, multires) if multires > 0 else points
    n = _n_layers(sd)
    x = pe.clone()
    inv = torch.sqrt(torch.tensor([2.0]))
    for i in range(n):
        if i in skip_in:
            x = torch.cat([x, pe], 1) / inv
        x = _layer_eval(sd, i, x, train)
        x = _relu(x, masks) if i < n - 1 else torch.tanh(x)
        if hidden is not None and i < n - 1:
            hidden.append(x.detach())
    return x


def vf_mlp_train(points: Tensor, sd: Dict[str, Tensor], multires: int = 6, skip_in=(4,), masks: Optional[list] = None,
                 hidden: Optional[list] = None) -> Tensor:
    """Train-mode forward of the VF net (vector_field_network.py:146-173): [M,3] -> [M, 3 + F + 9].  Batch-statistics
    BatchNorm, then one ``autograd.grad`` per vector column with ``grad_outputs = 1`` for every row, concatenated as
    [d y0 / d p, d y1 / d p, d y2 / d p].  With batch statistics the rows are coupled, so row m of that "Jacobian" is
    sum_r d y_c[r] / d p[m] — the backward of train-mode BatchNorm, not the per-row Jacobian of an eval-mode net.
    ``points`` gets requires_grad in place, as the reference does; the running statistics in ``sd`` advance once."""
    with torch.enable_grad():
        points.requires_grad_(True)
        y = vf_mlp(points, sd, multires, skip_in, masks=masks, hidden=hidden, train=True)
        ones = torch.ones_like(y[:, 0])
        rows = [torch.autograd.grad(y[:, c], points, ones, create_graph=True, retain_graph=True)[0] for c in range(3)]
        return torch.cat([y] + rows, dim=-1)


def render_mlp(points: Tensor, normals: Tensor, view_dirs: Tensor, feats: Tensor,
               sd: Dict[str, Tensor], multires: int = 4, hidden: Optional[list] = None, masks: Optional[list] = None,
               train: bool = False) -> Tensor:
    """mode 'idr' (rendering_network.py:84-86): cat[p, PE(d), n, feat] -> ReLU MLP -> sigmoid."""
    d = positional_encoding(view_dirs, multires) if multires > 0 else view_dirs
    x = torch.cat([points, d, normals, feats], dim=-1)
    n = _n_layers(sd)
    for i in range(n):
        x = _layer_eval(sd, i, x, train)
        if i < n - 1:
            x = _relu(x, masks)
            if hidden is not None:
                hidden.append(x.detach())
    return torch.sigmoid(x)


# --------------------------------------------------------------------------------------
# a5: windowed cosine similarity  (models/helpers/functions.py:41-72)
# --------------------------------------------------------------------------------------
def window_cosine(normals: Tensor, weights: Tensor) -> Tensor:
    """normals[N,S,3], weights[W] -> [N,S-1].  Entry j is cos(n_j, n_{j+1}); for the
    interior j in [start, S-1-start) it becomes the weighted sum over n_{j+1..j+start-1}
    (forward) and n_{j-1..j-(start-2)} (backward), weights normalised by sum |w|."""
    w_n = weights.shape[0]
    start = int((w_n + 1) / 2 + 1)
    middle = int((w_n - 1) / 2)
    norm = torch.tensor(0.0)
    for i in range(w_n):
        norm = norm + weights[i].abs()
    x, y = normals[:, :-1, :], normals[:, 1:, :]
    c = F.cosine_similarity(x, y, dim=2)
    out = c.clone()
    lo, hi = start, c.shape[1] - start
    if hi > lo:
        acc = c[:, lo:hi] * weights[middle] / norm
        for i in range(1, start - 1):
            fwd = F.cosine_similarity(x[:, lo:hi], y[:, lo + i:hi + i], dim=2)
            bwd = F.cosine_similarity(x[:, lo:hi], y[:, lo - i - 1:hi - i - 1], dim=2)
            acc = acc + fwd * weights[middle + i].abs() / norm + bwd * weights[middle - i].abs() / norm
        out[:, lo:hi] = acc
    return out


# --------------------------------------------------------------------------------------
# a7: Laplace density  (models/helpers/density_functions.py:20-34, 129-204)
# --------------------------------------------------------------------------------------
@dataclass
class DensityParams:
    beta: float = 0.5
    mean: float = 0.7
    scale: float = 100.0
    beta_bounds: tuple = (1e-4, 1e9)
    mean_bounds: tuple = (0.6, 1.0)
    scale_min: float = 1.0
    # Density.forward drops ``cutoff`` (density_functions.py:34, Q5): always -0.5.
    cutoff: float = -0.5

    def tensors(self):
        return (torch.tensor(self.beta), torch.tensor(self.mean), torch.tensor(self.scale))


def laplace_cdf(x: Tensor, beta: Tensor, scale: Tensor, mean: Tensor) -> Tensor:
    """density_functions.py:153-167."""
    return scale * (0.5 + 0.5 * torch.sign(x - mean) * (1 - torch.exp(-torch.abs(x - mean) / beta)))


def laplace_density(x: Tensor, p: DensityParams, beta=None, mean=None, scale=None) -> Tensor:
    """relu(cdf(x) - cdf(cutoff)) with clamped parameters (density_functions.py:129-204).
    beta/mean/scale may be passed as (autograd) tensors, else taken from ``p``."""
    b0, m0, s0 = p.tensors()
    beta = b0 if beta is None else beta
    mean = m0 if mean is None else mean
    scale = s0 if scale is None else scale
    beta = torch.clamp(beta, torch.tensor(p.beta_bounds[0]), torch.tensor(p.beta_bounds[1]))
    mean = torch.clamp(mean, torch.tensor(p.mean_bounds[0]), torch.tensor(p.mean_bounds[1]))
    scale = torch.max(scale.abs(), torch.tensor(p.scale_min))
    return torch.relu(laplace_cdf(x, beta, scale, mean) - laplace_cdf(torch.tensor([p.cutoff]), beta, scale, mean))


# --------------------------------------------------------------------------------------
# a6: density along a ray  (models/nerf/vector_field_nerf.py:442-474)
# --------------------------------------------------------------------------------------
def ray_density(normals: Tensor, ray_dirs: Tensor, n_window: int, dir_to_normal_th: float,
                p: DensityParams, beta=None, mean=None, scale=None, return_parts: bool = False):
    """normals[N,S,3], ray_dirs[N,3] (unit, one per ray) -> sigma[N,S] (last column 0).
    Uniform window weights ones/W are used whatever the annealed config says (Q6)."""
    n, s, _ = normals.shape
    w = torch.ones(n_window) / n_window
    c = window_cosine(normals, w)
    rd = ray_dirs.unsqueeze(1).expand(n, s, 3)
    c_ray = F.cosine_similarity(normals[:, :-1, :], rd[:, :-1, :], dim=2)
    masked = torch.logical_and(c_ray < dir_to_normal_th, c < 0)
    sigma = laplace_density(-c.reshape(-1, 1), p, beta, mean, scale).reshape(n, s - 1)
    sigma = torch.where(masked, torch.zeros_like(sigma), sigma)
    sigma = torch.cat([sigma, torch.zeros(n, 1)], dim=-1)
    if return_parts:
        return sigma, c, c_ray
    return sigma


# --------------------------------------------------------------------------------------
# a8: VolSDF weights  (utils/rendering.py:122-148)
# --------------------------------------------------------------------------------------
def volsdf_weights(z: Tensor, sigma: Tensor, normalize: bool = True) -> Tensor:
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.full((z.shape[0], 1), 1e10)], dim=-1)
    e = dists * sigma
    shifted = torch.cat([torch.zeros(z.shape[0], 1), e[:, :-1]], dim=-1)
    transmittance = torch.exp(-torch.cumsum(shifted, dim=-1))
    w = (1.0 - torch.exp(-e)) * transmittance
    if normalize:
        w = w / (w.sum(dim=-1, keepdim=True) + 1e-5)
    return w


# --------------------------------------------------------------------------------------
# a9: range fine sampler  (models/samplers/ray_sampler.py:264-302)
# --------------------------------------------------------------------------------------
def range_fine_z_vals(z_c: Tensor, w_c: Tensor, n_fine: int, near: float, far: float, half_range: float,
                      u_add: Tensor, u_fine: Optional[Tensor] = None):
    """-> (z[N,S_c+N_f] sorted, argmax[N] int64).  u_add = the always-drawn rand(N,N_f)
    (Q9); u_fine = the stratification draw (None when deterministic).  Rays whose argmax
    is 0 take the uniform u_add samples, the others a window of +-range (not clamped)."""
    imax = torch.argmax(w_c, dim=-1)
    z_star = z_c[torch.arange(z_c.shape[0]), imax]
    window = z_star[:, None] - half_range + 2 * half_range / (n_fine - 1) * torch.arange(n_fine)
    if u_fine is not None:
        window = stratify(window, u_fine)
    z_add = u_add * (far - near) + near
    pick = (imax > 0).unsqueeze(1)
    extra = torch.where(pick, window, z_add)
    z = torch.sort(torch.cat([z_c, extra], dim=-1), dim=-1)[0]
    return z, imax


# --------------------------------------------------------------------------------------
# a12: render()  (models/nerf/vector_field_nerf.py:216-338), eval-mode networks
# --------------------------------------------------------------------------------------
@dataclass
class RenderSettings:
    n_samples: int = 64
    n_fine: int = 64            # min(fine_sampler.N_samples, max_samples) at call time (Q16)
    near: float = 0.0
    far: float = 1.0
    fine_range: float = 0.3
    perturb: bool = False
    numerical_jacobian: bool = False   # vector_field_nerf.py:258-262,299-301
    train_mode: bool = False           # networks after VectorFieldNerf.train(): batch-statistics BatchNorm, Jacobian columns
    n_window: int = 11
    dir_to_normal_th: float = -2.0
    normalize: bool = True
    vf_multires: int = 6
    vf_skip_in: tuple = (4,)
    render_multires: int = 4
    feature_dims: int = 256
    detach_normals: bool = True        # RenderingNetConfig.detach_normals (rendering_network.py:76-77); the shipped conf sets True
    density: DensityParams = field(default_factory=DensityParams)


def render(uv: Tensor, pose: Tensor, intrinsics: Tensor, vf_sd: Dict[str, Tensor], rn_sd: Dict[str, Tensor],
           cfg: RenderSettings, u_coarse: Optional[Tensor] = None, u_fine: Optional[Tensor] = None,
           u_add: Optional[Tensor] = None, far=None, beta=None, mean=None, scale=None,
           hidden: Optional[list] = None, masks: Optional[list] = None) -> Dict[str, Tensor]:
    """Full forward of the path; returns every stage so tests can compare stage-wise.
    Random draws are explicit inputs, in the order the reference draws them
    (ray_sampler.py:138, :287, :292).  The proposal ("coarse") pass only evaluates the VF
    net (vector_field_nerf.py:252-277); the returned rgb/depth come from the S_c+N_f pass
    (Q2)."""
    out: Dict[str, Tensor] = {}
    n = uv.shape[0]
    far_v = cfg.far if far is None else far
    directions, ray_dirs, cam_loc = ray_directions(uv, pose, intrinsics)
    out.update(directions=directions, ray_dirs=ray_dirs, cam_loc=cam_loc)

    z_c = uniform_z_vals(n, cfg.n_samples, cfg.near, far_v, u_coarse if cfg.perturb else None)
    pts_c = points_along_rays(cam_loc, directions, z_c)
    out.update(z_coarse=z_c, points_coarse=pts_c)

    vf_call = (lambda q, **kw: vf_mlp_train(q, vf_sd, cfg.vf_multires, cfg.vf_skip_in, masks=kw.get("masks"),
                                            hidden=kw.get("hidden"))) if cfg.train_mode \
        else (lambda q, **kw: vf_mlp(q, vf_sd, cfg.vf_multires, cfg.vf_skip_in, **kw))
    nf = 3 + cfg.feature_dims
    with torch.no_grad():
        vf_c = vf_call(pts_c.reshape(-1, 3))          # train mode: enable_grad inside, the result is used without grad
        nrm_c = vf_c[:, :3].reshape(n, cfg.n_samples, 3)
        sigma_c, cos_c, cosray_c = ray_density(nrm_c, ray_dirs, cfg.n_window, cfg.dir_to_normal_th, cfg.density,
                                               beta, mean, scale, return_parts=True)
        w_c = volsdf_weights(z_c, sigma_c, cfg.normalize)
        dd_c = None
        if cfg.numerical_jacobian:
            dd_c = numerical_directional_derivatives(pts_c.reshape(-1, 3), vf_c[:, :3], vf_call, fine=False).reshape(-1, 3)
        elif cfg.train_mode:       # vector_field_nerf.py:260-261
            dd_c = directional_derivatives(vf_c[:, :3], vf_c[:, nf:nf + 9]).reshape(-1, 3)
    out.update(normals_coarse=nrm_c, window_cos_coarse=cos_c, cos_ray_coarse=cosray_c,
               sigma_coarse=sigma_c, weights_coarse=w_c)

    far_f = cfg.far if far is None else far
    if u_add is None:
        raise ValueError('u_add (the rand(N,N_f) draw of ray_sampler.py:292) is required')
    z_f, imax = range_fine_z_vals(z_c, w_c, cfg.n_fine, cfg.near, far_f, cfg.fine_range, u_add,
                                  u_fine if cfg.perturb else None)
    pts_f = points_along_rays(cam_loc, directions, z_f)
    s_t = cfg.n_samples + cfg.n_fine
    out.update(max_indices=imax, z_vals=z_f, points=pts_f)

    vf_f = vf_call(pts_f.reshape(-1, 3), masks=masks, hidden=hidden) if cfg.train_mode else \
        vf_mlp(pts_f.reshape(-1, 3), vf_sd, cfg.vf_multires, cfg.vf_skip_in, hidden=hidden, masks=masks)
    nrm_flat = vf_f[:, :3]
    feats = vf_f[:, 3:3 + cfg.feature_dims]
    nrm_f = nrm_flat.reshape(n, s_t, 3)
    sigma_f, cos_f, cosray_f = ray_density(nrm_f, ray_dirs, cfg.n_window, cfg.dir_to_normal_th, cfg.density,
                                           beta, mean, scale, return_parts=True)
    w_f = volsdf_weights(z_f, sigma_f, cfg.normalize)
    rep_dirs = ray_dirs.unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
    colors = render_mlp(pts_f.reshape(-1, 3), nrm_flat.detach() if cfg.detach_normals else nrm_flat, rep_dirs, feats, rn_sd, cfg.render_multires, hidden=hidden,
                        masks=masks, train=cfg.train_mode)
    rgb = torch.sum(w_f.unsqueeze(-1) * colors.reshape(n, s_t, 3), dim=1)
    depth = torch.sum(w_f.unsqueeze(-1) * z_f.unsqueeze(-1), dim=1)
    if cfg.numerical_jacobian:
        dd_f = numerical_directional_derivatives(pts_f.reshape(-1, 3), nrm_flat, vf_call, fine=True).reshape(-1, 3)
        out["directional_derivatives"] = torch.cat([dd_c, dd_f], dim=0).norm(dim=-1)
    elif cfg.train_mode:           # vector_field_nerf.py:303-305: the fine-pass values are computed and dropped (Q10)
        out["directional_derivatives"] = torch.cat([dd_c, dd_c], dim=0).norm(dim=-1)
    out.update(vf_out=vf_f, normals=nrm_f, window_cos=cos_f, cos_ray=cosray_f, sigma=sigma_f, weights=w_f,
               colors=colors, rgb=rgb, depth=depth, ray_dirs_repeated=rep_dirs)
    return out


# --------------------------------------------------------------------------------------
# trainer-side restatements, PINNED by tests/golden/trainer_steps.npz — outputs of the reference's own
# VFLoss, SphereSampler, supervision helpers and VectorFieldNerfRunner.train_epoch (tests/golden/make_train_golden.py)
# --------------------------------------------------------------------------------------
@dataclass
class LossWeights:      # confs/vf_nerf.conf:82-96
    rgb: float = 2.0
    depth: float = 0.5
    unit_norm: float = 0.1
    supervision: float = 1.0
    norm_smaller_than_one: float = 0.1
    directional_derivatives: float = 0.0
    norm_smaller_than_one_start: int = 11000
    depth_loss_clamp: float = 0.5
    directional_derivatives_start: int = 100


LOSS_TERM_NAMES = ("rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss",
                   "directional_derivatives_loss")


def vf_loss_terms(rgb: Tensor, depth: Tensor, normals: Tensor, supervised: Tensor, rgb_gt: Tensor, depth_gt: Tensor,
                  supervised_gt: Tensor, w: LossWeights, epoch: int = 0, directional: Optional[Tensor] = None):
    """VFLoss.forward (models/losses/vf_loss.py:34-87) -> (total, the six terms in the order of its log dictionary).
    L1 rgb; L1 depth clamped per element at depth_loss_clamp (skipped when the batch has no depth); mean (|n| - 1)^2; MSE on
    the supervised normals (skipped when there are none); mean relu(|n| - 1)^2 from norm_smaller_than_one_start on; mean of
    the directional-derivative norms from directional_derivatives_start on when render() returned any."""
    zero = torch.tensor(0.0)
    rgb_l = F.l1_loss(rgb, rgb_gt)                                                                      # :45
    depth_l = F.l1_loss(depth, depth_gt, reduction="none").clamp(max=w.depth_loss_clamp).mean() \
        if depth_gt.nelement() > 0 else zero                                                            # :48-51
    flat = normals.reshape(-1, 3)
    unit_l = torch.mean((torch.norm(flat, dim=1) - 1) ** 2)                                             # :54
    sup_l = F.mse_loss(supervised, supervised_gt) if supervised.nelement() > 0 else zero                # :57-60
    small_l = torch.mean(torch.pow(F.relu(torch.norm(flat, dim=1) - 1), 2)) \
        if epoch >= w.norm_smaller_than_one_start else zero                                             # :63-66
    dd_l = zero
    if directional is not None and epoch >= w.directional_derivatives_start:                            # :69-71
        dd_l = torch.mean(directional)
    total = w.rgb * rgb_l + w.depth * depth_l + w.unit_norm * unit_l + w.supervision * sup_l + \
        w.norm_smaller_than_one * small_l + w.directional_derivatives * dd_l                            # :74-79
    return total, (rgb_l, depth_l, unit_l, sup_l, small_l, dd_l)


def vf_loss(rgb: Tensor, depth: Tensor, normals: Tensor, supervised: Tensor, rgb_gt: Tensor, depth_gt: Tensor,
            supervised_gt: Tensor, w: LossWeights, epoch: int = 0, directional: Optional[Tensor] = None) -> Tensor:
    return vf_loss_terms(rgb, depth, normals, supervised, rgb_gt, depth_gt, supervised_gt, w, epoch, directional)[0]


def border_indices_and_gt(points: Tensor, normals: Tensor, far: float, radius: float, centroid: Tensor):
    """get_border_indices_and_gt (models/helpers/functions.py:75-98): normals of the ray samples farther than far/2 - radius
    from the centroid, unit vectors from them to the centroid.  points / normals [N,S,3]."""
    keep = torch.norm(points - centroid, dim=2) > (far / 2 - radius)
    return normals.reshape(points.shape)[keep], F.normalize(centroid - points[keep], dim=1)


def center_indices_and_gt(points: Tensor, normals: Tensor, centroid: Tensor, radius: float):
    """get_center_indices_and_gt (models/helpers/functions.py:137-157): normals of the ray samples closer than radius to the
    centroid, unit vectors from the centroid to them."""
    keep = torch.norm(points - centroid, dim=2) < radius
    return normals.reshape(points.shape)[keep], F.normalize(points[keep] - centroid, dim=1)


def trainer_parameter_list(vf_params: Dict[str, Tensor], rn_params: Dict[str, Tensor], density_params: Dict[str, Tensor]):
    """VectorFieldNerf.parameters() (models/nerf/vector_field_nerf.py:127-137): VF, rendering, density, then the VF parameters
    AGAIN (fine_vector_field_network is the same object, Q4)."""
    vf = list(vf_params.values())
    return vf + list(rn_params.values()) + list(density_params.values()) + vf


def trainer_epoch(vf_sd: Dict[str, Tensor], rn_sd: Dict[str, Tensor], density: Dict[str, Tensor], param_names, batches, cfg: RenderSettings,
                  w: LossWeights, epoch: int, centroid: Tensor, border_radius: float, far: float, lr: float, lr_gamma: float,
                  clip_norm: float, on_step=None):
    """VectorFieldNerfRunner.train_epoch (train/vector_field_nerf_train.py:161-260) for the shipped branch (VF init not
    "center": border shell + centre ball supervision, :193-216) and regime (networks in eval mode, :140-141).

    vf_sd / rn_sd: state dicts whose entries named in ``param_names['vf'|'rn']`` are leaf tensors with requires_grad (updated
    in place), density: {'beta','mean','scale'} leaves.  batches[t]: uv, pose, intrinsics, rgb_gt, depth_gt, the three render
    draws (u_coarse, u_fine, u_add) and the two sphere samplers' numpy draws (border_draws, center_draws: columns phi,
    cos(theta), u).  Optimizer = torch.optim.Adam over the duplicated list with the per-entry loop (foreach=False — what the
    reference's torch 1.x did and what torch 2.x does on CPU tensors), ExponentialLR(gamma), clip_grad_norm_ with the per-entry
    loop.  Returns one record per step: total, terms, clip norm, lr."""
    vf_p = {k: vf_sd[k] for k in param_names["vf"]}
    rn_p = {k: rn_sd[k] for k in param_names["rn"]}
    plist = trainer_parameter_list(vf_p, rn_p, density)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")        # "duplicate parameters" — the point of the exercise
        opt = torch.optim.Adam(plist, lr=lr, weight_decay=0.0, foreach=False)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, lr_gamma)
    records = []
    for b in batches:
        out = render(b["uv"], b["pose"], b["intrinsics"], vf_sd, rn_sd, cfg, u_coarse=b.get("u_coarse"), u_fine=b.get("u_fine"),
                     u_add=b["u_add"], beta=density["beta"], mean=density["mean"], scale=density["scale"])        # :177
        n_sup = (out["points"].shape[0] * out["points"].shape[1]) // 10
        bp, b_gt = sphere_shell_points_from_draws(b["border_draws"], far - 5 * border_radius, far, centroid, inward=True)   # :196-200
        assert bp.shape[0] == n_sup
        sup = [vf_mlp(bp, vf_sd, cfg.vf_multires, cfg.vf_skip_in)[:, :3]]                                         # :201
        sup_gt = [b_gt]
        rc_n, rc_gt = center_indices_and_gt(out["points"], out["normals"], centroid, border_radius)               # :204-207
        cp, c_gt = sphere_shell_points_from_draws(b["center_draws"], 0.0, border_radius, centroid, inward=False)   # :208-211
        sup += [rc_n, vf_mlp(cp, vf_sd, cfg.vf_multires, cfg.vf_skip_in)[:, :3]]                                  # :213
        sup_gt += [rc_gt, c_gt]
        total, terms = vf_loss_terms(out["rgb"], out["depth"], out["normals"], torch.cat(sup, 0), b["rgb_gt"], b["depth_gt"],
                                     torch.cat(sup_gt, 0), w, epoch, out.get("directional_derivatives"))          # :233
        opt.zero_grad()                                                                                           # :251
        total.backward()
        norm = torch.nn.utils.clip_grad_norm_(plist, clip_norm, foreach=False)                                    # :255
        lr_now = opt.param_groups[0]["lr"]
        opt.step()                                                                                                # :259
        sched.step()                                                                                              # :260
        rec = {"loss": total.detach(), "terms": [t.detach() for t in terms], "clip_total_norm": norm.detach(), "lr": lr_now,
               "out": {k: v.detach() for k, v in out.items()}, "ray_center_normals": rc_n.detach(), "ray_center_gt": rc_gt,
               "border_points": bp, "border_gt": b_gt, "center_points": cp, "center_gt": c_gt}
        records.append(rec)
        if on_step is not None:
            on_step(len(records) - 1, rec)
    return records, opt


def psnr(a: Tensor, b: Tensor) -> float:
    """-10 log10(mean((a-b)^2))  (utils/utils.py:235-245)."""
    mse = torch.mean((a - b) ** 2).item()
    return float('inf') if mse == 0 else -10.0 * math.log10(mse)



# --------------------------------------------------------------------------------------
# directional derivatives  (models/nerf/vector_field_nerf.py:476-526; only reached with numerical_jacobian=True or a
# train-mode VF net — not in the shipped regime)
# --------------------------------------------------------------------------------------
def directional_derivatives(normals: Tensor, jac: Tensor) -> Tensor:
    """jac[M,3,3] (or [M,9]) applied to two unit tangents of every normal: t1 = normalize((n_y, -n_x, 0)),
    t2 = normalize(n x (n_y, -n_x, 0)) -> [M,2,3]  (vector_field_nerf.py:476-498)."""
    jac = jac.reshape(-1, 3, 3)
    n1 = torch.zeros_like(normals)
    n1[:, 0] = normals[:, 1]
    n1[:, 1] = -normals[:, 0]
    n2 = torch.cross(normals, n1, dim=-1)
    out = torch.zeros(normals.shape[0], 2, 3, dtype=normals.dtype)
    out[:, 0, :] = torch.bmm(jac, F.normalize(n1, dim=-1).unsqueeze(-1)).squeeze(-1)
    out[:, 1, :] = torch.bmm(jac, F.normalize(n2, dim=-1).unsqueeze(-1)).squeeze(-1)
    return out


def numerical_directional_derivatives(points: Tensor, normals: Tensor, vf, fine: bool, epsilon: float = 1e-5) -> Tensor:
    """Central differences of the 3 vector outputs (vector_field_nerf.py:500-526).  ``vf``: points[M,3] -> [M,>=3].
    The reference fills the Jacobian by COLUMNS in the proposal pass (J[:, :, i] = d f / d x_i) and by ROWS in the fine
    pass (J[:, i] = d f / d x_i, i.e. the transposed Jacobian) — kept."""
    jac = torch.zeros(points.shape[0], 3, 3, dtype=points.dtype)
    for i in range(3):
        pos, neg = points.clone(), points.clone()
        pos[:, i] += epsilon
        neg[:, i] -= epsilon
        col = (vf(pos)[:, :3] - vf(neg)[:, :3]) / (2.0 * epsilon)
        if fine:
            jac[:, i] = col
        else:
            jac[:, :, i] = col
    return directional_derivatives(normals, jac)


# --------------------------------------------------------------------------------------
# supervision points  (models/samplers/sampler.py:160-193, models/helpers/functions.py:100-135)
# --------------------------------------------------------------------------------------
def sphere_shell_points_from_draws(draws: Tensor, r_min: float, r_max: float, centroid: Tensor, inward: bool) -> Tuple[Tensor, Tensor]:
    """SphereSampler.sample (models/samplers/sampler.py:170-193) + sample_border_points / sample_center_points
    (models/helpers/functions.py:100-135) on the sampler's own numpy draws: draws[n,3] float64 = phi ~ U(0, 2 pi),
    cos(theta) ~ U(-1, 1), u ~ U(0, 1), in the order the sampler draws them.  float64 arithmetic, cast to float32 BEFORE the
    centroid is added (functions.py:111,128); gt = normalize(+-(p - c), eps 1e-12)."""
    d = draws.double()
    phi, cos_t, u = d[:, 0], d[:, 1], d[:, 2]
    theta = torch.arccos(cos_t)
    r = torch.pow(u, 1.0 / 3.0) * (r_max - r_min) + r_min           # np.cbrt
    local = torch.stack([r * torch.sin(theta) * torch.cos(phi), r * torch.sin(theta) * torch.sin(phi), r * torch.cos(theta)], dim=1)
    points = local.float() + centroid.float()
    vec = (centroid - points) if inward else (points - centroid)
    return points, F.normalize(vec, dim=1)


def sphere_shell_points(u: Tensor, r_min: float, r_max: float, centroid: Tensor, inward: bool) -> Tuple[Tensor, Tensor]:
    """The same on UNIT uniforms u[n,3] (phi = 2 pi u0, cos(theta) = 2 u1 - 1, radius draw u2): the form the device sampler
    (vfn_sample_sphere_shell) is replayed against."""
    u64 = u.double()
    draws = torch.stack([2.0 * math.pi * u64[:, 0], 2.0 * u64[:, 1] - 1.0, u64[:, 2]], dim=1)
    return sphere_shell_points_from_draws(draws, r_min, r_max, centroid, inward)


# --------------------------------------------------------------------------------------
# dense-grid stages of the mesh extraction (evaluation/utils/mc_utils.py:34-86,107-223;
# evaluation/utils/guassian_smoothing.py:9-97).  Direct gathers instead of the reference's conv3d chains.
# --------------------------------------------------------------------------------------
_CORNERS = ((0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0), (0, 0, 1), (0, 1, 1), (1, 1, 1), (1, 0, 1))   # selection-filter order


def _corner_gather(grid: Tensor, n: int) -> Tensor:
    """grid[n,n,n,C] -> [n,n,n,8,C]: the 8 corners of every cell, zero beyond the grid (conv3d padding=1, then [1:,1:,1:])."""
    padded = F.pad(grid, (0, 0, 0, 1, 0, 1, 0, 1))
    return torch.stack([padded[a:a + n, b:b + n, c:c + n] for a, b, c in _CORNERS], dim=3)


def grid_divergence(vt_values: Tensor, n: int, threshold: float = -0.5) -> Tensor:
    """mc_utils.py:34-86: flux of the normalised field through the 2x2x2 corners of every cell (outward unit diagonals),
    sum_c x_c |x_c| (sqrt(3)/4) / (sqrt(2)/3); 1 where <= threshold; the last planes stay 0."""
    v = F.normalize(vt_values, dim=1).reshape(n, n, n, 3)
    s = torch.zeros(n - 1, n - 1, n - 1, dtype=v.dtype)
    for c in range(8):
        a, b, cc = c >> 2, (c >> 1) & 1, c & 1
        d = F.normalize(torch.tensor([2.0 * a - 1, 2.0 * b - 1, 2.0 * cc - 1], dtype=v.dtype), dim=0)
        x = (v[a:a + n - 1, b:b + n - 1, cc:cc + n - 1] * d).sum(-1)
        s = s + x * x.abs() * (math.sqrt(3.0) / 4.0)
    s = s / (math.sqrt(2.0) / 3.0)
    out = torch.zeros(n, n, n, dtype=v.dtype)
    out[:-1, :-1, :-1] = s
    return torch.where(out > threshold, torch.zeros_like(out), torch.ones_like(out))


def smooth_field(vf: Tensor, k: int = 3, sigma: float = 1.0) -> Tensor:
    """guassian_smoothing.py:9-97: depthwise k^3 kernel, the product over the axes of exp(-((x - mean) / (2 sigma))^2)
    — the reference's exponent, i.e. a Gaussian of standard deviation sigma sqrt(2) — normalised to sum 1; replicate
    padding.  vf[n,n,n,3]."""
    ax = torch.arange(k, dtype=torch.float32)
    mean = (k - 1) / 2.0
    g = 1.0 / (sigma * math.sqrt(2 * math.pi)) * torch.exp(-(((ax - mean) / (2 * sigma)) ** 2))
    kern = g[:, None, None] * g[None, :, None] * g[None, None, :]
    kern = (kern / kern.sum()).to(vf.dtype)
    x = F.pad(vf.permute(3, 0, 1, 2).unsqueeze(0), (k // 2,) * 6, mode="replicate")
    w = kern.view(1, 1, k, k, k).repeat(3, 1, 1, 1, 1)
    return F.conv3d(x, w, groups=3).squeeze(0).permute(1, 2, 3, 0)


def grid_unify_direction(divergence_grid: Tensor, vt_grid: Tensor, n: int) -> Tensor:
    """mc_utils.py:107-167.  vt_grid[3,n,n,n] (normalised field).  For surface cells: the most opposed corner pair
    (argmax of 1 - <v_a, v_b> over the 64 ordered pairs, first maximum), then per corner the nearer of the two."""
    corners = _corner_gather(vt_grid.permute(1, 2, 3, 0), n)                      # [n,n,n,8,3]
    sel = divergence_grid == 1
    sv = corners[sel]                                                              # [m,8,3]
    dots = (sv[:, :, None, 0] * sv[:, None, :, 0] + sv[:, :, None, 1] * sv[:, None, :, 1]) + sv[:, :, None, 2] * sv[:, None, :, 2]
    far = torch.argmax((1.0 - dots).reshape(-1, 64), dim=-1)
    first = sv[torch.arange(sv.shape[0]), far // 8]
    second = sv[torch.arange(sv.shape[0]), far % 8]
    d1 = torch.norm(first[:, None, :] - sv, dim=-1)
    d2 = torch.norm(second[:, None, :] - sv, dim=-1)
    choice = torch.argmin(torch.stack((d1, d2), dim=-1), dim=-1)
    out = torch.zeros(n, n, n, 8, dtype=choice.dtype)
    out[sel] = choice
    return out.reshape(-1, 8)


def grid_comb_format(choice_side: Tensor, norms: Tensor, n: int):
    """mc_utils.py:170-223: the 28 corner pairs (a < b) of every cell."""
    nr = _corner_gather(norms.reshape(n, n, n, 1), n).reshape(n ** 3, 8)
    pairs = [(a, b) for a in range(7) for b in range(a + 1, 8)]
    different = torch.stack([(choice_side[:, a] != choice_side[:, b]).to(norms.dtype) for a, b in pairs], dim=1)
    pair_norms = torch.stack([torch.stack((nr[:, a], nr[:, b]), dim=-1) for a, b in pairs], dim=1)
    return different, pair_norms
