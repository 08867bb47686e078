"""The body of the reference trainer's loop as one callable.

``TrainStep`` is ``VectorFieldNerfRunner.train_epoch``'s per-batch body (reference ``train/vector_field_nerf_train.py:172-260``,
the branch every shipped scene takes: VF init not "center", border + centre supervision, eval-mode networks, :140-141) on the
device-side pieces of this package, in the reference's order:

    render(pose, pixels, intrinsics, epoch, white)                                      :177
    border shell points  -> vector_field_network(points)[:, :3]                         :196-202
    ray samples inside the centre ball + centre ball points -> vector_field_network     :203-214
    VFLoss(predictions, ground_truth, epoch)                                            :218-233
    optimizer.zero_grad(); loss.backward()                                              :251-252
    [one all-reduce of the flat gradient bucket, when there is more than one rank]      (replaces nn.DataParallel, :70-75)
    clip_grad_norm_(model.parameters(), clip_norm)   — the duplicated list, Q4          :254-255
    optimizer.step(); scheduler.step()                                                  :258-260

The reference's trainer object itself (datasets, wandb, checkpoints folders) stays the reference's; it runs unchanged on this
package through ``vf_nerf_amd.dropin``.  ``TrainStep`` is what ``bench.py``, ``tools/train_curve.py`` and the convergence
tests time and check, so that all three run the same step.

``TeacherTargets``: a learnable synthetic target for such runs — a pool of rays over a few orbit views
with rgb / depth rendered by a TEACHER model of a different weight seed.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch

from . import loss as vloss
from . import optim, supervision, synthetic

# confs/vf_nerf.conf:74-90
SHIPPED_LOSS_CONFIG = dict(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100)
SHIPPED_LOSS_WEIGHTS = dict(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0)


class TrainStep:
    def __init__(self, model, centroid, border_radius: float = 0.15, far: Optional[float] = None, criterion=None, bucket=None,
                 compact_selection: bool = False) -> None:
        self.model = model
        dev = model.config.cuda_config.device
        self.centroid = torch.as_tensor(centroid, dtype=torch.float32, device=dev)
        self.centroid_host = tuple(float(x) for x in (centroid.tolist() if isinstance(centroid, torch.Tensor) else centroid))
        self.radius = float(border_radius)
        self.far = float(model.ray_sampler.far if far is None else far)
        self.criterion = criterion or vloss.VFLoss(SimpleNamespace(**SHIPPED_LOSS_CONFIG), SimpleNamespace(**SHIPPED_LOSS_WEIGHTS))
        self.bucket = bucket                       # distributed.GradientBucket or None
        # The ray samples inside the centre ball are a data-dependent subset (functions.py:137-157: boolean-mask indexing, i.e. a
        # device synchronisation in the middle of the step while the host learns the row count).  False (default): the same loss
        # term without compaction — unselected rows enter with prediction = ground truth = 0 and the mean's denominator is the
        # selected-row count kept on the device (``supervised_rows``, an extension of this package's VFLoss): no synchronisation,
        # the same value and gradients up to the order of the sum.  True: the reference's compaction, sync included.
        self.compact_selection = bool(compact_selection)
        self.last_total_norm = None
        self.last_outputs = None
        self.last_colour_counts = None             # one-call step: device [samples the colour branch ran on, all samples]
        # the whole step as ONE C call (vfn_train_step) when the regime allows it (onecall.OneCallStep.applicable: the shipped one);
        # ``model.one_call_train_step = False`` keeps the launch-by-launch path below, which the tests hold equal to it
        from .onecall import OneCallStep
        self.one_call = OneCallStep(self)

    def __call__(self, pose, pixels, intrinsics, rgb_gt, depth_gt, epoch: int = 0, white: bool = False,
                 uniforms: Optional[Dict[str, torch.Tensor]] = None) -> Tuple[torch.Tensor, Dict[str, float]]:
        model, dev = self.model, self.centroid.device
        cfg = model.config
        if self.one_call.applicable(pose, white, pixels.shape[0]):
            return self.one_call.run(pose, pixels, intrinsics, rgb_gt, depth_gt, epoch, uniforms)
        outputs = model.render(pose, pixels, intrinsics, epoch, white, uniforms=uniforms)
        n_sup = (outputs.points_coarse.shape[0] * outputs.points_coarse.shape[1]) // 10
        fused = bool(getattr(self.criterion, "fused", False)) and not self.compact_selection and outputs.coarse_normals.is_cuda
        sup, sup_gt = [], []
        rows = None                                # number of supervised rows as a device scalar (dense selection without the fused loss)
        ray_center = None
        # The reference evaluates the vector-field net on the border points and on the centre points in two calls of the FULL forward and
        # keeps the three vector columns (`vector_field_network(points)[:, :3]`, :201,213).  Same values from ONE vector-only call on both
        # batches (the forward is pointwise; the feature block — 12.5 % of the MACs — is never read): one launch forward and one dX chain
        # instead of two each, which at the reference's 1 024-ray batches are 0.4-round launches that leave the chip mostly idle.
        pts, gts = [], []
        if cfg.border_supervision:
            bp, b_gt = supervision.sample_border_points(self.far - 5 * self.radius, self.far, n_sup, self.centroid, dev)
            pts.append(bp)
            gts.append(b_gt)
        if cfg.center_supervision:
            cp, c_gt = supervision.sample_center_points(self.centroid, self.radius, n_sup, dev)
            pts.append(cp)
            gts.append(c_gt)
            if fused:          # selected, counted and differentiated inside the loss kernels (csrc/vfn_loss.hip): nothing to build here
                ray_center = (outputs.points_coarse, self.centroid_host, self.radius)
            elif self.compact_selection:
                rc_n, rc_gt = supervision.get_center_indices_and_gt(outputs.points_coarse, outputs.coarse_normals, self.centroid, self.radius)
                sup.append(rc_n)
                sup_gt.append(rc_gt)
            else:
                rc_n, rc_gt, n_sel = supervision.center_rows_dense(outputs.points_coarse, outputs.coarse_normals, self.centroid, self.radius)
                rows = n_sel + float(n_sup * len(pts))
                sup.append(rc_n)
                sup_gt.append(rc_gt)
        if pts:
            net = model.vector_field_network
            both = torch.cat(pts) if len(pts) > 1 else pts[0]
            if not net.training:
                sup.append(net(both, vector_only=True))
            else:
                # training mode: BatchNorm normalises every batch with ITS statistics and advances the running ones once per call, so
                # the two batches go through two calls exactly as the reference makes them (:201,213).  The nine Jacobian columns the
                # reference's training-mode forward appends are sliced away there (`[:, :3]`); they change no state, so they are not
                # computed here (three backward passes through the batch statistics per call: ~10 % of a training-mode step).
                sup.append(torch.cat([net(p, jacobian=False)[:, :3] for p in pts]) if len(pts) > 1 else net(pts[0], jacobian=False)[:, :3])
            sup_gt.append(torch.cat(gts) if len(gts) > 1 else gts[0])
        predictions = {"rgb": outputs.coarse_rgb_values, "depth": outputs.coarse_depth_map,
                       "normals": outputs.coarse_normals.reshape(-1, 3),
                       "directional_derivatives": outputs.directional_derivtives}
        ground_truth = {"rgb": rgb_gt.reshape(-1, 3), "depth": depth_gt}
        if fused:
            predictions["supervised_segments"] = list(zip(sup, sup_gt))
            predictions["supervised_normals"] = ground_truth["supervised_normals"] = torch.empty(0, 3, device=dev)
            if ray_center is not None:
                predictions["ray_center"] = ray_center
        else:
            predictions["supervised_normals"] = torch.cat(sup, dim=0) if sup else torch.empty(0, 3, device=dev)
            ground_truth["supervised_normals"] = torch.cat(sup_gt, dim=0) if sup_gt else torch.empty(0, device=dev)
            if rows is not None:
                predictions["supervised_rows"] = rows
        loss, terms = self.criterion(predictions, ground_truth, epoch)
        if self.bucket is not None:
            self.bucket.zero()
        else:
            model.optimizer.zero_grad()
        loss.backward()
        if self.bucket is not None:
            self.bucket.all_reduce_mean()
        # = torch's clip_grad_norm_(..., foreach=False) over the duplicated list: clipped once per occurrence (Q4)
        self.last_total_norm = optim.clip_grad_norm_(model.parameters(), cfg.scheduler_config.clip_norm)
        model.optimizer.step()
        model.scheduler.step()
        self.last_outputs = outputs
        return loss.detach(), terms


class TeacherTargets:
    """Rays of ``views`` orbit views of a ``width`` x ``height`` pinhole camera with rgb / depth rendered by ``teacher`` (exact-fp32
    kernels, deterministic sampling: a pixel's target does not depend on a draw).  ``batch(step, n)`` returns the same ``n``
    rays for the same ``step`` whatever model is being trained, so that two precision modes see identical data."""

    def __init__(self, teacher, views: int = 8, width: int = 64, height: int = 64, focal: float = 60.0, seed: int = 0,
                 chunk: int = 4096) -> None:
        dev = teacher.config.cuda_config.device
        uv, pose, K = [], [], []
        for i in range(views):
            u, p, k = synthetic.pinhole_image(width, height, focal, device=dev, pose=synthetic.orbit_pose(-35.0 + 10.0 * i, 5.0 + 2.0 * (i % 3), 0.9))
            uv.append(u)
            pose.append(p)
            K.append(k)
        self.uv, self.pose, self.K = torch.cat(uv), torch.cat(pose), torch.cat(K)
        keep = (teacher.precision, teacher.ray_sampler.deterministic, teacher.fine_sampler.deterministic)
        teacher.precision = "fp32"
        teacher.ray_sampler.deterministic = teacher.fine_sampler.deterministic = True
        rgb, depth = [], []
        with torch.no_grad():
            for lo in range(0, self.uv.shape[0], chunk):
                out = teacher.render(self.pose[lo:lo + chunk], self.uv[lo:lo + chunk], self.K[lo:lo + chunk], 0)
                rgb.append(out.coarse_rgb_values)
                depth.append(out.coarse_depth_map)
        teacher.precision, teacher.ray_sampler.deterministic, teacher.fine_sampler.deterministic = keep
        self.rgb, self.depth = torch.cat(rgb), torch.cat(depth)
        self.seed = int(seed)

    def __len__(self) -> int:
        return self.uv.shape[0]

    def batch(self, step: int, n: int):
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(step))
        idx = torch.randint(0, len(self), (n,), generator=g).to(self.uv.device)
        return self.pose[idx], self.uv[idx], self.K[idx], self.rgb[idx], self.depth[idx]

    @torch.no_grad()
    def psnr(self, model, chunk: int = 4096) -> float:
        """PSNR (utils/utils.py:235-245) of ``model``'s deterministic render of the whole pool against the teacher's."""
        keep = (model.ray_sampler.deterministic, model.fine_sampler.deterministic)
        model.ray_sampler.deterministic = model.fine_sampler.deterministic = True
        se = torch.zeros((), device=self.uv.device, dtype=torch.float64)
        for lo in range(0, len(self), chunk):
            out = model.render(self.pose[lo:lo + chunk], self.uv[lo:lo + chunk], self.K[lo:lo + chunk], 0)
            se += ((out.coarse_rgb_values - self.rgb[lo:lo + chunk]).double() ** 2).sum()
        model.ray_sampler.deterministic, model.fine_sampler.deterministic = keep
        return float(-10.0 * torch.log10(se / (len(self) * 3)))
