"""The reference trainer's loop body, CALL FOR CALL (train/vector_field_nerf_train.py:172-275, the branch every shipped scene takes: VF init
not "center", border + centre supervision, eval-mode networks), on the names ``vf_nerf_amd.dropin`` installs under the reference's import
paths — test / bench infrastructure: the reference itself is not importable on the GPU box, and its trainer object needs datasets, a config
tree and wandb around it.  What the trainer calls through ``functions.*`` is ``vf_nerf_amd.supervision`` here (dropin.install puts exactly
those functions into ``models.helpers.functions``), ``self.loss`` is ``vf_nerf_amd.loss.VFLoss`` (installed as ``models.losses.vf_loss.VFLoss``),
``torch.nn.utils.clip_grad_norm_`` is the wrapped one, and the dataset is a stand-in with the reference datasets' three getters.

    step = ReferenceLoop(model, criterion, dataset, border_radius, clip_norm)
    loss_value = step(train_data, epoch)          # one iteration of ``for i, train_data in enumerate(self.dataloader)``
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch

import vf_nerf_amd.dropin as dropin  # noqa: F401  (installs the aliases and the clip_grad_norm_ wrapper)
from vf_nerf_amd import supervision as functions


class StandInDataset:
    """The three getters train_epoch reads (datasets/normal_datasets/replica_dataset.py:214-233, base_dataset.py:108-127)."""

    white_bkgd = False

    def __init__(self, centroid, far: float, near: float = 0.0) -> None:
        self.gt_mesh_centroid = torch.as_tensor(centroid, dtype=torch.float32).cpu()
        self.near, self.far = float(near), float(far)

    def get_centroid(self, device) -> torch.Tensor:
        return self.gt_mesh_centroid.to(device)

    def get_bounds(self):
        return self.near, self.far

    def get_vf_init_method(self):
        return ("exterior", "")


dropin.cache_centroid(StandInDataset)          # what dropin.install() does to the reference's dataset classes


class ReferenceLoop:
    def __init__(self, model, criterion, dataset, border_radius: float, clip_norm: Optional[float] = None, sync_each_step: bool = True) -> None:
        self.model, self.loss, self.dataset = model, criterion, dataset
        self.border_radius = float(border_radius)
        self.clip_norm = float(model.config.scheduler_config.clip_norm if clip_norm is None else clip_norm)
        self.device = model.config.cuda_config.device
        # train.py:262-275 reads loss.item() and the six terms after every step (the running averages): a device synchronisation per step
        # that belongs to the reference's loop.  False: the same calls without those reads (what a loop that logs every k steps does).
        self.sync_each_step = bool(sync_each_step)
        self.average_losses: Optional[Dict[str, float]] = None
        self.last_outputs = None
        self.last_total_norm = None
        # parity replays only: the three torch.rand draws of render() in the reference's call order (the reference has no such argument)
        self.render_uniforms: Optional[Dict[str, torch.Tensor]] = None
        # host-time accounting (tools/host_profile.py): {stage: seconds} accumulated over the calls when a dict is put here
        self.stage_seconds: Optional[Dict[str, float]] = None

    def _mark(self, stage: str) -> None:
        acc = self.stage_seconds
        if acc is not None:
            now = time.perf_counter()
            acc[stage] = acc.get(stage, 0.0) + now - self._t
            self._t = now

    def __call__(self, train_data: Dict[str, torch.Tensor], epoch: int):
        model, dataset, device, radius = self.model, self.dataset, self.device, self.border_radius
        cfg = model.config
        self._t = time.perf_counter()
        pixels = train_data["uv"].squeeze(0).to(device)                                      # :175-177
        intrinsics = train_data["intrinsics"].squeeze(0).to(device)
        pose = train_data["pose"].squeeze(0).to(device)
        if self.render_uniforms is None:
            outputs = model.render(pose, pixels, intrinsics, epoch, dataset.white_bkgd)      # :180
        else:
            outputs = model.render(pose, pixels, intrinsics, epoch, dataset.white_bkgd, uniforms=self.render_uniforms)
        self._mark("render")
        n_points = (outputs.points_coarse.shape[0] * outputs.points_coarse.shape[1]) // 10
        supervised_normals = torch.empty(0, 3).to(device)                                    # :196-197
        gt_normals = torch.empty(0).to(device)
        if cfg.border_supervision:                                                           # :198-205
            border_points, border_gt_normals = functions.sample_border_points(dataset.get_bounds()[1] - 5 * radius, dataset.get_bounds()[1], n_points,
                                                                              dataset.get_centroid(device), outputs.points_coarse.device)
            supervised_normals = torch.cat([supervised_normals, model.vector_field_network(border_points)[:, :3]], dim=0)
            gt_normals = torch.cat([gt_normals, border_gt_normals], dim=0)
        if cfg.center_supervision:                                                           # :206-217
            ray_center_normals, ray_center_gt_normals = functions.get_center_indices_and_gt(outputs.points_coarse, outputs.coarse_normals,
                                                                                            dataset.get_centroid(device), radius)
            center_points, center_gt_normals = functions.sample_center_points(dataset.get_centroid(device), radius, n_points, outputs.points_coarse.device)
            supervised_normals = torch.cat([supervised_normals, ray_center_normals, model.vector_field_network(center_points)[:, :3]], dim=0)
            gt_normals = torch.cat([gt_normals, ray_center_gt_normals, center_gt_normals], dim=0)
        self._mark("supervision")
        predictions = {"rgb": outputs.coarse_rgb_values, "depth": outputs.coarse_depth_map, "normals": outputs.coarse_normals.reshape(-1, 3),
                       "supervised_normals": supervised_normals, "directional_derivatives": outputs.directional_derivtives}            # :220-226
        ground_truth = {"rgb": train_data["rgb"].reshape(-1, 3).to(device), "depth": train_data["depth"].squeeze(0).to(device),
                        "supervised_normals": gt_normals}
        loss, losses_dict = self.loss(predictions, ground_truth, epoch)                      # :233
        self._mark("loss")
        total_loss = loss                                                                    # :236-249: fine_normals is None
        model.optimizer.zero_grad()                                                          # :251-252
        self._mark("zero_grad")
        total_loss.backward()
        self._mark("backward")
        self.last_total_norm = torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip_norm)      # :254-255
        self._mark("clip")
        model.optimizer.step()                                                               # :258-260
        self._mark("optimizer.step")
        model.scheduler.step()
        self._mark("scheduler.step")
        self.last_outputs = outputs
        if self.sync_each_step:                                                              # :262-275
            if self.average_losses is None:
                self.average_losses = losses_dict
                self.average_losses["loss"] = loss.item()
            else:
                self.average_losses["loss"] += loss.item()
                for key in losses_dict.keys():
                    self.average_losses[key] += losses_dict[key]
            self._mark("loss.item()")
        return loss, losses_dict


def reference_render_view(model, all_pose, all_pixels, all_intrinsics, image_size, epoch: int, split_size: int = 512, device="cuda", white: bool = False):
    """The per-image body of the reference evaluator's loop, CALL FOR CALL (evaluation/methods.py:507-540): the view split into
    ``split_size``-ray chunks, each uploaded, rendered with ``model.render`` and pulled back with ``.cpu().numpy()`` (rgb, depth, and the
    pixel indices four times) — what runs when ``vf_nerf_amd.dropin.install(patch_evaluator=False)`` leaves ``render_images`` alone.
    -> (rgb[H,W,3], depth[H,W,1]) numpy arrays."""
    import numpy as np
    num_pixels = all_pixels.shape[0]
    num_batches = int(np.ceil(num_pixels / split_size))
    pixels_split = torch.split(all_pixels, split_size, dim=0)
    pose_split = torch.split(all_pose, split_size, dim=0)
    intrinsics_split = torch.split(all_intrinsics, split_size, dim=0)
    rgb = np.zeros((image_size[0], image_size[1], 3))
    depth_map = np.zeros((image_size[0], image_size[1], 1))
    with torch.no_grad():
        for j in range(num_batches):
            pixels = pixels_split[j].to(device)
            pose = pose_split[j].to(device)
            intrinsics = intrinsics_split[j].to(device)
            output = model.render(pose, pixels, intrinsics, epoch, white)
            rgb_values = output.coarse_rgb_values.cpu().numpy()
            predicted_depth = output.coarse_depth_map.cpu().numpy()
            rgb[pixels[:, 1].long().cpu().numpy(), pixels[:, 0].long().cpu().numpy(), :] = rgb_values
            depth_map[pixels[:, 1].long().cpu().numpy(), pixels[:, 0].long().cpu().numpy(), :] = predicted_depth
    return rgb, depth_map
