"""The evaluator's image loop on the HIP path (reference: ``evaluation/methods.py:472-545``, ``render_images``).

The reference renders a view ``split_size`` rays at a time and, per chunk, uploads three tensors, calls ``model.render`` and
pulls rgb and depth (and the pixel indices, four times) back with ``.cpu()`` — four device synchronisations per 512..1024-ray
chunk, with the device idle in between.  ``render_view`` does the same work with consecutive chunks on alternating HIP streams
(``VectorFieldNerf.render_chunked``: same per-chunk arithmetic, so the same values) and ONE download per image;
``render_images`` is the reference function's loop around it — same dataset, same files written — and is what
``vf_nerf_amd.dropin.install()`` puts in place of ``evaluation.methods.render_images``, so that ``evaluation/evaluate.py`` runs
unchanged."""
from __future__ import annotations

import importlib
import os
from typing import Tuple

import numpy as np
import torch


MIN_CHUNK = 8192        # rays per launch group of a whole view (see render_view)
# render_view keeps rgb and depth only (as the reference's loop does, evaluation/methods.py:528-540), so it asks render() for SPARSE
# colours: the rendering net is evaluated only for the samples whose weight is non-zero — a few percent — which leaves rgb and depth
# bit-identical (model.sparse_colours, include/vfn.h vfn_render_params.sparse_colours).  False: the dense plan.
SPARSE_COLOURS = True


@torch.no_grad()
def render_view(model, all_pose: torch.Tensor, all_pixels: torch.Tensor, all_intrinsics: torch.Tensor, epoch: int,
                split_size: int = 512, white: bool = False, n_streams: int = 2, min_chunk: int = MIN_CHUNK) -> Tuple[np.ndarray, np.ndarray]:
    """One view, as the dataset hands it over (host or device tensors, pose / intrinsics replicated per ray or shared) ->
    (rgb[N,3], depth[N,1]) numpy arrays.

    ``split_size`` is the reference loop's chunk (methods.py:513-545), there to bound ITS activation memory: its un-fused
    forward materialises [rays x samples, 256] per layer.  Here nothing of the kind exists (the fused kernels keep activations on
    chip; a chunk's outputs are 52 B per sample), and rays are independent end to end, so the view is rendered in chunks of
    max(split_size, ``min_chunk``) rays: at the shipped sampler sizes (100 + 35 samples) a 512-ray chunk is 1.6 + 0.5 rounds of
    workgroups per fused launch — 1.32 M rays/s against 2.06 M in 8 192-ray chunks on one MI355X.  Every ray's value is the one
    the 512-ray loop computes for the same random draws; which draws a ray gets (stratified jitter, the uniform fine samples of
    rays without a surface, Q9) depends on its position in the stream as it does in the reference.  ``min_chunk = 0``: exactly the
    reference's chunking."""
    chunk = max(int(split_size), int(min_chunk))
    keep = getattr(model, "sparse_colours", False)
    model.sparse_colours = bool(SPARSE_COLOURS) or keep
    try:
        rgb, depth = model.render_chunked(all_pose, all_pixels, all_intrinsics, epoch, chunk=chunk, n_streams=n_streams, white=white)
    finally:
        model.sparse_colours = keep
    host = torch.empty(rgb.shape[0], 4, pin_memory=True)
    host[:, :3].copy_(rgb, non_blocking=True)
    host[:, 3:].copy_(depth, non_blocking=True)
    torch.cuda.current_stream(rgb.device).synchronize()
    out = host.numpy()
    return out[:, :3].copy(), out[:, 3:].copy()


@torch.no_grad()
def render_images(model, eval_path: str, dataset_config, epoch: int, split_size: int = 512, device: torch.device = torch.device("cuda")) -> None:
    """Drop-in for ``evaluation.methods.render_images`` (methods.py:472-545): every image of the dataset rendered with all of its
    pixels and written as ``rendered_images/image-{i}.png`` / ``depth-{i}`` through the reference's own dataset and ``utils``
    helpers (they stay the reference's: datasets and image I/O are outside the hot path)."""
    dataset_dict = importlib.import_module("datasets.normal_datasets").dataset_dict
    utils = importlib.import_module("utils.utils")
    dataset = dataset_dict[dataset_config.dataset_name](dataset_config)
    dataset.all_pixels = True
    model.ray_sampler.near, model.ray_sampler.far = dataset.get_bounds()
    if model.config.ray_sampler_config.fine_sampling():
        model.fine_sampler.near, model.fine_sampler.far = dataset.get_bounds()
    dataloader = torch.utils.data.DataLoader(dataset, batch_size=1, shuffle=False)
    path = os.path.join(eval_path, "rendered_images")
    utils.mkdir_ifnotexists(path)
    for i, batch in enumerate(dataloader):
        pixels = batch["uv"].squeeze(0)
        rgb_values, depth_values = render_view(model, batch["pose"].squeeze(0), pixels, batch["intrinsics"].squeeze(0), epoch, split_size,
                                               dataset.white_bkgd)
        rows, cols = pixels[:, 1].long().cpu().numpy(), pixels[:, 0].long().cpu().numpy()
        rgb = np.zeros((dataset.image_size[0], dataset.image_size[1], 3))
        depth_map = np.zeros((dataset.image_size[0], dataset.image_size[1], 1))
        rgb[rows, cols, :] = rgb_values
        depth_map[rows, cols, :] = depth_values
        utils.save_rgb(os.path.join(path, f"image-{i}.png"), rgb)
        utils.save_depth(os.path.join(path, f"depth-{i}"), depth_map)
