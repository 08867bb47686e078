#!/usr/bin/env python3
"""Headline benchmark: rays/s of ``VectorFieldNerf.render`` (forward, eval mode) on 4096-ray chunks with
128 samples per ray (S_c = N_f = 64), synthetic random-weight scene, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 3

A step = one render() of one 4096-ray chunk per rank (rays shard embarrassingly: no data-path collective,
weak scaling).  The timed region is bracketed by barrier + torch.cuda.synchronize() on both sides; the time is
the max over ranks; rank 0 prints ONE JSON line.  Inputs (uv / pose / intrinsics) are resident in HBM before
the timed region; random numbers come from the device Philox stream inside the timed region.

Extra objects on the line:
  roofline          the dominant kernel class of the step (largest share of the timed region; with the defaults the VF
                    MLP launch of the split inference pipeline, two launches per step): ALGORITHMIC fp32-equivalent FLOPs
                    of one launch (SURVEY.md §8d per-point figures x its points) / its average duration measured with
                    HIP events on the launch stream inside the timed region; peak = dense f16 MFMA 2500 TFLOP/s / 3 (three
                    f16 products per fp32-equivalent product), or 157.3 TFLOP/s fp32 matrix with --precision fp32;
                    traffic from the committed PMC passes (profiles/).
  cpu_baseline      the CPU oracle (torch fp32, 32 host threads) on a bounded sample of the same workload.
  parity_vs_oracle  the "PSNR vs ref" half of the metric: the HIP path against the oracle on those sample rays.

Other workloads (not the headline line): --workload view | grid | train (BASELINE.json configs[1], [4], [2]);
--no-reuse evaluates the VF net on the proposal samples twice, as the reference does (one fused launch).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

VF_MACS, RN_MACS = 525056, 271360          # per point
# of which the COLOUR BRANCH (csrc/vfn_mlp16.hip, M16_C2): the 256 x 256 feature block of the VF net's last Linear and the
# rendering net except the 33 encoding columns (point, PE(view direction), normal) of its first layer
COLOUR_MACS = 256 * 256 + RN_MACS - 33 * 256
PEAK_F32_MFMA = 157.3                      # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA = 2500.0                     # TFLOP/s dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
SUSTAINED_F16_MFMA = 1570.0                # TFLOP/s a pure 32x32x16 f16 MFMA loop sustains on random operands with every CU
                                           # busy (power-limited clock ~1.8 GHz): tools/micro/mfma_power.hip, DESIGN.md §3


def hbm_traffic(f16: bool, kernel_class: str = "fused16", colour_products: int = 3):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 cannot run inside the
    benchmark): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md §HBM) + WRITE_SIZE, in bytes."""
    path = os.path.join(REPO, "profiles", "r02", "traffic_f16x3.json" if colour_products == 2 else "traffic_f16x3_3products.json")
    if not f16 or not os.path.exists(path):
        return None
    with open(path) as fh:
        t = json.load(fh)
    syms = {"vf_feat16": ("vfn_mlp16_kernel<9>",), "render16": ("vfn_mlp16_kernel<18>",),
            "fused16": ("vfn_mlp16_kernel<35>",) if colour_products == 2 else ("vfn_mlp16_kernel<3>",)}.get(kernel_class, ())
    for sym in syms:
        fetch, write = t["all_kernels"].get(f"FETCH_SIZE|{sym}"), t["all_kernels"].get(f"WRITE_SIZE|{sym}")
        if fetch is not None and write is not None:
            return int((2.0 * fetch + write) * 1024)
    return None


def build_scene(dev, n_rays, s_c, n_f, seed, perturb=True):
    import vf_nerf_amd
    from vf_nerf_amd import synthetic
    torch.manual_seed(0)
    cfg = vf_nerf_amd.shipped_config(dev, n_samples=s_c, n_importance=n_f, perturb=perturb, dir_to_normal_th=-0.2)
    model = vf_nerf_amd.VectorFieldNerf(cfg)
    model.eval()
    model.rng_seed = seed
    synthetic.scale_hidden_weights(model.vector_field_network, model.rendering_network, 2.0)
    with torch.no_grad():
        pts = synthetic.frustum_points(20000, seed=1234).to(dev)
        mean, std = synthetic.vector_head_stats_from_tanh(model.vector_field_network(pts, vector_only=True))
        synthetic.recentre_vector_head(model.vector_field_network, mean, std)
    # Replica-like pinhole: 1200x680, f = 600
    uv, pose, K = synthetic.pinhole_batch(n_rays, 1200, 680, 600.0, seed=100 + seed, device=dev)
    return model, uv, pose, K


def cpu_baseline(model, uv, pose, K, s_c, n_f, sample_rays=1024, budget_s=12.0):
    """Oracle render() on the host cores, bounded to ~10-30 s.  torch's intra-op pool stops scaling on these
    small per-layer GEMMs well before the box's 256 hardware threads (measured on the GPU box, rays/s at
    8/16/32/64/128 threads: 505/568/602/406/213), so 32 threads are used and reported."""
    from oracle import vfnerf_oracle as O
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    vf_sd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().cpu() for k, v in model.rendering_network.state_dict().items()}
    settings = O.RenderSettings(n_samples=s_c, n_fine=n_f, perturb=True, dir_to_normal_th=-0.2, fine_range=0.3,
                                density=O.DensityParams(scale_min=1.0))
    uv_c, pose_c, K_c = uv[:sample_rays].cpu(), pose[:sample_rays].cpu(), K[:sample_rays].cpu()
    g = torch.Generator().manual_seed(5)
    uni = dict(u_coarse=torch.rand(sample_rays, s_c, generator=g), u_fine=torch.rand(sample_rays, n_f, generator=g),
               u_add=torch.rand(sample_rays, n_f, generator=g))
    with torch.no_grad():
        O.render(uv_c, pose_c, K_c, vf_sd, rn_sd, settings, **uni)  # warm-up
        reps, t0 = 0, time.perf_counter()
        while True:
            ref = O.render(uv_c, pose_c, K_c, vf_sd, rn_sd, settings, **uni)
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget_s or reps >= 50:
                break
        # the "PSNR vs ref" half of the metric: the HIP path on the same rays with the same uniforms, checked
        # against the oracle's output (the oracle is the checker here, never the thing measured as `value`)
        out = model.render(pose[:sample_rays], uv[:sample_rays], K[:sample_rays], epoch=0, uniforms=uni)
    rgb, depth = out.coarse_rgb_values.cpu(), out.coarse_depth_map.cpu()
    same = (out.z_vals.cpu() == ref["z_vals"]).all(dim=1)
    parity = {"rays": sample_rays, "psnr_rgb_db": round(min(O.psnr(rgb, ref["rgb"]), 200.0), 2),
              "mean_abs_depth_err": float((depth - ref["depth"].reshape(depth.shape)).abs().mean()),
              "max_abs_rgb_err_identically_sampled_rays": float((rgb - ref["rgb"]).abs().max(dim=1)[0][same].max()),
              "rays_sampled_bit_identically": round(float(same.float().mean()), 4)}
    return {"value": round(sample_rays * reps / el, 1), "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x oracle render() of {sample_rays} rays x {s_c + n_f} samples, torch fp32 CPU, "
                      f"{threads} threads, {el:.1f} s"}, parity


def _oracle_inputs(model):
    vf_sd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().cpu() for k, v in model.rendering_network.state_dict().items()}
    return vf_sd, rn_sd


def view_bench(args, dev):
    """BASELINE.json configs[1]: one full Replica-like view (1200x680 = 816 000 rays) rendered in 1024-ray chunks x 128
    samples, perturb off, forward only.  A step = one full view.  The PSNR / depth error is taken against the oracle's
    image of the same camera at 1/8 resolution (150x85, intrinsics scaled), which the CPU finishes in ~20 s."""
    from vf_nerf_amd import synthetic
    from oracle import vfnerf_oracle as O
    chunk, (w, h, f) = args.rays if args.rays != 4096 else 1024, (1200, 680, 600.0)
    s_c, n_f = args.coarse, args.fine
    model, _, _, _ = build_scene(dev, 16, s_c, n_f, seed=0, perturb=False)
    model.precision = args.precision
    uv, pose, K = synthetic.pinhole_image(w, h, f, device=dev)
    n = uv.shape[0]

    def full_view():
        return model.render_chunked(pose, uv, K, epoch=0, chunk=chunk, n_streams=args.streams)

    with torch.no_grad():
        for _ in range(max(1, args.warmup // 3)):
            full_view()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full_view()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0

        # parity image: same camera at 1/8 resolution on both sides, identical u_add draw (Q9)
        ws, hs = w // 8, h // 8
        uv_s, pose_s, K_s = synthetic.pinhole_image(ws, hs, f / 8.0, device=dev)
        g = torch.Generator().manual_seed(21)
        uni = {"u_add": torch.rand(ws * hs, n_f, generator=g)}
        o = model.render(pose_s, uv_s, K_s, epoch=0, uniforms=uni)
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        settings = O.RenderSettings(n_samples=s_c, n_fine=n_f, perturb=False, dir_to_normal_th=-0.2, fine_range=0.3,
                                    density=O.DensityParams(scale_min=1.0))
        vf_sd, rn_sd = _oracle_inputs(model)
        t1 = time.perf_counter()
        ref = O.render(uv_s.cpu(), pose_s.cpu(), K_s.cpu(), vf_sd, rn_sd, settings, **uni)
        cpu_s = time.perf_counter() - t1
    rgb, depth = o.coarse_rgb_values.cpu(), o.coarse_depth_map.cpu()
    print(json.dumps({
        "metric": "rays/sec (full 1200x680 view, 1024-ray chunks, 128 samples/ray) + PSNR/depth vs ref",
        "value": round(n * args.steps / elapsed, 1), "unit": "rays/s", "n_gpus": 1, "steps": args.steps,
        "warmup": max(1, args.warmup // 3), "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3+f32acc" if args.precision == "f16x3" else "f32", "data": "synthetic",
        "config": {"workload": f"full view {w}x{h} = {n} rays in {chunk}-ray chunks x {s_c + n_f} samples, perturb off, "
                               f"forward only, consecutive chunks on {args.streams} stream(s) (BASELINE.json configs[1])"},
        "parity_vs_oracle": {"image": f"{ws}x{hs} (same camera, intrinsics / 8), {ws * hs} rays",
                             "psnr_rgb_db": round(min(O.psnr(rgb, ref["rgb"]), 200.0), 2),
                             "mean_abs_depth_err": float((depth - ref["depth"].reshape(depth.shape)).abs().mean()),
                             "max_abs_rgb_err": float((rgb - ref["rgb"]).abs().max()),
                             "argmax_indices_equal": bool((o.z_vals.cpu() == ref["z_vals"]).all()),
                             "oracle_seconds": round(cpu_s, 1)}}), flush=True)


def grid_bench(args, dev, rank, world, dist, sync):
    """BASELINE.json configs[4]: dense-grid queries of the vector field (marching-cubes input): res^3 points of one
    quadrant through ``grid.get_set_predictions`` (host grid -> pinned upload -> vector-only VF kernel -> pinned
    download), blocks of 100 000 points dealt round-robin to the ranks.  A step = one res^3 quadrant."""
    from vf_nerf_amd import grid
    from oracle import vfnerf_oracle as O
    model, _, _, _ = build_scene(dev, 16, 64, 64, seed=0)
    model.precision = args.precision
    dec = model.fine_vector_field_network
    res = args.grid_res
    ax = torch.linspace(-1.0, 1.0, res)
    samples = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3).contiguous()
    n = samples.shape[0]
    for _ in range(max(1, args.warmup // 3)):
        grid.get_set_predictions(dec, samples, 100000, dev, rank=rank, world_size=world)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        got = grid.get_set_predictions(dec, samples, 100000, dev, rank=rank, world_size=world)
    sync()
    elapsed = time.perf_counter() - t0
    # device-resident rate (grid already in HBM, no host copies): what the kernel itself sustains
    dsamples = samples[: min(n, 1 << 24)].to(dev)
    grid.get_set_predictions(dec, dsamples, 100000, dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    grid.get_set_predictions(dec, dsamples, 100000, dev)
    torch.cuda.synchronize()
    resident = dsamples.shape[0] / (time.perf_counter() - t1)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        vf_sd, _ = _oracle_inputs(model)
        idx = torch.arange(0, n, max(1, n // 4096))[:4096]
        idx = idx[(idx // 100000) % world == 0]                      # rows this rank evaluated
        ref = O.vf_mlp(samples[idx], vf_sd, 6, (4,))[:, :3]
        err = float((got[idx] - ref).abs().max())
        print(json.dumps({
            "metric": "grid points/sec (vector-field queries for quadrant marching cubes)",
            "value": round(n * args.steps / elapsed, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": max(1, args.warmup // 3), "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16x3+f32acc" if args.precision == "f16x3" else "f32", "data": "synthetic",
            "config": {"workload": f"{res}^3 = {n} grid points per quadrant, host grid in, host [n,3] out, "
                                   f"max_batch 100000 (BASELINE.json configs[4], one quadrant)",
                       "parallelism": f"blocks x{world}"},
            "device_resident_points_per_s": round(resident, 1),
            "max_abs_err_vs_oracle_4096_points": err}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def train_step_accounting(n_rays, s_c, s_t, n_sup, storage, gradients="fp32", separate_proposal=False):
    """ALGORITHMIC work and workspace traffic of one training step (DESIGN.md section 3 "Backward").
    FLOPs: the fine pass forward and twice that for its backward (VF + rendering, S_t samples) + forward and backward of the
    2 x n_sup supervision points through the VF net (+ the reference's separate vector-only proposal pass over the S_c samples
    only when it is actually run: with one VF evaluation per distinct sample the proposal samples are part of the fine pass).
    Bytes: what the kernels of the 16-bit path have to move through HBM per step — the saved activations (13 slots per fine
    point, 9 per supervision point; 1 KiB per slot and point as fp32, 512 B as f16 except the tanh'ed feature slot), their
    sign-bit words (32 B), the pre-activation gradients dY (1 KiB per slot and point, written by the chain, read by the
    weight-gradient kernels), the saved activations read once by the weight-gradient kernels, and the per-sample inputs and
    outputs (point, normal, colour, their gradients)."""
    m_f, m_s = n_rays * s_t, 2 * n_sup
    flops = 2.0 * ((n_rays * s_c * VF_MACS if separate_proposal else 0.0) + 3.0 * m_f * (VF_MACS + RN_MACS) + 3.0 * m_s * VF_MACS)
    relu_slot = 512 if storage == "f16" else 1024
    saved = m_f * (12 * relu_slot + 1024) + m_s * (8 * relu_slot + 1024)      # written by the forward ...
    masks = 32 * (13 * m_f + 9 * m_s)
    dy = (512 if gradients in ("bf16", "f16") else 1024) * (13 * m_f + 9 * m_s)
    small = (12 + 12 + 12 + 4 + 12 + 12) * m_f + 2 * 160 * (m_f + m_s)        # points, normals, colours, z + grads, aux tiles
    total = 2 * saved + 2 * masks + 2 * dy + small                             # ... and read back once; dY written + read
    return flops, total


def train_bench(args, model, uv, pose, K, dev, dist, rank, world, sync, emit=True):
    """One step = what the reference trainer does per batch, with synthetic targets (config 3 of BASELINE.json)."""
    from vf_nerf_amd import distributed as vdist, optim as voptim, supervision
    supervision.manual_seed(0x5eed + 7919 * (rank + 1))     # every rank draws its own supervision points
    centroid = torch.tensor([0.0, 0.0, 0.6], device=dev)
    s_t = args.coarse + args.fine
    g = torch.Generator().manual_seed(7 + rank)
    rgb_gt = torch.rand(args.rays, 3, generator=g).to(dev)
    depth_gt = (0.2 + 0.6 * torch.rand(args.rays, 1, generator=g)).to(dev)
    n_sup = (args.rays * s_t) // 10
    bucket = vdist.GradientBucket(model) if (world > 1 or dist is not None) else None
    clip = model.config.scheduler_config.clip_norm

    def step():
        out = model.render(pose, uv, K, epoch=0)
        # border + centre supervision points, as the trainer draws them (train/vector_field_nerf_train.py:198-214)
        bp, b_gt = supervision.sample_border_points(0.75, 1.0, n_sup, centroid, dev)
        cp, c_gt = supervision.sample_center_points(centroid, 0.05, n_sup, dev)
        sup_n = model.vector_field_network(torch.cat([bp, cp]))[:, :3]
        sup_gt = torch.cat([b_gt, c_gt])
        normals = out.coarse_normals.reshape(-1, 3)
        loss = 2.0 * (out.coarse_rgb_values - rgb_gt).abs().mean() + \
            0.5 * torch.clamp((out.coarse_depth_map - depth_gt).abs(), max=0.5).mean() + \
            0.1 * ((normals.norm(dim=-1) - 1.0) ** 2).mean() + 1.0 * ((sup_n - sup_gt) ** 2).mean()
        if bucket is not None:
            bucket.zero()
        else:
            model.optimizer.zero_grad()
        loss.backward()
        if bucket is not None:
            bucket.all_reduce_mean()
        voptim.clip_grad_norm_(model.parameters(), clip)       # = torch's clip_grad_norm_(..., foreach=False), Q4-exact
        model.optimizer.step()
        model.scheduler.step()
        return loss

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    from vf_nerf_amd.backward import StoredFinePass
    stored = model.reuse_proposal and StoredFinePass.applicable(model, args.rays, args.coarse, s_t - args.coarse)
    flops, ws_bytes = train_step_accounting(args.rays, args.coarse, s_t, n_sup, model.activation_storage, model.gradient_storage,
                                            separate_proposal=not stored)
    ms = elapsed / args.steps * 1e3
    rec = {"metric": "training rays/sec (4096-ray batch, 128 samples/ray, fwd+bwd+clip+Adam)",
           "value": round(args.rays * args.steps * world / elapsed, 1), "unit": "rays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3 fwd + bf16x3 bwd, f32 accumulate",
           "data": "synthetic", "final_loss": round(float(loss), 5),
           "activation_storage": model.activation_storage, "gradient_storage": model.gradient_storage,
           "workspace_layout": model.workspace_layout,
           "networks": "training mode (batch-statistics BatchNorm, layer-at-a-time fp32 kernels)"
           if model.vector_field_network.training else "eval mode (the shipped regime, fused kernels)",
           # per GPU: algorithmic FLOPs of a step / its duration against the f16 / 3 matrix ceiling, and the workspace bytes the
           # step has to move against the HBM peak (it sits between the two roofs)
           "algorithmic_tflop_per_step": round(flops / 1e12, 4),
           "achieved_tflops": round(flops / (ms * 1e-3) / 1e12, 1),
           "frac_of_f16_mfma_div3": round(flops / (ms * 1e-3) / 1e12 / (PEAK_F16_MFMA / 3.0), 4),
           "workspace_gb_per_step": round(ws_bytes / 1e9, 2),
           "workspace_tb_per_s": round(ws_bytes / (ms * 1e-3) / 1e12, 3),
           "frac_of_hbm_peak": round(ws_bytes / (ms * 1e-3) / 8e12, 4),
           "config": {"workload": f"train step: render({args.rays} rays x {s_t}) + 2x{n_sup} supervision "
                                  f"points through the VF net + L1/depth/unit-norm/supervision loss + "
                                  f"backward + clip_grad_norm_ + Adam (sequential semantics over the duplicated parameter list)"
                                  + (", gradients all-reduced over one flat bucket" if bucket is not None else "")}}
    if not emit:
        return rec
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rec


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--coarse", type=int, default=64)
    ap.add_argument("--fine", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustain-seconds", type=float, default=2.0,
                    help="untimed run of the same work right before the timed steps (the chip's clock settles under load)")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step sub-object of the default line")
    ap.add_argument("--train-steps", type=int, default=12, help="optimizer steps timed for the training sub-object")
    ap.add_argument("--activations", choices=("fp32", "f16"), default="f16",
                    help="training: storage of the hidden activations for the weight-gradient kernels (f16 = the default, "
                         "11-bit operands in one factor of dW, half the workspace traffic; fp32 = fp32-equivalent gradients)")
    ap.add_argument("--gradients", choices=("fp32", "f16", "bf16"), default=None,
                    help="training: storage of the pre-activation gradients between the dX chain and the weight-gradient kernels "
                         "(default: the model's)")
    ap.add_argument("--train-colour-products", type=int, choices=(2, 3), default=None,
                    help="training: products of the colour branch in the activation-saving forward (default: the model's, 3)")
    ap.add_argument("--sorted-fine-pass", action="store_true",
                    help="training: the reference's call structure (gradient-free proposal pass, then the saving forward over all "
                         "sorted samples) instead of one VF evaluation per distinct sample (backward.StoredFinePass)")
    ap.add_argument("--layout", choices=("fragment", "rows"), default=None,
                    help="training: workspace layout of the 16-bit path (default: the model's, fragment order)")
    ap.add_argument("--batch-statistics", action="store_true",
                    help="train workload with the networks in training mode (model.train(): batch-statistics BatchNorm, "
                         "Jacobian columns, directional derivatives) instead of the shipped eval-mode regime (SURVEY Q8)")
    ap.add_argument("--precision", choices=("f16x3", "fp32"), default="f16x3",
                    help="MLP kernels: f16x3 = split-half products on the f16 matrix cores, fp32 accumulate (default, "
                         "fp32-equivalent accuracy); fp32 = exact fp32 MFMA")
    ap.add_argument("--colour-products", type=int, choices=(2, 3), default=None,
                    help="f16 products per fp32-equivalent product in the colour branch of the f16x3 render (default: the model's, 2: "
                         "weights of the feature block + rendering net as their f16 roundings, colours within 2e-5; 3: fp32-equivalent)")
    ap.add_argument("--no-reuse", action="store_true",
                    help="evaluate the VF net on the proposal samples twice, as the reference does (one fused VF+rendering "
                         "launch over all S_c+N_f samples), instead of once")
    ap.add_argument("--workload", choices=("render", "view", "grid", "train"), default="render",
                    help="render = the headline line (default); view = BASELINE configs[1] full view in 1024-ray chunks + "
                         "PSNR/depth vs the oracle image; grid = configs[4] dense grid queries; train = configs[2] step")
    ap.add_argument("--grid-res", type=int, default=256)
    ap.add_argument("--streams", type=int, default=2,
                    help="view workload: HIP streams the consecutive ray chunks alternate over (1 = strictly one after the other)")
    ap.add_argument("--train", action="store_true",
                    help="time a training step instead (render with autograd + 2 supervision VF forwards + loss + "
                         "backward + clip + Adam, train/vector_field_nerf_train.py:177-260); not the headline metric")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dist = None
    # VFN_BENCH_FORCE_DIST=1: create the RCCL process group (barriers, the max-over-ranks all-reduce, the gradient
    # bucket) even with one rank — a single-GPU smoke test of the multi-GPU code path
    force_dist = os.environ.get("VFN_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from vf_nerf_amd import lib
    lib.load()  # fail loudly when the HIP extension is missing
    if args.train:
        args.workload = "train"
    if args.workload in ("view", "grid"):
        def sync0():
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
        if args.workload == "view":
            if world > 1:
                raise SystemExit("--workload view is a single-GPU configuration")
            view_bench(args, dev)
        else:
            grid_bench(args, dev, rank, world, dist, sync0)
        return
    s_c, n_f = args.coarse, args.fine
    s_t = s_c + n_f
    model, uv, pose, K = build_scene(dev, args.rays, s_c, n_f, seed=rank)
    model.precision = args.precision
    model.reuse_proposal = not args.no_reuse
    if args.colour_products:
        model.colour_products = args.colour_products

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.workload == "train":
        model.activation_storage = args.activations
        if args.gradients:
            model.gradient_storage = args.gradients
        if args.layout:
            model.workspace_layout = args.layout
        if args.batch_statistics:
            model.train()
        model.reuse_proposal_training = not args.sorted_fine_pass
        if args.train_colour_products:
            model.training_colour_products = args.train_colour_products
        train_bench(args, model, uv, pose, K, dev, dist, rank, world, sync)
        return

    with torch.no_grad():
        for _ in range(args.warmup):
            model.render(pose, uv, K, epoch=0)
        # The kernel is power-limited: right after idle the chip boosts, and K = 20 steps are 40 ms.  So the same work runs
        # untimed for --sustain-seconds first and the timed steps follow it without a gap: `value` is a sustained figure by
        # construction, whatever K the caller asks for.
        torch.cuda.synchronize()
        t_burn = time.perf_counter()
        while time.perf_counter() - t_burn < args.sustain_seconds:
            for _ in range(16):
                model.render(pose, uv, K, epoch=0)
            torch.cuda.synchronize()
        events = []
        sync()
        t0 = time.perf_counter()
        for i in range(args.steps):
            # HIP events around the MLP launches on every fourth step of the timed region: a pair of timing events costs a
            # few microseconds of stream time, 1.4 % of the step when every launch of every step is bracketed
            model._kernel_events = events if i % 4 == 0 else None
            out = model.render(pose, uv, K, epoch=0)
        sync()
        elapsed = time.perf_counter() - t0
    model._kernel_events = None
    # dominant kernel class = the one with the largest share of the timed region (HIP events on the launch stream)
    per_class = {}
    for name, e0, e1 in events:
        per_class.setdefault(name, []).append(e0.elapsed_time(e1))
    dom = max(per_class, key=lambda k: sum(per_class[k]))
    kernel_ms = sum(per_class[dom]) / len(per_class[dom])
    launches_per_step = len(per_class[dom]) / max(1, (args.steps + 3) // 4)

    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        rays_per_s = args.rays * args.steps * world / elapsed
        # ALGORITHMIC fp32-equivalent FLOPs of one launch of that kernel (SURVEY.md §8d per-point figures x its points)
        # default pipeline: the fused VF + rendering launch twice, on the S_c proposal samples and on the N_f new samples (one VF
        # evaluation per distinct sample; results scattered to their sorted positions); --no-reuse / fp32: the proposal pass
        # (vector columns only) and one fused launch over all S_t samples
        split = args.precision == "f16x3" and not args.no_reuse
        points = {"vf_feat16": args.rays * s_c, "render16": args.rays * s_c,
                  "fused16": args.rays * s_t / (launches_per_step if split else 1.0)}.get(dom, args.rays * s_t)
        macs = {"vf_feat16": VF_MACS, "render16": RN_MACS}.get(dom, VF_MACS + RN_MACS)
        flops_launch = 2.0 * macs * points
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        hits = float((out.coarse_depth_map > 0).float().mean())
        f16 = args.precision == "f16x3"
        # achieved = ALGORITHMIC fp32-equivalent FLOPs per launch / measured duration.  The f16x3 kernel spends three
        # f16 MFMA products per fp32-equivalent product — two in the colour branch when colour_products == 2 — so its matrix-pipe
        # ceiling is the dense f16 peak / (f16 products per fp32-equivalent product, averaged over the launch's MACs).
        cp = int(model.colour_products) if f16 and dom == "fused16" else 3
        products = 3.0 - (COLOUR_MACS / (VF_MACS + RN_MACS) if cp == 2 else 0.0)
        peak = PEAK_F16_MFMA / products if f16 else PEAK_F32_MFMA
        roof = {"bound": "mfma",
                "kernel": {"vf_feat16": "vfn_mlp16_kernel<M16_VF_BLK> (VF MLP on the proposal samples, feature blocks out)",
                           "render16": "vfn_mlp16_kernel<M16_RN_BLK> (rendering MLP on the proposal samples' stored feature blocks)",
                           "fused16": ("vfn_mlp16_kernel<M16_FUSED | M16_C2>" if cp == 2 else "vfn_mlp16_kernel<M16_FUSED>") +
                                      " (VF MLP + rendering MLP; default: one launch on the proposal samples, one on the new fine samples)",
                           "fused32": "vfn_mlp_kernel<MODE_FUSED> (VF MLP + rendering MLP, fine pass)"}[dom],
                "launches_per_step": launches_per_step,
                "kernel_ms_per_step_by_class": {k: round(sum(v) / max(1, (args.steps + 3) // 4), 4) for k, v in per_class.items()},
                "event_sampling": "HIP events around the MLP launches of every 4th timed step",
                "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": hbm_traffic(f16, dom, cp),
                # what one launch has to move: points in, vector columns out, plus the 1 KiB feature block per point that the
                # split launches hand over (written by vf_feat16, read by render16); weights stream from L2
                "algorithmic_bytes": int(points * {"vf_feat16": 12 + 12 + 1024, "render16": 1024 + 12 + 12 + 4 + 24}.get(dom, 12 + 4 + 24)),
                "flops_per_launch": flops_launch, "avg_launch_ms": round(kernel_ms, 4),
                "peak_definition": ((f"dense f16 MFMA 2500 TFLOP/s / {products:.4g} f16 products per fp32-equivalent product (3 in the vector-field "
                                     f"trunk, vector head and encoding columns, 2 in the colour branch = {COLOUR_MACS} of {VF_MACS + RN_MACS} MACs per sample)"
                                     if cp == 2 else "dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-equivalent product") if f16 else
                                    "fp32 MFMA 157.3 TFLOP/s"),
                "executed_f16_tflops": round(achieved * products, 1) if f16 else None,
                # BASELINE.md §3 states the path's roofline against the fp32 matrix peak:
                "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA, 4)}
        if f16:   # context, not the graded fraction: the power-limited MFMA rate measured on this chip
            roof["frac_of_measured_sustained_f16_mfma"] = round(achieved / (SUSTAINED_F16_MFMA / products), 4)
        line = {
            "metric": "rays/sec (4096-ray chunk, 128 samples/ray) + PSNR vs ref",
            "value": round(rays_per_s, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16x3+f32acc (colour branch: f16 weights x split activations, 2 products)" if cp == 2 else "f16x3+f32acc") if f16 else "f32",
            "data": "synthetic",
            "sustained": f"timed steps follow {args.sustain_seconds:g} s of the same work without a gap",
            "config": {"workload": f"VectorFieldNerf.render forward, {args.rays}-ray chunk x {s_t} samples/ray "
                                   f"(S_c={s_c} + N_f={n_f}), shipped 9x256 VF + 5x256 rendering MLPs, eval-mode BN, "
                                   f"stratified sampling on device Philox, Replica-like 1200x680 pinhole",
                       "rays_per_chunk_per_gpu": args.rays, "samples_per_ray": s_t, "parallelism": f"rays x{world}",
                       "vf_evaluations_per_ray": s_t if getattr(model, "reuse_proposal", False) and f16 else s_c + s_t,
                       "colour_products": cp if f16 else None,
                       "rays_with_nonzero_depth": round(hits, 3)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity_vs_oracle"] = cpu_baseline(model, uv, pose, K, s_c, n_f)
    # BASELINE.json configs[2] beside the headline line (outside its timed region): a few optimizer steps on the same batch
    # size, every rank, gradients all-reduced when there is more than one
    train_rec = None
    if not args.no_train:
        targs = argparse.Namespace(**vars(args))
        targs.steps, targs.warmup = args.train_steps, 5      # (the first steps size the caching allocator's 15 GB of workspace)
        del out
        torch.cuda.empty_cache()
        tmodel, tuv, tpose, tK = build_scene(dev, args.rays, s_c, n_f, seed=rank)
        tmodel.precision = args.precision
        tmodel.activation_storage = args.activations
        if args.gradients:
            tmodel.gradient_storage = args.gradients
        if args.layout:
            tmodel.workspace_layout = args.layout
        train_rec = train_bench(targs, tmodel, tuv, tpose, tK, dev, dist, rank, world, sync, emit=False)
    if rank == 0:
        if train_rec is not None:
            line["train"] = {k: train_rec[k] for k in ("value", "unit", "ms_per_step", "steps", "dtype", "activation_storage", "gradient_storage", "workspace_layout",
                                                       "algorithmic_tflop_per_step", "achieved_tflops", "frac_of_f16_mfma_div3",
                                                       "workspace_gb_per_step", "workspace_tb_per_s", "frac_of_hbm_peak", "final_loss")}
            line["train"]["workload"] = train_rec["config"]["workload"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
