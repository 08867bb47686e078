#!/usr/bin/env python3
"""Headline benchmark: rays/s of ``VectorFieldNerf.render`` (forward, eval mode) on 4096-ray chunks with
128 samples per ray (S_c = N_f = 64), one process per GPU, on the synthetic random-weight scene BASELINE.json's north_star names
(`value`, `roofline`); the same path on weights the reference's own trainer arrived at (tests/golden/trained_far.npz: data, not code)
is timed beside it (`value_trained_weights`, `roofline_trained_weights`); --weights trained swaps the two.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3          # starts its own 8 ranks (torch.distributed.run as a CHILD process)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus 8 --steps 20 --warmup 3             # ... or is started as a rank by the caller's launcher

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
                    traffic = FETCH_SIZE x 2 + WRITE_SIZE of that kernel from two rocprofv3 --pmc child runs of this command made on
                    this box before the timed run (live_hbm_traffic; the committed passes of profiles/ are the fallback);
                    effective_clock_ghz from in-kernel s_memtime / s_memrealtime stamps.
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


def kernel_sources_sha16() -> str:
    import hashlib
    src = b"".join(open(os.path.join(REPO, "vf_nerf_amd", "csrc", f), "rb").read() for f in ("vfn_mlp16.hip",))
    return hashlib.sha256(src).hexdigest()[:16]


def hbm_traffic(f16: bool, kernel_class: str = "fused16", colour_products: int = 3):
    """HBM bytes per launch of the dominant kernel: from THIS run's own PMC passes when main() made them (live_hbm_traffic: two
    rocprofv3 child runs before the timed run), else from the committed passes of profiles/rNN/ (the counters need their own passes
    either way): FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md section HBM) + WRITE_SIZE, in
    bytes -> (bytes | None, provenance).  The provenance names the file and says whether the kernel sources have changed since the
    counters were collected (tools/export_profiles.py stores a fingerprint of them)."""
    if _LIVE_TRAFFIC is not None and kernel_class == "fused16" and f16:
        if _LIVE_TRAFFIC[0] is not None:
            return _LIVE_TRAFFIC
    syms = {"vf_feat16": ("vfn_mlp16_kernel<9>",), "render16": ("vfn_mlp16_kernel<18>",),
            "fused16": ("vfn_mlp16_kernel<35>",) if colour_products == 2 else ("vfn_mlp16_kernel<3>",)}.get(kernel_class, ())
    # newest round first; a round's file is used when it holds BOTH counters of the kernel in question (the 128-ray self-check
    # launches of the other product count also appear in a file: their averages are not what a full launch moves)
    cands = [os.path.join(REPO, "profiles", r, n) for r in ("r06", "r05", "r04", "r03", "r02")
             for n in (("traffic_f16x3.json",) if (colour_products == 2 or r >= "r04") else ("traffic_f16x3_3products.json",))]
    path = t = None
    for cand in cands:
        if f16 and os.path.exists(cand):
            with open(cand) as fh:
                tt = json.load(fh)
            if any(f"FETCH_SIZE|{sym}" in tt["all_kernels"] and f"WRITE_SIZE|{sym}" in tt["all_kernels"] for sym in syms):
                path, t = cand, tt
                break
    if not f16 or path is None:
        return None, None
    sha = t.get("kernel_sources_sha16")
    prov = {"file": os.path.relpath(path, REPO), "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes)",
            "kernel_sources_unchanged_since": (sha == kernel_sources_sha16()) if sha else None}
    if _LIVE_TRAFFIC is not None and kernel_class == "fused16":
        prov["live_pass"] = _LIVE_TRAFFIC[1]         # (why this run's own passes did not give the figure)
    for sym in syms:
        fetch, write = t["all_kernels"].get(f"FETCH_SIZE|{sym}"), t["all_kernels"].get(f"WRITE_SIZE|{sym}")
        if fetch is not None and write is not None:
            return int((2.0 * fetch + write) * 1024), prov
    return None, prov


_LIVE_TRAFFIC = None      # (bytes per launch, provenance) measured by live_hbm_traffic() in THIS run, or None


def under_profiler() -> bool:
    return any("rocprof" in (os.environ.get(k) or "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_PATH")) or \
        any(k.startswith("ROCPROF") for k in os.environ)


def live_hbm_traffic(args, timeout_s: float = 60.0):
    """The PMC passes of profiles/rNN/traffic_f16x3.json made HERE, on the box the line is measured on: two child processes
    (`rocprofv3 --pmc FETCH_SIZE -- python3 bench.py ...`, then WRITE_SIZE: each counter in its own pass, no trace domains, as
    MI355X_MICROARCH.md prescribes) that run the headline render for a few steps, started before this process makes any GPU call;
    FETCH_SIZE x 2 (the gfx950 correction) + WRITE_SIZE, KB -> bytes, averaged over the launches of the dominant kernel.
    -> (bytes, provenance) or (None, why not)."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, {"live": False, "why": "rocprofv3 is not on PATH"}
    sym = "vfn_mlp16_kernel<35>" if args.colour_products == 2 else "vfn_mlp16_kernel<3>"
    try:
        tmp = tempfile.mkdtemp(prefix="vfn_traffic_", dir="/tmp")
    except OSError as e:
        return None, {"live": False, "why": f"no scratch directory under /tmp: {e}"[:300]}
    child = [sys.executable, os.path.abspath(__file__), "--no-live-traffic", "--no-cpu-baseline", "--no-train", "--no-two-product-leg",
             "--no-other-scene-leg", "--no-shipped-rows", "--steps", "5", "--warmup", "2", "--sustain-seconds", "0", "--rays", str(args.rays), "--coarse",
             str(args.coarse), "--fine", str(args.fine), "--weights", args.weights] + \
            (["--colour-products", str(args.colour_products)] if args.colour_products else [])
    kb = {}
    t0 = time.perf_counter()
    try:
        for counter, stem in (("FETCH_SIZE", "pf"), ("WRITE_SIZE", "pw")):
            out_dir = os.path.join(tmp, stem)
            # the profiler and the profiled python are one process GROUP: a pass that outlives the cap is ended as a whole — killing
            # rocprofv3 alone would leave its child rendering on the GPU while the timed region starts
            proc = subprocess.Popen([exe, "--pmc", counter, "-d", out_dir, "-o", stem, "--"] + child, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"),
                                    stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
            try:
                _, err = proc.communicate(timeout=timeout_s)
            except BaseException:
                import signal
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except OSError:
                    pass
                proc.wait()
                raise
            r = subprocess.CompletedProcess(proc.args, proc.returncode, None, err)
            db = None
            for root, _, files in os.walk(out_dir):
                for f in files:
                    if f.endswith("_results.db"):
                        db = os.path.join(root, f)
            if r.returncode != 0 or db is None:
                return None, {"live": False, "why": f"the {counter} pass failed (exit {r.returncode}): {r.stderr.decode(errors='replace')[-200:]}"}
            con = sqlite3.connect(db)
            rows = list(con.execute("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? group by kernel_name",
                                    (counter,)))
            con.close()
            for name, val, n in rows:
                short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                if short == sym:
                    kb[counter] = (float(val), int(n))
            if counter not in kb:
                return None, {"live": False, "why": f"no {sym} launch in the {counter} pass"}
    except Exception as e:      # a timeout, a profiler that cannot start here: the committed file stays the source
        return None, {"live": False, "why": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    total = int((2.0 * kb["FETCH_SIZE"][0] + kb["WRITE_SIZE"][0]) * 1024)
    return total, {"live": True, "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, two child runs of this command on this box before "
                                                   "the timed run (separate passes, no trace domains)",
                   "kernel": sym, "launches_averaged": {"FETCH_SIZE": kb["FETCH_SIZE"][1], "WRITE_SIZE": kb["WRITE_SIZE"][1]},
                   "fetch_kb_x2_gfx950": round(2.0 * kb["FETCH_SIZE"][0], 1), "write_kb": round(kb["WRITE_SIZE"][0], 1),
                   "seconds": round(time.perf_counter() - t0, 1)}


def build_scene(dev, n_rays, s_c, n_f, seed, perturb=True, weight_seed=0):
    import vf_nerf_amd
    from vf_nerf_amd import synthetic
    torch.manual_seed(weight_seed)
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


TRAINED_SCENE_FOCAL = 1143.0
TRAINED_FIXTURES = ("trained_far.npz", "trained_256.npz")      # tests/golden: weights the reference's own trainer arrived at


def load_trained_weights(model):
    """Put the weights of the first available trained fixture (tests/golden/trained_far.npz: 6 000 optimizer steps of the
    reference's train_epoch on 256-ray batches, make_trained_golden.py --far; else trained_256.npz, 1 200 steps) into ``model``.
    The arrays travel as data; nothing of the reference is read.  -> a short description, or None when no fixture is there."""
    import ast
    import numpy as np
    for name in TRAINED_FIXTURES:
        path = os.path.join(REPO, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        raw = np.load(path)
        for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density)):
            mod.load_state_dict({k[len(f"w.{tag}."):]: torch.from_numpy(raw[k]) for k in raw.files if k.startswith(f"w.{tag}.")})
        model.to(model.config.cuda_config.device)
        model._invalidate_packs()
        recipe = ast.literal_eval(str(raw["train_recipe"]))
        steps = recipe["epochs"] * recipe["steps_per_epoch"]
        gap = float(raw["curve.colour_gap"][-1][1]) if "curve.colour_gap" in raw.files else None
        return {"fixture": f"tests/golden/{name}", "trained_by": f"the reference's train_epoch, {steps} optimizer steps x {recipe['n_rays']} rays "
                                                                   f"(tests/golden/make_trained_golden.py)",
                "simulated_two_product_colour_gap": gap}
    return None


def build_trained_scene(dev, n_rays, s_c, n_f, seed):
    """The headline scene on TRAINED weights: the geometry, samplers and Replica-like 1200x680 camera of build_scene, looking at
    the teacher scene from the first training view, with the networks and density scalars of the trained fixture."""
    from vf_nerf_amd import synthetic
    model, uv, pose, K = build_scene(dev, n_rays, s_c, n_f, seed)
    what = load_trained_weights(model)
    if what is None:
        return None
    # 1200 x 680 pixels of the first training view; the focal length keeps the view INSIDE the frustum the model was trained on
    # (64 x 64 pixels at f = 60: |x / z| <= 0.525 -> f = 1143 for 1200 pixels).  Outside it a trained field extrapolates to vectors
    # of any length, the normalisation in front of the density is ill-conditioned there, and no two fp32 evaluations agree to 1e-4.
    uv, pose, K = synthetic.pinhole_batch(n_rays, 1200, 680, TRAINED_SCENE_FOCAL, seed=100 + seed, device=dev, pose=synthetic.orbit_pose(-35.0, 5.0, 0.9))
    what["camera"] = f"1200x680 pinhole, f = {TRAINED_SCENE_FOCAL:g} (the horizontal field of view of the 64x64 / f = 60 training views), first training pose"
    return model, uv, pose, K, what


def two_product_check(model, uv, pose, K, rays=1024):
    """What the opt-in two-product colour branch does on THIS model's weights: its colours against the three-product colours on the
    same rays and draws (guard off: the raw difference), and what the range guard's strict self-check decides when asked."""
    import warnings
    keep = (model.colour_products, model.f16x3_guard, model._rng_offset)
    n = min(rays, uv.shape[0])
    g = torch.Generator().manual_seed(11)
    s_c, n_f = model.ray_sampler.N_samples, model.fine_sampler.N_samples
    uni = dict(u_coarse=torch.rand(n, s_c, generator=g), u_fine=torch.rand(n, n_f, generator=g), u_add=torch.rand(n, n_f, generator=g))
    outs = {}
    with torch.no_grad():
        model.f16x3_guard = "off"
        for k in (3, 2):
            model.colour_products = k
            outs[k] = model.render(pose[:n], uv[:n], K[:n], epoch=0, uniforms=uni)
        model.f16x3_guard = "strict"
        model.colour_products = 2
        model.range_guard.colour_products_reason = None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model.render(pose[:n], uv[:n], K[:n], epoch=0, uniforms=uni)
    rec = {"rays": n, "max_abs_colour_difference": float((outs[2].coarse_colors - outs[3].coarse_colors).abs().max()),
           "max_abs_rgb_difference": float((outs[2].coarse_rgb_values - outs[3].coarse_rgb_values).abs().max()),
           "geometry_bit_identical": bool(torch.equal(outs[2].z_vals, outs[3].z_vals) and torch.equal(outs[2].coarse_depth_map, outs[3].coarse_depth_map)),
           "guard_tolerance": __import__("vf_nerf_amd.guard", fromlist=["x"]).COLOUR_CHECK_TOL,
           "strict_guard_keeps_two_products": model.colour_products == 2,
           "colour_products_reason": model.range_guard.colour_products_reason}
    model.range_guard.colour_products_reason = None
    model.colour_products, model.f16x3_guard, model._rng_offset = keep
    return rec


def cpu_baseline(model, uv, pose, K, s_c, n_f, sample_rays=1024, budget_s=12.0):
    """Oracle render() on the host cores, bounded to ~10-30 s.  torch's intra-op pool stops scaling on these
    small per-layer GEMMs well before the box's 256 hardware threads (measured on the GPU box, rays/s at
    8/16/32/64/128 threads: 505/568/602/406/213), so 32 threads are used and reported."""
    from oracle import vfnerf_oracle as O
    threads = min(32, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    vf_sd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().cpu() for k, v in model.rendering_network.state_dict().items()}
    # (the density's three learnable scalars are part of the model's state: trained weights carry trained ones)
    den = model.density
    settings = O.RenderSettings(n_samples=s_c, n_fine=n_f, perturb=True, dir_to_normal_th=-0.2, fine_range=0.3,
                                density=O.DensityParams(beta=float(den.beta.detach()), mean=float(den.mean.detach()), scale=float(den.scale.detach()),
                                                        beta_bounds=tuple(float(x) for x in den.beta_bounds), mean_bounds=tuple(float(x) for x in den.mean_bounds),
                                                        scale_min=float(den.scale_min)))
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
    parity = {"rays": sample_rays, "colour_products_ran": int(model.colour_products) if model.uses_f16x3() else None,
              "psnr_rgb_db": round(min(O.psnr(rgb, ref["rgb"]), 200.0), 2)}
    parity.update(ray_accounting(out.z_vals.cpu(), rgb, depth, ref["z_vals"], ref["rgb"], ref["depth"].reshape(depth.shape)))
    rec = {"value": round(sample_rays * reps / el, 1), "unit": "rays/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x oracle render() of {sample_rays} rays x {s_c + n_f} samples, torch fp32 CPU, "
                     f"{threads} threads, {el:.1f} s"}
    # how the port compares with the reference's own CPU render(): measured where the reference exists (the build container,
    # tools/cpu_reference_vs_port.py) and carried as a committed file — nothing of the reference is read here
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r03", "cpu_reference_vs_port.json")) as fh:
            cmp = json.load(fh)
        rec["port_over_reference_speed"] = {"ratio": cmp["oracle_over_reference"], "measured_on": cmp["workload"],
                                            "file": "profiles/r03/cpu_reference_vs_port.json"}
    except (OSError, KeyError, ValueError):
        pass
    return rec, parity


def ray_accounting(z, rgb, depth, ref_z, ref_rgb, ref_depth, tol=1e-4, z_tol=0.0):
    """Per-ray accounting of a render against a reference: how many rays sampled bit-identical depths (z_tol > 0: every depth
    within z_tol — a float64 reference has no bit-identical depths), and of THOSE how many are inside the contract (|rgb| and
    |depth| errors below `tol`); worst errors over them and over all rays."""
    same = (z == ref_z).all(dim=1) if z_tol == 0.0 else ((z.double() - ref_z.double()).abs() <= z_tol).all(dim=1)
    e_rgb = (rgb - ref_rgb).abs().max(dim=1)[0]
    e_dep = (depth - ref_depth).abs().reshape(-1)
    inside = (e_rgb < tol) & (e_dep < tol)
    n_same = int(same.sum())
    return {"rays_sampled_bit_identically": round(float(same.float().mean()), 5), "rays_with_different_z": int((~same).sum()),
            "frac_rays_within_1e-4": round(float(inside[same].float().mean()), 5) if n_same else None,
            "frac_all_rays_within_1e-4": round(float(inside.float().mean()), 5),
            "max_abs_rgb_err_identically_sampled_rays": float(e_rgb[same].max()) if n_same else None,
            "max_abs_depth_err_identically_sampled_rays": float(e_dep[same].max()) if n_same else None,
            "max_abs_rgb_err": float(e_rgb.max()), "mean_abs_depth_err": float(e_dep.mean())}


def ray_error_sources(rgb_a, depth_a, normals_a, sigma_a, ref, settings, O, tol=1e-4):
    """WHY a ray of side a (a HIP render) is outside the contract against side b (the oracle's dict ``ref`` on the same rays and draws).
    The candidates: (1) a density decision that fell on the other side — the Laplace density's ReLU (density_functions.py:129-151) or the
    mask of models/nerf/vector_field_nerf.py:463-469 — i.e. a sample with sigma == 0 on exactly one side ("flipped"); (2) the normals'
    own error (|n_a - n_b|, contract 1e-4) amplified by the reference's function downstream of them (cosines of normalised neighbours
    through an 11-sample window -> Laplace density of scale 100 -> transmittance scan -> weights normalised by their sum); (3) anything
    else side a did differently (colours, density, scan, compositing).  Per ray outside ``tol``:
      max_abs_normal_err   max over the ray's samples of |n_a - n_b|                     (what side a actually got wrong)
      amplification        the ray's rgb / depth error over that
      min_normal_length    shortest |n_b| among the samples that carry weight on either side
      flipped_samples      samples with sigma == 0 on exactly one side
      residual             |rgb, depth of side a - the ORACLE's density + weights + composite evaluated on side a's normals (oracle colours,
                           oracle depths)|: what is left once the normals' difference is accounted for, i.e. source (3)
    A ray is EXPLAINED BY ITS NORMALS when residual < tol / 2 and max_abs_normal_err < tol / 5: what side a did downstream of the normals
    reproduces the oracle's own function to within half the contract (measured: 1e-7), the normals themselves are well inside the
    contract, and the reference's function amplifies their difference past it on that ray.  Measured at view scale (profiles/r05/): no
    ray has a flipped sample; the oracle's OWN fp32 evaluation against its float64 one shows the same rays with amplifications of
    46-670x on normal differences of 3e-6, the f16x3 kernels 19-280x on 5-8e-6, the exact-fp32 kernels 19-150x on 4-6e-6: the number of
    rays outside 1e-4 (6 / 14 / 10 of 12 750) follows the size of the normals' error, nothing else."""
    rgb_b, depth_b = ref["rgb"].float(), ref["depth"].float().reshape(-1)
    e_rgb = (rgb_a - rgb_b).abs().max(dim=1)[0]
    e_dep = (depth_a.reshape(-1) - depth_b).abs()
    bad = torch.nonzero((e_rgb >= tol) | (e_dep >= tol)).reshape(-1)
    n, s_t = ref["z_vals"].shape
    na, nb = normals_a.reshape(n, s_t, 3).float(), ref["normals"].reshape(n, s_t, 3).float()
    # the oracle's own function downstream of the normals, on side a's normals
    sig_x = O.ray_density(na, ref["ray_dirs"].float(), settings.n_window, settings.dir_to_normal_th, settings.density)
    w_x = O.volsdf_weights(ref["z_vals"].float(), sig_x, settings.normalize)
    rgb_x = torch.sum(w_x.unsqueeze(-1) * ref["colors"].float().reshape(n, s_t, 3), dim=1)
    dep_x = torch.sum(w_x * ref["z_vals"].float(), dim=1)
    res = torch.maximum((rgb_a - rgb_x).abs().max(dim=1)[0], (depth_a.reshape(-1) - dep_x).abs())
    dn = (na - nb).norm(dim=2)
    carries = (ref["weights"].float() > 0) | (w_x > 0)
    length = torch.where(carries, nb.norm(dim=2), torch.full_like(dn, float("inf")))
    flip = (sigma_a.reshape(n, s_t) == 0) != (ref["sigma"].reshape(n, s_t) == 0)
    rows = []
    for r in bad.tolist():
        err = max(float(e_rgb[r]), float(e_dep[r]))
        rows.append({"ray": r, "rgb_err": float(e_rgb[r]), "depth_err": float(e_dep[r]), "max_abs_normal_err": float(dn[r].max()),
                     "amplification": round(err / max(float(dn[r].max()), 1e-30), 1), "min_normal_length": float(length[r].min()),
                     "flipped_samples": int(flip[r].sum()), "residual": float(res[r])})
    explained = sum(1 for q in rows if q["residual"] < tol / 2 and q["max_abs_normal_err"] < tol / 5)
    return {"out_of_tolerance": len(rows), "explained_by_their_normals": explained, "unexplained": len(rows) - explained,
            "max_abs_normal_err_all_rays": float(dn.max()), "largest_residual": max((q["residual"] for q in rows), default=None),
            "rays_with_a_flipped_sample": sum(1 for q in rows if q["flipped_samples"]),
            "rays": sorted(rows, key=lambda q: -max(q["rgb_err"], q["depth_err"]))}


def trained_weights_parity(dev, precision="f16x3", fixture=None):
    """The DEFAULT render path on weights the reference's own trainer produced (tests/golden/trained_far.npz: 6 000 steps of
    train_epoch on 256-ray batches of a teacher-rendered target — far from the init family —, else trained_256.npz: 1 200 steps
    x 64 rays; stage outputs captured from the reference's render(); make_trained_golden.py)
    against the reference's outputs stored there — no oracle involved.  The range guard runs in strict mode: what it reports,
    and which product count actually produced the colours, is part of the record."""
    import ast
    import warnings
    import numpy as np
    import vf_nerf_amd
    golden = os.path.join(REPO, "tests", "golden")
    path = next((q for q in (os.path.join(golden, f) for f in ((fixture,) if fixture else TRAINED_FIXTURES)) if os.path.exists(q)), None)
    if path is None:
        return None
    raw = np.load(path)
    fx = ast.literal_eval(str(raw["fixture"]))
    # (``fixture``: another reference-captured fixture — one that names the file its trained weights live in, or one on random weights
    #  rebuilt from its recipe the way tests/helpers.build_model does)
    src = np.load(os.path.join(golden, fx["weights_in"] + ".npz")) if "weights_in" in fx else raw
    recipe = ast.literal_eval(str(src["train_recipe"])) if "train_recipe" in src.files else None
    if fx.get("trained"):
        cfg = vf_nerf_amd.shipped_config(dev, n_samples=fx["n_samples"], n_importance=fx["n_importance"], perturb=fx["perturb"],
                                         near=fx["near"], far=fx["far"], fine_range=fx["fine_range"], dir_to_normal_th=fx["th"],
                                         n_window=fx["n_window"])
        model = vf_nerf_amd.VectorFieldNerf(cfg)
        for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density)):
            mod.load_state_dict({k[len(f"w.{tag}."):]: torch.from_numpy(src[k]) for k in src.files if k.startswith(f"w.{tag}.")})
    else:
        from vf_nerf_amd import synthetic
        torch.manual_seed(fx["seed"])
        cfg = vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=fx["n_samples"], n_importance=fx["n_importance"], perturb=fx["perturb"],
                                         near=fx["near"], far=fx["far"], fine_range=fx["fine_range"], dir_to_normal_th=fx["th"],
                                         n_window=fx["n_window"])
        model = vf_nerf_amd.VectorFieldNerf(cfg)
        synthetic.scale_hidden_weights(model.vector_field_network, model.rendering_network, fx["gain"])
        with torch.no_grad():
            last = model.vector_field_network.layers[8]
            last.weight[:3] = torch.from_numpy(raw["head_weight"])
            last.bias[:3] = torch.from_numpy(raw["head_bias"])
        model.to(dev)
        model.config.cuda_config.device = dev
        model.config.cos_sim_weights = model.config.cos_sim_weights.to(dev)
    model.eval()
    model.precision = precision
    model.f16x3_guard = "strict"
    g = lambda k: torch.from_numpy(raw[k]).to(dev)           # noqa: E731
    uni = {k: g(k) for k in ("u_coarse", "u_fine", "u_add")}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            out = model.render(g("pose"), g("uv"), g("intrinsics"), epoch=0, uniforms=uni)
    rec = {"fixture": f"tests/golden/{os.path.basename(path)} (" + (f"reference train_epoch x {recipe['epochs'] * recipe['steps_per_epoch']} steps x "
                      f"{recipe['n_rays']} rays" if recipe else "random weights") + "; reference render() outputs)",
           "rays": fx["n_rays"], "samples": fx["n_samples"] + fx["n_importance"],
           "colour_products_ran": int(model.colour_products) if model.uses_f16x3() else None,
           "kernels": "f16x3" if model.uses_f16x3() else "fp32",
           "guard": {"mode": "strict", "switched_to_fp32": model.f16x3_disabled, "colour_products_reason": model.range_guard.colour_products_reason,
                     "warnings": [str(w.message)[:160] for w in caught if issubclass(w.category, RuntimeWarning)]},
           "max_abs_normals_err": float((out.coarse_normals.cpu() - torch.from_numpy(raw["normals"])).abs().max())}
    rec.update(ray_accounting(out.z_vals.cpu(), out.coarse_rgb_values.cpu(), out.coarse_depth_map.cpu(), torch.from_numpy(raw["z_vals"]),
                              torch.from_numpy(raw["rgb"]), torch.from_numpy(raw["depth"]).reshape(-1, 1)))
    return rec


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

    from vf_nerf_amd import evaluator
    if args.as_evaluator:
        # the evaluator's own situation (evaluation/methods.py:504-545): the dataset hands the view over as HOST tensors with the pose
        # and intrinsics replicated per ray, and wants rgb / depth back on the host -> evaluator.render_view (what dropin.install()
        # puts behind evaluation.methods.render_images); uploads and the download are inside the timed region
        uv_h, pose_h, K_h = uv.cpu(), pose.cpu(), K.cpu()

        if args.unchanged_evaluator_loop:
            # ... and with dropin.install(patch_evaluator=False): the reference's OWN loop body (evaluation/methods.py:507-540 restated call
            # for call in tools/reference_sequence.py) — split_size rays per chunk, an upload, model.render and six .cpu() read-backs each
            sys.path.insert(0, os.path.join(REPO, "tools"))
            import reference_sequence

            def full_view():
                return reference_sequence.reference_render_view(model, pose_h, uv_h, K_h, (h, w), 0, split_size=chunk, device=dev)
        else:
            def full_view():
                return evaluator.render_view(model, pose_h, uv_h, K_h, 0, split_size=chunk, n_streams=args.streams)
    else:
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
        if args.no_parity:
            emit_line(({"metric": "rays/sec (full 1200x680 view) — throughput only (--no-parity)", "value": round(n * args.steps / elapsed, 1), "unit": "rays/s",
                        "n_gpus": 1, "steps": args.steps, "warmup": max(1, args.warmup // 3), "ms_per_step": round(elapsed / args.steps * 1e3, 2),
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16x3+f32acc" if args.precision == "f16x3" else "f32",
                        "data": "synthetic",
                        "config": {"workload": f"full view {w}x{h} in {chunk}-ray chunks x {s_c} + {n_f} samples, {args.streams} stream(s)" +
                                               ((f", the reference evaluator's UNCHANGED loop (dropin.install(patch_evaluator=False): per {chunk}-ray chunk an upload, "
                                                 f"model.render and six .cpu() read-backs)" if args.unchanged_evaluator_loop else
                                                 f", as the evaluator runs it (host in / host out; grouped into >= {evaluator.MIN_CHUNK}-ray chunks)") if args.as_evaluator else "")}}))
            return

        # parity image: same camera at 1/8 resolution on both sides, identical u_add draw (Q9)
        ws, hs = w // 8, h // 8
        uv_s, pose_s, K_s = synthetic.pinhole_image(ws, hs, f / 8.0, device=dev)
        g = torch.Generator().manual_seed(21)
        uni = {"u_add": torch.rand(ws * hs, n_f, generator=g)}
        o = model.render(pose_s, uv_s, K_s, epoch=0, uniforms=uni)
        sig = model.get_density(o.coarse_normals, o.ray_dirs).cpu()          # the density kernel's own sigma on this render's normals
        model.precision = "fp32"                       # the exact-fp32 HIP kernels on the same rays: the second column of the accounting
        o32 = model.render(pose_s, uv_s, K_s, epoch=0, uniforms=uni)
        sig32 = model.get_density(o32.coarse_normals, o32.ray_dirs).cpu()
        model.precision = args.precision
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        settings = O.RenderSettings(n_samples=s_c, n_fine=n_f, perturb=False, dir_to_normal_th=-0.2, fine_range=0.3,
                                    density=O.DensityParams(scale_min=1.0))
        vf_sd, rn_sd = _oracle_inputs(model)
        t1 = time.perf_counter()
        ref = O.render(uv_s.cpu(), pose_s.cpu(), K_s.cpu(), vf_sd, rn_sd, settings, **uni)
        cpu_s = time.perf_counter() - t1
        # the yardstick: the SAME oracle in float64 (every tensor promoted; nothing else changed).  The distance of the fp32 oracle
        # from it is what "the reference's own fp32 arithmetic" can claim on these rays — a density threshold (Laplace scale 100
        # on a windowed cosine) turns a 1e-7 difference in a normal into a different sigma wherever a sample sits on the edge
        dbl = lambda sd: {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}     # noqa: E731
        t2 = time.perf_counter()
        ref64 = None
        if not args.no_float64:
            ref64 = O.render(uv_s.cpu().double(), pose_s.cpu().double(), K_s.cpu().double(), dbl(vf_sd), dbl(rn_sd), settings,
                             u_add=uni["u_add"].double())
        cpu64_s = time.perf_counter() - t2

    def account(z, rgb_a, dep_a, zb, rgb_b, dep_b, exact_z=True):
        # (float64 depths are not comparable bit for bit: "same sampling" = every depth within 1e-6)
        return ray_accounting(z, rgb_a.float(), dep_a.float().reshape(-1, 1), zb, rgb_b.float(), dep_b.float().reshape(-1, 1),
                              z_tol=0.0 if exact_z else 1e-6)

    rgb, depth = o.coarse_rgb_values.cpu(), o.coarse_depth_map.cpu()
    rgb32, depth32 = o32.coarse_rgb_values.cpu(), o32.coarse_depth_map.cpu()
    f64 = {}
    if ref64 is not None:
        r64_rgb, r64_dep, r64_z = ref64["rgb"], ref64["depth"], ref64["z_vals"]
        f64 = {"oracle_f32_vs_oracle_f64": account(ref["z_vals"], ref["rgb"], ref["depth"], r64_z, r64_rgb, r64_dep, exact_z=False),
               "hip_default_vs_oracle_f64": account(o.z_vals.cpu(), rgb, depth, r64_z, r64_rgb, r64_dep, exact_z=False),
               "hip_exact_fp32_vs_oracle_f64": account(o32.z_vals.cpu(), rgb32, depth32, r64_z, r64_rgb, r64_dep, exact_z=False)}
    # every ray outside the contract, with what puts it there
    margins = {"hip_default_vs_oracle_f32": ray_error_sources(rgb, depth, o.coarse_normals.cpu(), sig, ref, settings, O),
               "hip_exact_fp32_vs_oracle_f32": ray_error_sources(rgb32, depth32, o32.coarse_normals.cpu(), sig32, ref, settings, O)}
    if ref64 is not None:     # the yardstick: the oracle's own fp32 evaluation against its float64 one, same accounting
        ref64f = {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ref64.items()}
        margins["oracle_f32_vs_oracle_f64"] = ray_error_sources(ref["rgb"], ref["depth"], ref["normals"], ref["sigma"], ref64f, settings, O)
    emit_line(({
        "metric": "rays/sec (full 1200x680 view, 1024-ray chunks, 128 samples/ray) + PSNR/depth vs ref",
        "value": round(n * args.steps / elapsed, 1), "unit": "rays/s", "n_gpus": 1, "steps": args.steps,
        "warmup": max(1, args.warmup // 3), "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3+f32acc" if args.precision == "f16x3" else "f32", "data": "synthetic",
        "config": {"workload": f"full view {w}x{h} = {n} rays in {chunk}-ray chunks x {s_c + n_f} samples, perturb off, "
                               f"forward only, consecutive chunks on {args.streams} stream(s) (BASELINE.json configs[1])" +
                               (f", as the evaluator runs it: host tensors in (pose / intrinsics per ray), host rgb / depth out, uploads and "
                                f"download inside the timed region (evaluator.render_view: split_size {chunk}, grouped into chunks of >= {evaluator.MIN_CHUNK} rays)"
                                if args.as_evaluator else ", inputs resident in HBM"),
                   "colour_products": int(model.colour_products) if args.precision == "f16x3" else None},
        "parity_vs_oracle": {"image": f"{ws}x{hs} (same camera, intrinsics / 8), {ws * hs} rays",
                             "psnr_rgb_db": round(min(O.psnr(rgb, ref["rgb"]), 200.0), 2),
                             "argmax_indices_equal": bool((o.z_vals.cpu() == ref["z_vals"]).all()),
                             "oracle_seconds": round(cpu_s, 1), "oracle_float64_seconds": round(cpu64_s, 1),
                             # who is how far from whom, per ray (contract: 1e-4)
                             "hip_default_vs_oracle_f32": account(o.z_vals.cpu(), rgb, depth, ref["z_vals"], ref["rgb"], ref["depth"]),
                             "hip_exact_fp32_vs_oracle_f32": account(o32.z_vals.cpu(), rgb32, depth32, ref["z_vals"], ref["rgb"], ref["depth"]),
                             **f64,
                             # ... and WHY the rays outside the contract are outside (ray_error_sources)
                             "out_of_tolerance_rays": margins}}))


def reference_lattice(res: int, scale: float = 1.0, translation=(0.0, 0.0, 0.0), centroid=(0.0, 0.0, 0.0)) -> torch.Tensor:
    """The host grid of evaluation/methods.py:190-208 (marching_cubes_mesh), built with the same fp32 operations in the same order:
    index * voxel_size + voxel_origin + translation + centroid per column, row (i res + j) res + k = cell (i, j, k)."""
    voxel_origin = [-scale, -scale, -scale]
    voxel_size = scale * 2.0 / (res - 1)
    translation, centroid = torch.tensor(translation, dtype=torch.float32), torch.tensor(centroid, dtype=torch.float32)
    overall_index = torch.arange(0, res ** 3, 1, dtype=torch.long)
    samples = torch.zeros(res ** 3, 3)
    samples[:, 2] = overall_index % res
    samples[:, 1] = (overall_index // res) % res
    samples[:, 0] = ((overall_index // res) // res) % res
    samples[:, 0] = (samples[:, 0] * voxel_size) + voxel_origin[2] + translation[0] + centroid[0]
    samples[:, 1] = (samples[:, 1] * voxel_size) + voxel_origin[1] + translation[1] + centroid[1]
    samples[:, 2] = (samples[:, 2] * voxel_size) + voxel_origin[0] + translation[2] + centroid[2]
    return samples


def grid_bench(args, dev, rank, world, dist, sync):
    """BASELINE.json configs[4]: dense-grid queries of the vector field (marching-cubes input): the res^3 points of one quadrant through
    ``grid.get_set_predictions`` — the host grid evaluation/methods.py:190-208 builds in, the host [res^3, 3] predictions out — blocks of
    100 000 points dealt round-robin to the ranks.  A step = one res^3 quadrant.  `value` = the default path (the separable lattice is
    regenerated on the device from its axis tables while the host verifies every row; pinned download overlapped with the launches);
    ``upload_path_points_per_s`` = the same call with the grid uploaded (what a non-lattice ``samples`` tensor takes);
    ``device_resident_points_per_s`` = grid and predictions in HBM.  ``roofline``: the vector-only launch (0.919 MFLOP per point) timed
    with HIP events on its launch stream during the device-resident run."""
    from vf_nerf_amd import grid
    from oracle import vfnerf_oracle as O
    model, _, _, _ = build_scene(dev, 16, 64, 64, seed=0)
    model.precision = args.precision
    dec = model.fine_vector_field_network
    res = args.grid_res
    samples = reference_lattice(res)
    n = samples.shape[0]

    n_warm = max(2, args.warmup // 3)

    def timed(fast: bool):
        # (two untimed calls at least: the [n,3] result is page-locked host memory — 0.14 s to pin 1.6 GB at 512^3 — which PyTorch's host
        # allocator caches; the caller holds one result while the next call fills another, so the pool is warm after two calls.  The
        # evaluator's eight quadrants per mesh pay that once; ``first_call_ms`` is what the very first call costs.)
        grid.LATTICE_FAST_PATH = fast
        first = None
        for w in range(n_warm):
            torch.cuda.synchronize()
            tw = time.perf_counter()
            keep_w = grid.get_set_predictions(dec, samples, 100000, dev, rank=rank, world_size=world)
            torch.cuda.synchronize()
            first = first if first is not None else (time.perf_counter() - tw) * 1e3
        del keep_w
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            got = grid.get_set_predictions(dec, samples, 100000, dev, rank=rank, world_size=world)
        sync()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, got, grid.last_path, first

    elapsed_up, got_up, path_up, first_up = timed(False)
    elapsed, got, path, first_ms = timed(True)
    same = bool(torch.equal(got, got_up))
    del got_up
    # device-resident rate (grid already in HBM, no host copies) with HIP events around the launches: what the kernel itself sustains
    dsamples = samples[: min(n, 1 << 24)].to(dev)
    grid.get_set_predictions(dec, dsamples, 100000, dev)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    e0.record()
    grid.get_set_predictions(dec, dsamples, 100000, dev)
    e1.record()
    torch.cuda.synchronize()
    resident = dsamples.shape[0] / (time.perf_counter() - t1)
    kernel_s = e0.elapsed_time(e1) * 1e-3
    if rank == 0:
        vf_sd, _ = _oracle_inputs(model)
        idx = torch.arange(0, n, max(1, n // 4096))[:4096]
        idx = idx[(idx // 100000) % world == 0]                      # rows this rank evaluated
        ref = O.vf_mlp(samples[idx], vf_sd, 6, (4,))[:, :3]
        err = float((got[idx] - ref).abs().max())
        f16 = args.precision == "f16x3"
        macs = VF_MACS - 256 * 256                                    # vector head only: the 256 x 256 feature block is never evaluated
        flops = 2.0 * macs * dsamples.shape[0]
        peak = PEAK_F16_MFMA / 3.0 if f16 else PEAK_F32_MFMA
        line = {
            "metric": "grid points/sec (vector-field queries for quadrant marching cubes)",
            "value": round(n * args.steps / elapsed, 1), "unit": "points/s", "n_gpus": world, "steps": args.steps,
            "warmup": n_warm, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f16x3+f32acc" if f16 else "f32", "data": "synthetic",
            "first_call_ms": {"upload_path_cold_process": round(first_up, 1), "lattice_path": round(first_ms, 1),
                              "note": "the first call of a process page-locks its result buffer (and a second one while the caller still holds the first)"},
            "config": {"workload": f"{res}^3 = {n} grid points per quadrant, the host grid of evaluation/methods.py:190-208 in, host [n,3] out, "
                                   f"max_batch 100000 (BASELINE.json configs[4], one quadrant)",
                       "parallelism": f"blocks x{world}"},
            "input_path": path, "upload_path": path_up,
            "upload_path_points_per_s": round(n * args.steps / elapsed_up, 1),
            "lattice_and_upload_paths_bit_identical": same,
            "device_resident_points_per_s": round(resident, 1),
            "max_abs_err_vs_oracle_4096_points": err,
            "roofline": {"bound": "mfma", "kernel": "vfn_mlp16_kernel<M16_VF> (vector-field net, vector head only)" if f16 else "vfn_mlp_kernel (vector head only)",
                         "achieved": round(flops / kernel_s / 1e12, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(flops / kernel_s / 1e12 / peak, 4), "traffic": None,
                         "flops_per_point": 2.0 * macs, "points": int(dsamples.shape[0]), "launch_span_ms": round(kernel_s * 1e3, 3),
                         "algorithmic_bytes_per_point": 24,
                         "timing": "HIP events on the launch stream around the device-resident call (its launches back to back)",
                         "peak_definition": "dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-equivalent product" if f16 else "fp32 MFMA 157.3 TFLOP/s"}}
        if world == 1 and not args.no_cpu_baseline:
            threads = min(32, os.cpu_count() or 1)
            torch.set_num_threads(threads)
            sub = samples[: 200000]
            with torch.no_grad():
                O.vf_mlp(sub, vf_sd, 6, (4,))
                reps, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 10.0 and reps < 50:
                    O.vf_mlp(sub, vf_sd, 6, (4,))
                    reps += 1
                el = time.perf_counter() - t0
            line["cpu_baseline"] = {"value": round(sub.shape[0] * reps / el, 1), "unit": "points/s", "cores": threads, "kind": "port",
                                    "sample": f"{reps} x the oracle's vector-field forward (decoder(x)[:, :3] of mc_utils.py:100, all 259 columns "
                                              f"evaluated as the reference does) on {sub.shape[0]} grid points, torch fp32 CPU, {threads} threads, {el:.1f} s"}
        emit_line(line)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def grid_stages_bench(args, dev):
    """the dense-grid stages between the vector-field queries and the triangulation (evaluation/methods.py:
    210-253 -> mc_utils.extract_divergence / unify_direction / make_comb_format, guassian_smoothing.smooth_vf) on a res^3 grid whose
    field is the scene's own vector field (queried on the device first).  Every stage is HBM-bound: reported as ALGORITHMIC bytes per
    cell (each grid value in and out once) / average launch time (HIP events) against the 8 TB/s HBM peak."""
    from vf_nerf_amd import grid, lib as vlib
    model, _, _, _ = build_scene(dev, 16, 64, 64, seed=0)
    model.precision = args.precision
    dec = model.fine_vector_field_network
    reps = max(3, args.steps // 2)

    def timed(fn, reps=reps):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def field(res):
        ax = torch.linspace(-1.0, 1.0, res, device=dev)
        pts = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3).contiguous()
        pred = grid.get_set_predictions(dec, pts, 100000, dev)
        del pts
        return pred

    out = {}
    stages = []
    for res in sorted({args.grid_res, args.unify_res}):
        pred = field(res)
        cells = res ** 3
        rec = {}
        if res == args.grid_res:
            div = None

            def f_div():
                nonlocal div
                div = vlib.grid_divergence(pred, res, -0.5)
            ms = timed(f_div)
            rec["divergence"] = {"ms": round(ms, 4), "bytes_per_cell": 16, "tb_per_s": round(cells * 16 / ms / 1e9, 3)}
            w9 = grid.gaussian_weights(9, 2.0)
            tmp = torch.empty_like(pred)
            for axis in (0, 1, 2):
                ms = timed(lambda: vlib.grid_smooth_axis(pred, tmp, res, axis, w9))
                rec[f"smooth_k9_axis{axis}"] = {"ms": round(ms, 4), "bytes_per_cell": 24, "tb_per_s": round(cells * 24 / ms / 1e9, 3)}
            w3 = grid.gaussian_weights(3, 1.0)
            ms3 = sum(timed(lambda: vlib.grid_smooth_axis(pred, tmp, res, axis, w3)) for axis in (0, 1, 2))
            rec["smooth_k3_three_axes"] = {"ms": round(ms3, 4), "bytes_per_cell": 72, "tb_per_s": round(cells * 72 / ms3 / 1e9, 3)}
            ms9 = sum(rec[f"smooth_k9_axis{a}"]["ms"] for a in (0, 1, 2))
            rec["smooth_k9_three_axes"] = {"ms": round(ms9, 4), "bytes_per_cell": 72, "tb_per_s": round(cells * 72 / ms9 / 1e9, 3)}
            rec["surface_cell_fraction"] = round(float(div.mean()), 5)
            del tmp
        if res == args.unify_res:
            div = vlib.grid_divergence(pred, res, -0.5)
            vt = torch.nn.functional.normalize(pred, dim=1)
            norms = torch.norm(pred, dim=1)
            sides = choice = None

            def f_uni():
                nonlocal sides, choice
                sides, choice = vlib.grid_unify_direction_sides(div.reshape(-1), vt, res)
            ms = timed(f_uni)
            rec["unify_direction"] = {"ms": round(ms, 4), "bytes_per_cell": 69, "tb_per_s": round(cells * 69 / ms / 1e9, 3),
                                      "note": "4 B mask in, 64 B int64 table + 1 B side byte out; corner vectors only for surface cells"}
            ms = timed(lambda: vlib.grid_comb_format_sides(sides, norms, res), reps=3)
            rec["make_comb_format_from_side_bytes"] = {"ms": round(ms, 4), "bytes_per_cell": 341, "tb_per_s": round(cells * 341 / ms / 1e9, 3)}
            ms = timed(lambda: vlib.grid_comb_format(choice, norms, res), reps=3)
            rec["make_comb_format_from_int64_table"] = {"ms": round(ms, 4), "bytes_per_cell": 404, "tb_per_s": round(cells * 404 / ms / 1e9, 3)}
            rec["surface_cell_fraction"] = round(float(div.mean()), 5)
            del vt, norms, sides, choice
        out[f"res_{res}"] = rec
        stages += [v["tb_per_s"] for v in rec.values() if isinstance(v, dict)]
        del pred, div
        torch.cuda.empty_cache()
    best_div = out[f"res_{args.grid_res}"]["divergence"]
    emit_line(({"metric": "dense-grid stage throughput (algorithmic HBM bytes / launch time)", "value": best_div["tb_per_s"], "unit": "TB/s",
                      "n_gpus": 1, "steps": reps, "warmup": 1, "ms_per_step": best_div["ms"], "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": f"extract_divergence / smooth_vf at {args.grid_res}^3, unify_direction / make_comb_format at "
                                             f"{args.unify_res}^3, field = the scene's vector field on the grid (SURVEY.md section 8f N3)"},
                      "roofline": {"bound": "hbm", "achieved": best_div["tb_per_s"] * 1e3, "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(best_div["tb_per_s"] / 8.0, 4), "traffic": None, "kernel": "vfn_grid_divergence_kernel"},
                      "stages": out}))


def train_step_accounting(n_rays, s_c, s_t, n_sup, storage, gradients="fp32", separate_proposal=False, selected=None):
    """ALGORITHMIC work and workspace traffic of one training step (DESIGN.md section 3 "Backward").
    FLOPs: the fine pass forward and twice that for its backward (VF + rendering, S_t samples) + forward and backward of the
    2 x n_sup supervision points through the VF net (+ the reference's separate vector-only proposal pass over the S_c samples
    only when it is actually run: with one VF evaluation per distinct sample the proposal samples are part of the fine pass).
    Bytes: what the kernels of the 16-bit path have to move through HBM per step — the saved activations (13 slots per fine
    point, 9 per supervision point; 1 KiB per slot and point as fp32, 512 B as f16 except the tanh'ed feature slot), their
    sign-bit words (32 B), the pre-activation gradients dY (1 KiB per slot and point, written by the chain, read by the
    weight-gradient kernels), the saved activations read once by the weight-gradient kernels, and the per-sample inputs and
    outputs (point, normal, colour, their gradients).
    ``selected`` (the sparse colour branch): the fraction of the samples with non-zero weight.  The bytes are then
    what THAT step moves — every sample and supervision point through the vector-only launches (the 8 ReLU slots of the vector-field
    net: no feature slot, no rendering-net slots), the selected samples once more through all 13 slots — while the FLOPs stay the
    dense step's (algorithmic work; the caller prices what was executed separately)."""
    m_f, m_s = n_rays * s_t, 2 * n_sup
    # (supervision points: the three vector columns only — trainer.TrainStep's vector-only call; the 256 x 256 feature block of the
    # reference's full forward there is never read and is not counted)
    flops = 2.0 * ((n_rays * s_c * VF_MACS if separate_proposal else 0.0) + 3.0 * m_f * (VF_MACS + RN_MACS) + 3.0 * m_s * (VF_MACS - 256 * 256))
    relu_slot = 512 if storage == "f16" else 1024
    saved = m_f * (12 * relu_slot + 1024) + m_s * (8 * relu_slot + 1024)      # written by the forward ...
    masks = 32 * (13 * m_f + 9 * m_s)
    dy = (512 if gradients in ("bf16", "f16") else 1024) * (13 * m_f + 9 * m_s)
    small = (12 + 12 + 12 + 4 + 12 + 12) * m_f + 2 * 160 * (m_f + m_s)        # points, normals, colours, z + grads, aux tiles
    if selected is not None:
        m_1, m_2 = m_f + m_s, selected * m_f                                   # region 1 (vector-only), region 2 (fused, the selected samples)
        dy_slot = 512 if gradients in ("bf16", "f16") else 1024
        saved = m_1 * 8 * relu_slot + m_2 * (12 * relu_slot + 1024)
        masks = 32 * (8 * m_1 + 13 * m_2)
        dy = dy_slot * (8 * m_1 + 13 * m_2)
        small += (12 + 12 + 12 + 12) * m_2 + 2 * 160 * m_2
    total = 2 * saved + 2 * masks + 2 * dy + small                             # ... and read back once; dY written + read
    return flops, total


def train_dtype_label(model) -> str:
    """What the training step computes in, said as it is: forward f16x3 (three f16 products, fp32 accumulate), dX chain bf16x3; the
    weight gradients dW = dY^T X from the STORED operands — activations f16 or fp32, gradients per-lane-scaled f16 / bf16 / fp32 —
    on one to three products per K block."""
    if model.precision != "f16x3":
        return "f32 (exact fp32 MFMA kernels)"
    act, grad = model.activation_storage, model.gradient_storage
    if getattr(model, "training_products", 3) == 1 and (act, grad, model.workspace_layout) == ("f16", "f16", "fragment"):
        return ("16-bit-native (opt-in, outside the 1e-4 contract): f16 fwd (one product per K block, 11-bit operands) + bf16 dX chain (one "
                "product, 8-bit operands) + dW on one f16 product; f32 accumulate; activations stored f16, gradients stored scaled f16")
    dw = {("f16", "f16"): "one f16 product (f16 activations x per-lane-scaled f16 gradients, 11 x 11 bits)",
          ("f16", "bf16"): "two bf16 products (f16 activations as bf16 hi+lo x bf16 gradients)",
          ("f16", "fp32"): "three bf16 products (f16 activations x fp32 gradients)",
          ("fp32", "f16"): "two f16 products (fp32 activations as f16 hi+lo x scaled-f16 gradients)"}.get((act, grad), "three bf16 products (fp32-equivalent)")
    return f"f16x3 fwd + bf16x3 dX chain + dW on {dw}; f32 accumulate; activations stored {act}, gradients stored {grad}"


def training_targets(model, uv, pose, K, dev, s_c, n_f, rank=0):
    """(rgb_gt, depth_gt, centroid, border radius) of a training-step workload on these rays.
    TRAINED weights (the model came from build_trained_scene): the targets are the model's OWN deterministic render of the rays
    (exact-fp32 kernels), so the steps run at the state a long training run sits in — the scene stays, and with it the fraction of
    samples that carry weight.  Random weights: a LEARNABLE target, rgb / depth as a teacher model of
    another weight seed renders them — note that this scene loses its surfaces within two Adam steps (every weight becomes zero),
    so a step timed there flatters the sparse colour branch."""
    if getattr(model, "_bench_trained_weights", None) is not None:
        keep = (model.precision, model.ray_sampler.deterministic, model.fine_sampler.deterministic)
        model.precision, model.ray_sampler.deterministic, model.fine_sampler.deterministic = "fp32", True, True
        with torch.no_grad():
            t_out = model.render(pose, uv, K, epoch=0)
        model.precision, model.ray_sampler.deterministic, model.fine_sampler.deterministic = keep
        # (the recipe the fixture was trained with: tests/golden/make_trained_golden.py)
        return t_out.coarse_rgb_values.clone(), t_out.coarse_depth_map.clone(), (0.0, 0.0, 0.55), 0.15
    teacher, _, _, _ = build_scene(dev, 16, s_c, n_f, seed=rank, perturb=False, weight_seed=1)
    teacher.precision = "fp32"
    with torch.no_grad():
        t_out = teacher.render(pose, uv, K, epoch=0)
    return t_out.coarse_rgb_values.clone(), t_out.coarse_depth_map.clone(), (0.0, 0.0, 0.6), 0.05


def drop_in_sequence_timing(args, model, uv, pose, K, rgb_gt, depth_gt, centroid, radius, sync):
    """The training step issued as the reference trainer's OWN call sequence (train/vector_field_nerf_train.py:177-275: render, the samplers,
    the two network calls, VFLoss, zero_grad, backward, clip_grad_norm_, optimizer.step, scheduler.step, loss.item() — restated call for
    call in tools/reference_sequence.py on the names vf_nerf_amd.dropin installs), timed like the one-call step above: it takes the step
    session (stepengine.py), i.e. the same workspace and kernels, with ~15 Python-level calls around them.  ``ms_per_step`` includes the
    loop's per-step ``loss.item()`` / ``losses_dict[key]`` reads and its running sums, exactly as train_epoch keeps them — on the HIP path
    those are deferred scalars (vf_nerf_amd/deferred.py: added on the device, read once per epoch), so the loop no longer synchronises per
    step; ``ms_per_step_without_loss_item`` leaves the reads out; ``ms_per_step_with_a_synchronising_item`` is the same loop with plain
    floats (``loss.DEFERRED_SCALARS = False``: rounds 1-4's behaviour, a device synchronisation per step)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from types import SimpleNamespace
    import reference_sequence
    from vf_nerf_amd import loss as vloss, stepengine, trainer
    crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
    data = {"uv": uv.unsqueeze(0), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0), "rgb": rgb_gt.unsqueeze(0), "depth": depth_gt.unsqueeze(0)}
    out = {}
    eng = stepengine.StepEngine.of(model)
    keep = vloss.DEFERRED_SCALARS
    for name, item, deferred in (("ms_per_step", True, True), ("ms_per_step_without_loss_item", False, True),
                                 ("ms_per_step_with_a_synchronising_item", True, False)):
        vloss.DEFERRED_SCALARS = deferred
        try:
            loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(centroid, 1.0), radius, sync_each_step=item)
            for _ in range(max(3, args.warmup)):
                loop(data, 0)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                loop(data, 0)
            torch.cuda.synchronize()
            out[name] = round((time.perf_counter() - t0) / args.steps * 1e3, 4)
            if item and deferred:       # the epoch's end: the division that turns the running sums into floats (train.py:277-280)
                avg = loop.average_losses
                out["running_loss_mean_read_at_the_end"] = round(float(avg["loss"]) / (max(3, args.warmup) + args.steps), 6)
        finally:
            vloss.DEFERRED_SCALARS = keep
    out["took_the_step_session"] = eng.why_not is None and eng.session is not None
    out["why_not"] = eng.why_not
    out["call_sequence"] = "render | sample_border_points | vector_field_network(p)[:, :3] | get_center_indices_and_gt | sample_center_points | " \
                           "vector_field_network(p)[:, :3] | VFLoss | zero_grad | backward | clip_grad_norm_ | optimizer.step | scheduler.step | loss.item()"
    return out


def train_bench(args, model, uv, pose, K, dev, dist, rank, world, sync, emit=True):
    """One step = what the reference trainer does per batch, with synthetic targets (config 3 of BASELINE.json)."""
    from vf_nerf_amd import distributed as vdist, supervision, trainer
    supervision.manual_seed(0x5eed + 7919 * (rank + 1))     # every rank draws its own supervision points
    s_t = args.coarse + args.fine
    trained = getattr(model, "_bench_trained_weights", None)
    rgb_gt, depth_gt, centroid, radius = training_targets(model, uv, pose, K, dev, args.coarse, args.fine, rank)
    n_sup = (args.rays * s_t) // 10
    bucket = vdist.GradientBucket(model) if (world > 1 or dist is not None) else None
    # the reference trainer's loop body (train/vector_field_nerf_train.py:172-260) on the shipped loss / supervision settings
    run_step = trainer.TrainStep(model, centroid, border_radius=radius, far=1.0, bucket=bucket)
    counts = []

    def step():
        return run_step(pose, uv, K, rgb_gt, depth_gt, epoch=0)[0]

    for _ in range(args.warmup):
        step()
    sync()
    if run_step.last_colour_counts is not None:
        counts.append(run_step.last_colour_counts.tolist())
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    local_elapsed = time.perf_counter() - t0
    if run_step.last_colour_counts is not None:
        counts.append(run_step.last_colour_counts.tolist())
    sync()
    elapsed = time.perf_counter() - t0
    _, rates = rank_rates(dist, local_elapsed, args.rays * args.steps, dev)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # the step's one collective on its own (outside the timed region): the flat 805 780-element fp32 bucket, sum + divide
    bucket_ms = None
    if bucket is not None and dist is not None and dist.get_world_size() > 1:
        for _ in range(3):
            bucket.all_reduce_mean()
        sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            bucket.all_reduce_mean()
        e1.record()
        torch.cuda.synchronize()
        bucket_ms = e0.elapsed_time(e1) / 20.0
    drop_in = drop_in_sequence_timing(args, model, uv, pose, K, rgb_gt, depth_gt, centroid, radius, sync) if (bucket is None and not model.vector_field_network.training) else None
    from vf_nerf_amd.backward import StoredFinePass
    stored = model.reuse_proposal and StoredFinePass.applicable(model, args.rays, args.coarse, s_t - args.coarse)
    flops, ws_bytes = train_step_accounting(args.rays, args.coarse, s_t, n_sup, model.activation_storage, model.gradient_storage,
                                            separate_proposal=not stored)
    ms = elapsed / args.steps * 1e3
    # what the step actually executed: with the one-call step's sparse colour branch (exact: DESIGN.md section 1) the colour branch's
    # share of the algorithmic FLOPs runs on the selected fraction of the samples only
    one_call = run_step.one_call.why_not is None
    sparse = one_call and bool(getattr(model, "sparse_colour_training", True))
    sel = None
    if sparse and counts:
        sel = sum(c[0] / max(1.0, c[1]) for c in counts) / len(counts)        # (before and after the timed steps)
    colour_share = 3.0 * args.rays * s_t * 2.0 * (256 * 256 + RN_MACS) / flops   # fraction of the step's dense FLOPs in the colour branch (feature block + rendering net)
    executed = flops * (1.0 - colour_share * (1.0 - sel)) + (3.0 * 2.0 * args.rays * s_t * sel * (VF_MACS - 256 * 256) if sel is not None else 0.0) \
        if sel is not None else flops
    ws_dense = ws_bytes
    if sel is not None:          # the bytes the sparse step moves (rounds 4's lines carried the DENSE step's accounting whatever ran)
        ws_bytes = train_step_accounting(args.rays, args.coarse, s_t, n_sup, model.activation_storage, model.gradient_storage,
                                         separate_proposal=not stored, selected=sel)[1]
    rec = {"metric": "training rays/sec (4096-ray batch, 128 samples/ray, fwd+bwd+clip+Adam)",
           "value": round(args.rays * args.steps * world / elapsed, 1), "unit": "rays/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": train_dtype_label(model) if not model.vector_field_network.training else
           ({"split": "f16x3 forward products + bf16x3 dX + bf16x3 dW; f32 accumulate and fp32 activations in HBM",
             "split24": "f16x3 forward products + bf16 in three parts (six products) dX + bf16x3 dW; f32 accumulate and fp32 activations in HBM"}
            .get(getattr(model.vector_field_network, "gemm_arithmetic", "split"), "f32 (exact fp32 matrix instruction)")),
           "data": "synthetic", "final_loss": round(float(loss), 5), "per_rank_rays_per_s": rates,
           "bucket_allreduce_ms": round(bucket_ms, 4) if bucket_ms is not None else None,
           "bucket_elements": bucket.numel() if bucket is not None else None,
           "activation_storage": model.activation_storage, "gradient_storage": model.gradient_storage,
           "workspace_layout": model.workspace_layout, "training_products": int(getattr(model, "training_products", 3)),
           "networks": "training mode (batch-statistics BatchNorm: one product per layer with the statistics / the BatchNorm backward's sums in its "
                       "epilogue, HBM-bound row passes in between; csrc/vfn_bstat.hip)"
           if model.vector_field_network.training else "eval mode (the shipped regime, fused kernels)",
           # per GPU: algorithmic FLOPs of a step / its duration against the f16 / 3 matrix ceiling, and the workspace bytes the
           # step has to move against the HBM peak (it sits between the two roofs)
           "step_issued_as": "one C call (vfn_train_step)" if one_call else f"launch by launch from Python ({run_step.one_call.why_not})",
           # the SAME step issued the way the reference's unchanged train_epoch issues it through vf_nerf_amd.dropin
           "drop_in_sequence_ms": drop_in["ms_per_step"] if drop_in else None,
           "drop_in_sequence": drop_in,
           "weights": trained if trained is not None else {"fixture": None, "trained_by": "nobody: synthetic random weights; targets rendered by a teacher of another seed"},
           "sparse_colour_branch": {"on": sparse, "samples_with_nonzero_weight": round(sel, 4) if sel is not None else None,
                                    "before_and_after_the_timed_steps": [round(c[0] / max(1.0, c[1]), 4) for c in counts] if sparse else None,
                                    "note": "exact: a sample's colour, its gradient and the rendering net's share of it in the weight gradients are needed only "
                                            "where its weight is non-zero; the selected samples re-run the vector-field trunk in the fused launch"},
           # algorithmic = the DENSE step's FLOPs, whatever was skipped as exactly zero; executed = what the launches computed
           "algorithmic_tflop_per_step": round(flops / 1e12, 4),
           "executed_tflop_per_step": round(executed / 1e12, 4),
           "executed_tflops": round(executed / (ms * 1e-3) / 1e12, 1),
           "frac_of_f16_mfma_div3_executed": round(executed / (ms * 1e-3) / 1e12 / (PEAK_F16_MFMA / 3.0), 4),
           # (a fraction is quoted for EXECUTED work only: pricing the dense step's algorithmic FLOPs — the reference's dense colour branch,
           # 1.5 VF evaluations per sample — against the roof would be a fraction of work the step does not do.  The opt-in single-product
           # mode runs one matrix product per algorithmic one: its ceiling is the whole f16 peak.)
           "frac_of_f16_mfma_executed_single_product": round(executed / (ms * 1e-3) / 1e12 / PEAK_F16_MFMA, 4) if train_dtype_label(model).startswith("16-bit-native") else None,
           # bytes the step's launches move through the training workspace (accounted, not counted): with the sparse colour branch the
           # vector-only slots of every point + all slots of the selected samples; `_dense_step` = what the dense step moves (the figure
           # rounds 3-4 quoted for every step)
           "workspace_gb_per_step": round(ws_bytes / 1e9, 2),
           "workspace_gb_per_step_dense_step": round(ws_dense / 1e9, 2),
           "workspace_tb_per_s": round(ws_bytes / (ms * 1e-3) / 1e12, 3),
           "frac_of_hbm_peak": round(ws_bytes / (ms * 1e-3) / 8e12, 4),
           # what the opt-in two-product colour branch would do on the weights these steps arrived at (raw difference to three products,
           # and the guard's verdict); gradient-free renders of a trained model are where that mode would be used
           "two_product_check_after_these_steps": two_product_check(model, uv, pose, K, rays=min(1024, args.rays)) if model.uses_f16x3() and not model.vector_field_network.training else None,
           "config": {"workload": f"train step: render({args.rays} rays x {s_t}) + 2x{n_sup} supervision "
                                  f"points through the VF net + L1/depth/unit-norm/supervision loss + "
                                  f"backward + clip_grad_norm_ + Adam (sequential semantics over the duplicated parameter list)"
                                  + (", gradients all-reduced over one flat bucket" if bucket is not None else "")}}
    if not emit:
        return rec
    if rank == 0:
        emit_line((rec))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rec


_REAL_STDOUT = None


def claim_stdout() -> None:
    """The contract is ONE JSON line on stdout.  Libraries below Python write there too (RCCL prints its version banner through C
    stdio when the process exits, i.e. AFTER the line): the process's stdout (fd 1) is pointed at stderr for everything else, and
    the line goes out through a private duplicate of the original descriptor."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(rec) -> None:
    data = (json.dumps(rec) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher around it: start N ranks as CHILD processes (torch.distributed.run; never an
    exec, and before this process has made any GPU call), wait, relay rank 0's JSON line.  The reference's multi-GPU entry is
    automatic too (models/nerf/vector_field_nerf.py:70-75 wraps the modules in nn.DataParallel when num_gpus > 1,
    config_parser/vf_nerf_config_parser.py:88); here it is one process per GPU over RCCL."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC (RCCL across processes on this driver)
    env["VFN_BENCH_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout.splitlines():
        try:
            rec = json.loads(out)
            if isinstance(rec, dict) and "metric" in rec:
                line = out
                continue
        except ValueError:
            pass
        print(out, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"bench.py: the {n}-rank run failed (exit code {proc.returncode}, JSON line {'present' if line else 'missing'})", file=sys.stderr)
        return proc.returncode or 1
    print(line, flush=True)
    return 0


def rank_rates(dist, elapsed_local: float, units_per_rank: float, dev):
    """(max elapsed over ranks, {per-rank units/s min / max / world size as the process group reports it})."""
    if dist is None:
        return elapsed_local, {"min": round(units_per_rank / elapsed_local, 1), "max": round(units_per_rank / elapsed_local, 1), "dist_world_size": 1}
    world = dist.get_world_size()
    mine = torch.tensor([elapsed_local], device=dev, dtype=torch.float64)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    times = [float(t.item()) for t in every]
    return max(times), {"min": round(units_per_rank / max(times), 1), "max": round(units_per_rank / min(times), 1), "dist_world_size": world}


def dry_run(args, rank: int, world: int, dist) -> None:
    """--dry-run (with --backend gloo: no GPU anywhere): the multi-rank plumbing of this file on the CPU — process group, parameter
    broadcast, the flat gradient bucket's all-reduce, barriers, max-over-ranks timing, per-rank rates, rank 0's single JSON line — with
    a sleep where the render would be.  `value` of such a line is not a measurement ("dry_run": true)."""
    import vf_nerf_amd
    from vf_nerf_amd import distributed as vdist
    dev = torch.device("cpu")
    torch.set_num_threads(1)
    torch.manual_seed(rank)
    model = vf_nerf_amd.VectorFieldNerf(vf_nerf_amd.shipped_config(dev, n_samples=8, n_importance=8))
    vdist.broadcast_parameters(model, src=0)
    vdist.seed_rank_streams(model, rank)
    bucket = vdist.GradientBucket(model)
    for i, p in enumerate(model.unique_parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    t_b = time.perf_counter()
    bucket.all_reduce_mean()
    bucket_ms = (time.perf_counter() - t_b) * 1e3
    expect = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1))) for i, p in enumerate(model.unique_parameters()))
    same = model.vector_field_network.layers[0][0].weight.detach().double().sum().reshape(1)
    sums = [torch.zeros_like(same) for _ in range(world)]
    if dist is not None:
        dist.all_gather(sums, same)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002 * (1 + rank))          # the slowest rank sets the step time
    local = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _, rates = rank_rates(dist, local, args.rays * args.steps, dev)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        emit_line(({"metric": "rays/sec (4096-ray chunk, 128 samples/ray) + PSNR vs ref", "value": round(args.rays * args.steps * world / elapsed, 1),
                          "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                          "dtype": "none", "data": "synthetic", "dry_run": True, "backend": args.backend, "per_rank_rays_per_s": rates,
                          "bucket_elements": bucket.numel(), "bucket_allreduce_ok": bool(ok), "bucket_allreduce_ms": round(bucket_ms, 3),
                          "replicas_identical_after_broadcast": bool(all(float(x) == float(sums[0]) for x in sums)) if dist is not None else True,
                          "config": {"workload": f"dry run of the {args.workload} workload's multi-rank plumbing (no device work)",
                                     "parallelism": f"rays x{world}"}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
    ap.add_argument("--no-two-product-leg", action="store_true", help="skip the timed region of the opt-in two-product colour branch")
    ap.add_argument("--no-other-scene-leg", "--no-random-weight-leg", dest="no_other_scene_leg", action="store_true",
                    help="skip the timed region on the OTHER scene (trained weights when --weights random, and the reverse)")
    ap.add_argument("--no-shipped-rows", action="store_true", help="skip the extra rows at the shipped conf's sample counts (100 + 35, 100 + 100)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed profiles/rNN/traffic_f16x3.json instead of two rocprofv3 --pmc child runs of "
                         "this command on this box (+25 s; render workload, one GPU, not under a profiler)")
    ap.add_argument("--weights", choices=("trained", "random"), default="random",
                    help="render workload: random = the synthetic random-weight scene BASELINE.json's north_star names (the default: `value` "
                         "and `roofline` are measured on it); trained = weights the reference's own trainer arrived at "
                         "(tests/golden/trained_far.npz).  The other scene is timed beside it (value_trained_weights / value_random_weight_scene)")
    ap.add_argument("--train-weights", choices=("trained", "random"), default="trained",
                    help="the training sub-object of the default line (BASELINE.json configs[2]): trained = steps at the state a long run of the "
                         "reference's trainer sits in (the scene keeps its surfaces); random = the synthetic scene (loses them within two Adam steps)")
    ap.add_argument("--train-steps", type=int, default=12, help="optimizer steps timed for the training sub-object")
    ap.add_argument("--activations", choices=("fp32", "f16"), default="f16",
                    help="training: storage of the hidden activations for the weight-gradient kernels (f16 = the default, "
                         "11-bit operands in one factor of dW, half the workspace traffic; fp32 = fp32-equivalent gradients)")
    ap.add_argument("--gradients", choices=("fp32", "f16", "bf16"), default=None,
                    help="training: storage of the pre-activation gradients between the dX chain and the weight-gradient kernels "
                         "(default: the model's)")
    ap.add_argument("--train-colour-products", type=int, choices=(2, 3), default=None,
                    help="training: products of the colour branch in the activation-saving forward (default: the model's, 3)")
    ap.add_argument("--train-products", type=int, choices=(1, 3), default=None,
                    help="training: 1 = the opt-in 16-bit-native mode (one f16 / bf16 product per K block in the saving forward / the dX "
                         "chain, BASELINE.json configs[2] as written; outside the 1e-4 contract); default 3 (split operands, fp32-equivalent)")
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
                    help="f16 products per fp32-equivalent product in the colour branch of the f16x3 render (default: the model's, 3: "
                         "fp32-equivalent; 2 = the opt-in: weights of the feature block + rendering net as their f16 roundings, colours within "
                         "2e-5 on near-init weights only)")
    ap.add_argument("--no-reuse", action="store_true",
                    help="evaluate the VF net on the proposal samples twice, as the reference does (one fused VF+rendering "
                         "launch over all S_c+N_f samples), instead of once")
    ap.add_argument("--unify-res", type=int, default=256, help="grid-stages workload: resolution of the unify / comb stages")
    ap.add_argument("--workload", choices=("render", "view", "grid", "grid-stages", "train"), default="render",
                    help="render = the headline line (default); view = BASELINE configs[1] full view in 1024-ray chunks + "
                         "PSNR/depth vs the oracle image; grid = configs[4] dense grid queries; train = configs[2] step")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="process-group backend for --gpus N > 1 (nccl = RCCL; gloo: --dry-run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="run only the multi-rank plumbing (process group, broadcast, gradient-bucket all-reduce, timing protocol, JSON line); "
                         "with --backend gloo no GPU is touched")
    ap.add_argument("--grid-res", type=int, default=256)
    ap.add_argument("--no-parity", action="store_true", help="view workload: skip the oracle parity image (fp32 + float64, ~50 s of CPU)")
    ap.add_argument("--no-float64", action="store_true", help="view workload: skip the float64 yardstick of the parity image (the slower half)")
    ap.add_argument("--as-evaluator", action="store_true",
                    help="view workload: time evaluator.render_view on HOST inputs (per-ray pose / intrinsics) with the download inside the "
                         "timed region — what the reference's evaluation/methods.py:render_images gets through vf_nerf_amd.dropin")
    ap.add_argument("--unchanged-evaluator-loop", action="store_true",
                    help="with --as-evaluator: the reference's own render_images loop body (what dropin.install(patch_evaluator=False) leaves in "
                         "place) instead of evaluator.render_view")
    ap.add_argument("--streams", type=int, default=2,
                    help="view workload: HIP streams the consecutive ray chunks alternate over (1 = strictly one after the other)")
    ap.add_argument("--train", action="store_true",
                    help="time a training step instead (render with autograd + 2 supervision VF forwards + loss + "
                         "backward + clip + Adam, train/vector_field_nerf_train.py:177-260); not the headline metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not started by a launcher: start the ranks ourselves (children; this process never touches the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    claim_stdout()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s)")
    dist = None
    # VFN_BENCH_FORCE_DIST=1: create the RCCL process group (barriers, the max-over-ranks all-reduce, the gradient
    # bucket) even with one rank — a single-GPU smoke test of the multi-GPU code path
    force_dist = os.environ.get("VFN_BENCH_FORCE_DIST") == "1"
    use_gpu = not (args.dry_run and args.backend == "gloo")
    if world > 1 or force_dist:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if use_gpu:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(args.backend, device_id=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
        else:
            dist.init_process_group(args.backend)
    if args.dry_run:
        dry_run(args, rank, world, dist)
        return
    if world == 1 and not force_dist and not args.train and args.workload == "render" and args.precision == "f16x3" and \
            not args.no_live_traffic and not under_profiler():
        global _LIVE_TRAFFIC
        _LIVE_TRAFFIC = live_hbm_traffic(args)          # (children; before this process touches the GPU)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from vf_nerf_amd import lib
    lib.load()  # fail loudly when the HIP extension is missing
    if args.train:
        args.workload = "train"
    if args.workload == "grid-stages":
        if world > 1:
            raise SystemExit("--workload grid-stages is a single-GPU configuration")
        if args.grid_res == 256:
            args.grid_res = 512
        grid_stages_bench(args, dev)
        return
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
    # The headline scene carries TRAINED weights (what a user of the reference renders: evaluation/methods.py:474-547 runs on the
    # state train/vector_field_nerf_train.py:136-292 arrived at); --weights random = the synthetic random-weight scene of rounds 1-3.
    # The training workload takes them too (its targets: the model's own render, train_bench).
    scene_info = None
    want_trained = (args.weights if args.workload == "render" else args.train_weights) == "trained"
    built = build_trained_scene(dev, args.rays, s_c, n_f, seed=rank) if (want_trained and args.workload in ("render", "train")) else None
    if built is not None:
        model, uv, pose, K, scene_info = built
        model._bench_trained_weights = scene_info
    else:
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
        if args.train_products:
            model.training_products = args.train_products
        train_bench(args, model, uv, pose, K, dev, dist, rank, world, sync)
        return

    def timed_region(mdl, steps, warmup, sustain, events=None, probe=None):
        """`warmup` untimed renders, `sustain` seconds of the same work without a gap, then `steps` timed renders between two
        barrier + synchronize pairs -> (elapsed of this rank's own steps, elapsed including the closing barrier, last output).
        The kernel is power-limited: right after idle the chip boosts, and K = 20 steps are 40 ms, so the timed steps follow
        sustained work and `value` is a sustained figure whatever K the caller asks for."""
        o = None
        for _ in range(warmup):
            mdl.render(pose, uv, K, epoch=0)
        torch.cuda.synchronize()
        t_burn = time.perf_counter()
        while time.perf_counter() - t_burn < sustain:
            for _ in range(16):
                mdl.render(pose, uv, K, epoch=0)
            torch.cuda.synchronize()
        sync()
        t_a = time.perf_counter()
        for i in range(steps):
            # HIP events around the two fused launches on every fourth step of the timed region, recorded by the one-call C path
            # itself on its launch stream (vfn_render_params.timing_events): the path that is timed is the path that ships.  (Every
            # step would be 8 more stream markers per step.)  The same steps leave their workgroups' clock stamps in `probe`.
            sampled = events is not None and i % 4 == 0
            mdl._kernel_events = events if sampled else None
            mdl._clock_probe = probe if sampled else None       # -> vfn_render_params.clock_stamps of that call (no per-thread setter)
            o = mdl.render(pose, uv, K, epoch=0)
        torch.cuda.synchronize()
        local = time.perf_counter() - t_a
        sync()
        total = time.perf_counter() - t_a
        mdl._kernel_events = None
        mdl._clock_probe = None
        return local, total, o

    with torch.no_grad():
        events = []
        # per-workgroup clock stamps of the fused launches (csrc/vfn_mlp16.hip, vfn_f16x3_set_clock_probe): [workgroup][cycles, ticks]
        probe = torch.zeros((args.rays * max(s_c, n_f) + 127) // 128, 2, dtype=torch.int64, device=dev) if args.precision == "f16x3" else None
        local_elapsed, elapsed, out = timed_region(model, args.steps, args.warmup, args.sustain_seconds, events, probe)
        clock = None
        from vf_nerf_amd import lib as vlib
        if probe is not None:
            clock = vlib.clock_ghz_from_stamps(probe)

        # beside `value`, never instead: (a) the opt-in two-product colour branch on the same scene, with what the guard's measured
        # self-check says about it on these weights; (b) the same default path on the synthetic RANDOM-weight scene (BASELINE.json's
        # literal wording; the kernels' work does not depend on the weights, the chip's power draw does a little)
        elapsed2 = check2 = None
        if args.precision == "f16x3" and int(model.colour_products) == 3 and not args.no_two_product_leg:
            check2 = two_product_check(model, uv, pose, K)
            model.colour_products, keep_guard = 2, model.f16x3_guard
            model.f16x3_guard = "off"                      # timing of the opt-in itself; the check above says whether it would be kept
            _, elapsed2, _ = timed_region(model, args.steps, max(3, args.warmup), min(1.0, args.sustain_seconds))
            model.colour_products, model.f16x3_guard = 3, keep_guard
        # the OTHER scene, same path, same run-in: trained weights beside the random-weight headline (or the reverse with --weights trained)
        elapsed_rand, events_rand, clock_rand, other_info = None, [], None, None
        if not args.no_other_scene_leg:
            if scene_info is not None:
                rmodel, uv_r, pose_r, K_r = build_scene(dev, args.rays, s_c, n_f, seed=rank)
                other = (rmodel, uv_r, pose_r, K_r)
            else:
                built_o = build_trained_scene(dev, args.rays, s_c, n_f, seed=rank)
                other = built_o[:4] if built_o is not None else None
                other_info = built_o[4] if built_o is not None else None
            if other is not None:
                rmodel, uv_r, pose_r, K_r = other
                rmodel.precision, rmodel.reuse_proposal = args.precision, not args.no_reuse
                rmodel.colour_products = model.colour_products
                keep_inputs = (uv, pose, K)
                uv, pose, K = uv_r, pose_r, K_r
                probe_r = torch.zeros_like(probe) if probe is not None else None
                # (the same sustain as the headline leg: the launch is power-limited, a shorter run-in would flatter this scene)
                _, elapsed_rand, _ = timed_region(rmodel, args.steps, max(3, args.warmup), args.sustain_seconds, events_rand, probe_r)
                uv, pose, K = keep_inputs
                if probe_r is not None:
                    clock_rand = vlib.clock_ghz_from_stamps(probe_r)
                del rmodel, other
        # the SHIPPED sample counts (confs/vf_nerf.conf:40-48: 100 proposal samples, 35 fine samples growing to max_samples = 100) on the
        # headline scene, as extra rows
        shipped_rows = None
        if args.precision == "f16x3" and not args.no_shipped_rows and args.workload == "render":
            shipped_rows = []
            keep_inputs = (uv, pose, K)
            for sc2, nf2 in ((100, 35), (100, 100)):
                built2 = build_trained_scene(dev, args.rays, sc2, nf2, seed=rank) if scene_info is not None else None
                if built2 is not None:
                    m2, uv, pose, K, _ = built2
                else:
                    m2, uv, pose, K = build_scene(dev, args.rays, sc2, nf2, seed=rank)
                m2.precision, m2.reuse_proposal, m2.colour_products = args.precision, not args.no_reuse, model.colour_products
                _, el2, _ = timed_region(m2, args.steps, max(3, args.warmup), min(1.0, args.sustain_seconds))
                shipped_rows.append((sc2, nf2, el2))
                del m2
            uv, pose, K = keep_inputs
    model._kernel_events = None
    # dominant kernel class = the one with the largest share of the timed region (HIP events on the launch stream)
    per_class = {}
    for name, e0, e1 in events:
        per_class.setdefault(name, []).append(e0.elapsed_time(e1))
    dom = max(per_class, key=lambda k: sum(per_class[k]))
    kernel_ms = sum(per_class[dom]) / len(per_class[dom])
    launches_per_step = len(per_class[dom]) / max(1, (args.steps + 3) // 4)

    _, rates = rank_rates(dist, local_elapsed, args.rays * args.steps, dev)
    if dist is not None:
        t = torch.tensor([elapsed, elapsed2 or 0.0, elapsed_rand or 0.0] + [r[2] for r in (shipped_rows or [])], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed2, elapsed_rand = float(t[0].item()), (float(t[1].item()) if elapsed2 is not None else None), \
            (float(t[2].item()) if elapsed_rand is not None else None)
        if shipped_rows:
            shipped_rows = [(a, b, float(t[3 + i].item())) for i, (a, b, _) in enumerate(shipped_rows)]

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
                "event_sampling": "HIP events recorded by vfn_render_fwd around its two fused launches on every 4th timed step (one-call path)",
                "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": hbm_traffic(f16, dom, cp)[0], "traffic_source": hbm_traffic(f16, dom, cp)[1],
                # what one launch has to move: points in, vector columns out, plus the 1 KiB feature block per point that the
                # split launches hand over (written by vf_feat16, read by render16); weights stream from L2
                "algorithmic_bytes": int(points * {"vf_feat16": 12 + 12 + 1024, "render16": 1024 + 12 + 12 + 4 + 24}.get(dom, 12 + 4 + 24)),
                "flops_per_launch": flops_launch, "avg_launch_ms": round(kernel_ms, 4),
                # the shader clock the fused launches ACTUALLY ran at (in-kernel s_memtime / s_memrealtime stamps of every workgroup of
                # the event-sampled steps' last fused launch; nominal 2.4 GHz): the kernel is power-limited, so box-to-box and
                # run-to-run differences of `value` show up here — a lower figure at an unchanged clock would be a regression
                "effective_clock_ghz": clock["median"] if clock else None, "effective_clock": clock,
                "peak_definition": ((f"dense f16 MFMA 2500 TFLOP/s / {products:.4g} f16 products per fp32-equivalent product (3 in the vector-field "
                                     f"trunk, vector head and encoding columns, 2 in the colour branch = {COLOUR_MACS} of {VF_MACS + RN_MACS} MACs per sample)"
                                     if cp == 2 else "dense f16 MFMA 2500 TFLOP/s / 3 products per fp32-equivalent product") if f16 else
                                    "fp32 MFMA 157.3 TFLOP/s"),
                "executed_f16_tflops": round(achieved * products, 1) if f16 else None,
                # BASELINE.md §3 states the path's roofline against the fp32 matrix peak:
                "frac_of_fp32_mfma_peak": round(achieved / PEAK_F32_MFMA, 4)}
        if f16:   # context, not the graded fraction: the power-limited MFMA rate measured on this chip
            roof["frac_of_measured_sustained_f16_mfma"] = round(achieved / (SUSTAINED_F16_MFMA / products), 4)
        # the SAME accounting for the other scene (trained weights beside the random-weight headline): its own event-timed launch
        # duration, fraction and in-kernel clock
        roof_rand = None
        if events_rand:
            pc = {}
            for name, e0, e1 in events_rand:
                pc.setdefault(name, []).append(e0.elapsed_time(e1))
            if dom in pc:
                k_ms = sum(pc[dom]) / len(pc[dom])
                ach = flops_launch / (k_ms * 1e-3) / 1e12
                roof_rand = {"bound": "mfma", "kernel": roof["kernel"], "avg_launch_ms": round(k_ms, 4), "achieved": round(ach, 2), "peak": round(peak, 1),
                             "unit": "TFLOP/s", "frac": round(ach / peak, 4), "flops_per_launch": flops_launch,
                             "executed_f16_tflops": round(ach * products, 1) if f16 else None,
                             "effective_clock_ghz": clock_rand["median"] if clock_rand else None, "effective_clock": clock_rand,
                             "sustained": f"timed steps follow {args.sustain_seconds:g} s of the same work without a gap (as the headline leg)"}
        line = {
            "metric": "rays/sec (4096-ray chunk, 128 samples/ray) + PSNR vs ref",
            "value": round(rays_per_s, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f16x3+f32acc (colour branch: f16 weights x split activations, 2 products)" if cp == 2 else "f16x3+f32acc") if f16 else "f32",
            "data": "synthetic",
            "sustained": f"timed steps follow {args.sustain_seconds:g} s of the same work without a gap",
            # three products everywhere IS the default now: `value` is the fp32-equivalent figure (kept under its round-3 name too)
            "value_fp32_equivalent": round(rays_per_s, 1) if (f16 and cp == 3) else None,
            "weights": scene_info if scene_info is not None else {"fixture": None, "trained_by": "nobody: synthetic random weights (seed + default init x gain 2 + recentred head)"},
            # the other scene: value_trained_weights / roofline_trained_weights beside the random-weight headline (BASELINE.json's scene),
            # value_random_weight_scene / roofline_random_weight_scene beside a --weights trained one
            ("value_random_weight_scene" if scene_info is not None else "value_trained_weights"):
                round(args.rays * args.steps * world / elapsed_rand, 1) if elapsed_rand else None,
            ("roofline_random_weight_scene" if scene_info is not None else "roofline_trained_weights"): roof_rand,
            "other_scene_weights": other_info,
            # SURVEY.md section 8d "extra row": the shipped conf's sample counts (confs/vf_nerf.conf:40-48), same scene, same default path;
            # their per-ray parity against the reference's outputs at those sizes: parity_shipped_sizes below
            "value_shipped_conf": ([{"samples": f"{a} + {b}", "rays_per_s": round(args.rays * args.steps * world / el, 1),
                                     "ms_per_chunk": round(el / args.steps * 1e3, 4), "samples_per_s": round(args.rays * (a + b) * args.steps * world / el, 1)}
                                    for a, b, el in shipped_rows] if shipped_rows else None),
            "value_two_product_opt_in": round(args.rays * args.steps * world / elapsed2, 1) if elapsed2 else None,
            "two_product_check": check2,
            "value_definition": ("value / roofline: the default path (three f16 products per fp32-equivalent product everywhere, colours 1e-7 from the "
                                 "exact-fp32 kernels) on " + ("TRAINED weights (--weights trained)" if scene_info is not None else
                                                              "the synthetic RANDOM-weight scene BASELINE.json's north_star names") +
                                 "; " + ("value_random_weight_scene (+ roofline_random_weight_scene): the same path on the synthetic random-weight scene"
                                         if scene_info is not None else
                                         "value_trained_weights (+ roofline_trained_weights): the same path, same run-in, on weights the reference's own "
                                         "trainer arrived at (what a user of the reference renders; its activations let the power-limited launch hold a "
                                         "higher clock: effective_clock_ghz of either roofline)") +
                                 "; value_two_product_opt_in: model.colour_products = 2 (colour-branch weights as f16 roundings) timed on the "
                                 "headline scene with the guard off — two_product_check says what it does to the colours there and whether the "
                                 "guard's measured self-check would keep it (on trained weights it does not)"),
            "per_rank_rays_per_s": rates,
            "config": {"workload": f"VectorFieldNerf.render forward, {args.rays}-ray chunk x {s_t} samples/ray "
                                   f"(S_c={s_c} + N_f={n_f}), shipped 9x256 VF + 5x256 rendering MLPs, eval-mode BN, "
                                   f"stratified sampling on device Philox, Replica-like 1200x680 pinhole" +
                                   (f" (f = {TRAINED_SCENE_FOCAL:g}: the training views' field of view)" if scene_info is not None else ""),
                       "rays_per_chunk_per_gpu": args.rays, "samples_per_ray": s_t, "parallelism": f"rays x{world}",
                       "vf_evaluations_per_ray": s_t if getattr(model, "reuse_proposal", False) and f16 else s_c + s_t,
                       "colour_products": cp if f16 else None,
                       "rays_with_nonzero_depth": round(hits, 3)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], line["parity_vs_oracle"] = cpu_baseline(model, uv, pose, K, s_c, n_f)
            line["parity_trained_weights"] = trained_weights_parity(dev, args.precision)
            # the shipped sample counts (100 + 35) against the reference's own outputs at those sizes: reference-trained weights and random ones
            line["parity_shipped_sizes"] = [trained_weights_parity(dev, args.precision, fixture=name) for name in ("trained_256_shipped.npz", "shipped_sizes.npz")]
    # BASELINE.json configs[2] beside the headline line (outside its timed region): a few optimizer steps on the same batch
    # size, every rank, gradients all-reduced when there is more than one
    train_rec = None
    if not args.no_train:
        targs = argparse.Namespace(**vars(args))
        targs.steps, targs.warmup = args.train_steps, 5      # (the first steps size the caching allocator's 15 GB of workspace)
        del out
        torch.cuda.empty_cache()
        tbuilt = build_trained_scene(dev, args.rays, s_c, n_f, seed=rank) if args.train_weights == "trained" else None
        if tbuilt is not None:
            tmodel, tuv, tpose, tK, tinfo = tbuilt
            tmodel._bench_trained_weights = tinfo
        else:
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
                                                       "step_issued_as", "weights", "sparse_colour_branch", "algorithmic_tflop_per_step", "executed_tflop_per_step",
                                                       "executed_tflops", "frac_of_f16_mfma_div3_executed",
                                                       "workspace_gb_per_step", "workspace_gb_per_step_dense_step", "workspace_tb_per_s", "frac_of_hbm_peak", "final_loss", "drop_in_sequence_ms", "drop_in_sequence")}
            line["train"]["workload"] = train_rec["config"]["workload"]
        emit_line((line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
