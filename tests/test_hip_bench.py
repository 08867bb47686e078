"""bench.py's contract, checked on the device: ONE JSON line on stdout, the required keys, and the arithmetic between them."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, env=env)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one line, got {len(lines)}: {proc.stdout[:500]}"
    return json.loads(lines[0])


def test_default_line_has_the_contract_keys_and_consistent_arithmetic():
    d = _run("--gpus", "1", "--steps", "8", "--warmup", "2", "--sustain-seconds", "0.3", "--train-steps", "3")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 8 and d["warmup"] == 2 and d["unit"] == "rays/s" and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert "4096" in d["metric"] and d["config"]["rays_per_chunk_per_gpu"] == 4096 and d["config"]["samples_per_ray"] == 128
    # value = rays of all timed steps / time
    assert abs(d["value"] - 4096 * 1000.0 / d["ms_per_step"]) < 2e-3 * d["value"]
    # the default path IS the fp32-equivalent one (three products everywhere) and `value` / `roofline` are measured on the synthetic
    # RANDOM-weight scene BASELINE.json's north_star names; the trained scene (with its own roofline object) and the opt-in two-product
    # colour branch are reported beside it
    assert d["value_fp32_equivalent"] == d["value"] and d["config"]["colour_products"] == 3
    assert d["weights"]["fixture"] is None and "value_random_weight_scene" not in d
    assert d["other_scene_weights"]["fixture"] in ("tests/golden/trained_far.npz", "tests/golden/trained_256.npz")
    assert d["value_trained_weights"] is not None and 0.85 < d["value_trained_weights"] / d["value"] < 1.2
    rt = d["roofline_trained_weights"]
    assert abs(rt["frac"] - rt["achieved"] / rt["peak"]) < 2e-3 and rt["flops_per_launch"] == d["roofline"]["flops_per_launch"]
    assert d["value_two_product_opt_in"] is not None and d["value_two_product_opt_in"] > d["value"] * 0.98
    c2 = d["two_product_check"]
    assert c2["geometry_bit_identical"] and c2["max_abs_colour_difference"] > 0
    assert c2["strict_guard_keeps_two_products"] == (c2["colour_products_reason"] is None)
    assert d["per_rank_rays_per_s"]["dist_world_size"] == 1
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert abs(r["achieved"] - r["flops_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e12) < 1e-2 * r["achieved"]
    assert 2 * r["avg_launch_ms"] <= d["ms_per_step"] * 1.02, "two launches of the dominant kernel fit in a step"
    assert 0.3 < r["frac"] < 1.0 and r["traffic"] is None or r["traffic"] > r["algorithmic_bytes"]
    # the HBM counters come from two rocprofv3 --pmc child runs on THIS box, or — when the profiler cannot run here — from the committed
    # file, with the reason beside it
    src = r["traffic_source"]
    print("roofline.traffic:", r["traffic"], json.dumps(src)[:400])
    assert src is not None and (src.get("live") is True or ("file" in src and src["live_pass"]["live"] is False and src["live_pass"]["why"]))
    if src.get("live"):
        assert src["launches_averaged"]["FETCH_SIZE"] >= 10 and 2e7 < r["traffic"] < 2e8
    assert 1.0 < r["effective_clock_ghz"] < 2.6 and r["effective_clock"]["workgroups"] >= 1024
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "rays/s" and c["cores"] >= 1 and 0 < c["value"] < d["value"] / 100
    p = d["parity_vs_oracle"]
    assert p["rays_with_different_z"] == 0 and p["frac_rays_within_1e-4"] == 1.0 and p["colour_products_ran"] == 3
    t = d["parity_trained_weights"]
    assert t["frac_rays_within_1e-4"] == 1.0 and t["rays_with_different_z"] == 0 and t["guard"]["switched_to_fp32"] is None
    tr = d["train"]
    assert tr["unit"] == "rays/s" and abs(tr["value"] - 4096 * 1000.0 / tr["ms_per_step"]) < 2e-3 * tr["value"] and 0.2 < tr["frac_of_f16_mfma_div3_executed"] < 1.0
    assert "frac_of_f16_mfma_div3" not in tr and tr["weights"]["fixture"] is not None      # (a fraction of EXECUTED work only; trained weights)
    # the SAME step as the reference trainer's own call sequence: it took the step session; with the loop's per-step
    # reads as deferred scalars it costs what it costs with those reads left out, and no more than with a synchronising item() (20 timed
    # steps each: generous slack for the run-to-run spread)
    seq = tr["drop_in_sequence"]
    assert seq["took_the_step_session"] and seq["why_not"] is None and tr["drop_in_sequence_ms"] == seq["ms_per_step"]
    assert seq["ms_per_step"] <= 1.15 * seq["ms_per_step_without_loss_item"] and seq["ms_per_step"] <= 1.05 * seq["ms_per_step_with_a_synchronising_item"]
    assert seq["ms_per_step"] < 1.25 * tr["ms_per_step"] and abs(seq["running_loss_mean_read_at_the_end"]) < 10.0
    # bytes of the step that ran (sparse colour branch) next to the dense step's accounting
    assert tr["workspace_gb_per_step"] < 0.8 * tr["workspace_gb_per_step_dense_step"]


def test_rccl_process_group_with_one_rank_keeps_stdout_to_the_json_line():
    """VFN_BENCH_FORCE_DIST=1: the multi-GPU code path (RCCL process group, barriers, max-over-ranks all-reduce, the gradient
    bucket) on one GPU.  RCCL prints a version banner through C stdio at exit: it must not reach stdout."""
    env_keep = os.environ.get("VFN_BENCH_FORCE_DIST")
    os.environ["VFN_BENCH_FORCE_DIST"] = "1"
    try:
        d = _run("--workload", "train", "--rays", "1024", "--steps", "5", "--warmup", "2")
    finally:
        if env_keep is None:
            del os.environ["VFN_BENCH_FORCE_DIST"]
        else:
            os.environ["VFN_BENCH_FORCE_DIST"] = env_keep
    assert d["n_gpus"] == 1 and d["bucket_elements"] == 805780 and d["per_rank_rays_per_s"]["dist_world_size"] == 1
    assert "all-reduced over one flat bucket" in d["config"]["workload"]


def test_single_product_line_says_what_it_is():
    """--train-products 1: the opt-in 16-bit-native mode is named in `dtype`, carries `training_products`, and is not priced against the
    three-product ceiling; the default line is."""
    d1 = _run("--workload", "train", "--rays", "1024", "--steps", "5", "--warmup", "2", "--train-products", "1", "--no-parity")
    d3 = _run("--workload", "train", "--rays", "1024", "--steps", "5", "--warmup", "2", "--no-parity")
    assert d1["training_products"] == 1 and d1["dtype"].startswith("16-bit-native") and 0 < d1["frac_of_f16_mfma_executed_single_product"] < 1
    assert d3["training_products"] == 3 and d3["dtype"].startswith("f16x3") and 0.1 < d3["frac_of_f16_mfma_div3_executed"] < 1.0
    assert d3["frac_of_f16_mfma_executed_single_product"] is None
    assert d1["metric"] == d3["metric"] and d1["final_loss"] == d1["final_loss"]


def test_view_workload_per_ray_accounting_at_view_scale():
    """BASELINE.json configs[1] (`bench.py --workload view`): the 12 750 rays of the 150 x 85 parity image of the full view, HIP default
    path against the CPU oracle (fp32), accounted per ray.  Sample depths are bit-identical on EVERY ray; at least 99.8 % of the rays are
    inside the 1e-4 contract in rgb and depth (bound 99.85 %, observed 99.89 %: on a handful of rays per ten thousand the reference's
    function amplifies a 5e-6 difference in the normals past 1e-4 — its own fp32 arithmetic is 1e-3 from its float64 value there, DESIGN.md
    section 4; the float64 yardstick itself is the tool's, not this test's: --no-float64)."""
    d = _run("--workload", "view", "--steps", "1", "--warmup", "3", "--no-float64", timeout=900)
    p = d["parity_vs_oracle"]
    a = p["hip_default_vs_oracle_f32"]
    print(f"view parity image {p['image']}: rays sampled identically {a['rays_sampled_bit_identically']}, inside 1e-4 {a['frac_all_rays_within_1e-4']}, "
          f"worst rgb {a['max_abs_rgb_err']:.2e}; exact-fp32 kernels: {p['hip_exact_fp32_vs_oracle_f32']['frac_all_rays_within_1e-4']}; PSNR {p['psnr_rgb_db']} dB")
    assert p["argmax_indices_equal"] and a["rays_with_different_z"] == 0 and a["rays_sampled_bit_identically"] == 1.0
    assert a["frac_all_rays_within_1e-4"] >= 0.9985 and p["hip_exact_fp32_vs_oracle_f32"]["frac_all_rays_within_1e-4"] >= 0.9985
    assert p["psnr_rgb_db"] > 90.0 and d["value"] > 1.0e6 and d["config"]["colour_products"] == 3
    # WHY those few rays are outside.  NOT a density decision that fell on the other side: no ray has a sample whose
    # sigma is zero on one side only.  It is the normals' own error — 4-8e-6, an order and more inside the 1e-4 contract — amplified 20-300x by
    # the reference's function downstream of them (the oracle's own fp32-against-float64 comparison shows the same rays with the same
    # amplifications; bench.ray_error_sources).  Pinned per ray: the ORACLE's density + weights + composite evaluated on the HIP normals
    # reproduce the HIP output (residual < 5e-5; observed 1e-7 with the default kernels: nothing downstream of the normals differs), and the
    # normals on that ray are within 2e-5 of the oracle's.
    for who in ("hip_default_vs_oracle_f32", "hip_exact_fp32_vs_oracle_f32"):
        m = p["out_of_tolerance_rays"][who]
        print(f"{who}: {m['out_of_tolerance']} rays outside 1e-4, {m['explained_by_their_normals']} explained by their normals (largest residual "
              f"{m['largest_residual']}), {m['rays_with_a_flipped_sample']} with a flipped density decision, {m['unexplained']} unexplained; max |dn| over "
              f"all rays {m['max_abs_normal_err_all_rays']:.2e}")
        for q in m["rays"]:
            print(f"    ray {q['ray']:6d}: rgb err {q['rgb_err']:.2e}, depth err {q['depth_err']:.2e}, max |dn| {q['max_abs_normal_err']:.2e} (x{q['amplification']}), "
                  f"shortest weighted normal {q['min_normal_length']:.2e}, flipped {q['flipped_samples']}, residual {q['residual']:.2e}")
        assert m["unexplained"] == 0 and m["rays_with_a_flipped_sample"] == 0 and m["max_abs_normal_err_all_rays"] < 2e-5, \
            [q for q in m["rays"] if q["residual"] >= 5e-5 or q["max_abs_normal_err"] >= 2e-5]
