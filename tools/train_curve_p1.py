"""tools/train_curve.py's experiment for the opt-in 16-bit-native training mode (``model.training_products = 1``): three random
streams of the default path (f16x3 forward, bf16x3 chain) beside three of the single-product kernels (f16 forward, bf16 chain) —
same teacher, initial weights, batches and streams as train_curve.py, so its exact-fp32 family is the common yardstick.

    python tools/train_curve_p1.py [steps] [rays per batch] > profiles/r03/train_curve_p1.json"""
import json, sys, time
import torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import supervision, trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
dev = torch.device("cuda:0")
n_rays, s_c, n_f = (int(sys.argv[2]) if len(sys.argv) > 2 else 1024), 64, 64
centroid = (0.0, 0.0, 0.55)
teacher, _, _, _ = bench.build_scene(dev, 16, s_c, n_f, seed=0, weight_seed=1)
pool = trainer.TeacherTargets(teacher, views=8, width=64, height=64, focal=60.0, seed=5)


def run(products, stream):
    model, _, _, _ = bench.build_scene(dev, 16, s_c, n_f, seed=0, weight_seed=0)
    model.training_products = products
    model.rng_seed, model._rng_offset = 11 + 1000 * stream, 0
    supervision.manual_seed(3 + 1000 * stream)
    psnr0 = pool.psnr(model)
    step = trainer.TrainStep(model, centroid, border_radius=0.15, far=1.0)
    losses = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(steps):
        pose, uv, K, rgb_gt, depth_gt = pool.batch(t, n_rays)
        losses.append(step(pose, uv, K, rgb_gt, depth_gt, epoch=0)[0])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    losses = [float(x) for x in losses]
    return {"ms_per_step": round(ms, 3), "loss_first_10_mean": sum(losses[:10]) / 10, "loss_last_25_mean": sum(losses[-25:]) / 25,
            "psnr_vs_teacher_before_after_db": [round(psnr0, 3), round(pool.psnr(model), 3)],
            "guard_switched_to_fp32": model.f16x3_disabled, "colour_products_reason": model.range_guard.colour_products_reason,
            "two_product_check_after_training": bench.two_product_check(model, *[pool.batch(777_000, 1024)[i] for i in (1, 0, 2)]) if model.uses_f16x3() else None,
            "non_finite_losses": sum(1 for x in losses if x != x),
            "loss_mean_per_100_steps": [round(sum(losses[i:i + 100]) / len(losses[i:i + 100]), 5) for i in range(0, steps, 100)]}


results = {}
for products, tag in ((3, "default (three products)"), (1, "single product")):
    for st in (0, 1, 2):
        results[f"{tag}, stream {st}"] = run(products, st)


def family(prefix):
    rs = [v for k, v in results.items() if k.startswith(prefix)]
    loss = [r["loss_last_25_mean"] for r in rs]
    psnr = [r["psnr_vs_teacher_before_after_db"][1] for r in rs]
    return {"runs": len(rs), "final_loss_mean": round(sum(loss) / len(loss), 4), "final_loss_min_max": [round(min(loss), 4), round(max(loss), 4)],
            "final_psnr_mean_db": round(sum(psnr) / len(psnr), 3), "final_psnr_min_max_db": [round(min(psnr), 3), round(max(psnr), 3)],
            "ms_per_step_mean": round(sum(r["ms_per_step"] for r in rs) / len(rs), 3)}


summary = {"default": family("default"), "single product": family("single product")}
summary["single_minus_default"] = {
    "final_loss_mean_ratio": round(summary["single product"]["final_loss_mean"] / summary["default"]["final_loss_mean"], 4),
    "final_psnr_mean_db": round(summary["single product"]["final_psnr_mean_db"] - summary["default"]["final_psnr_mean_db"], 3),
    "default_own_spread_db": round(summary["default"]["final_psnr_min_max_db"][1] - summary["default"]["final_psnr_min_max_db"][0], 3)}
print(json.dumps({"workload": f"{steps} optimizer steps of trainer.TrainStep on {n_rays}-ray batches x {s_c + n_f} samples, teacher-rendered targets "
                              "(tools/train_curve.py's pool, initial weights, batches and streams)", "summary": summary, "runs": results}, indent=1))
