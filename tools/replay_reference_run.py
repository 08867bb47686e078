#!/usr/bin/env python3
"""Replay the REFERENCE trainer's recorded run on the HIP path and compare the curves.

``tests/golden/trained_256_run.npz`` (tests/golden/make_run_golden.py) holds the task of ``trained_256.npz``'s recorded run — the 100 fixed
64-ray batches with the targets the reference's ``render()`` produced from the teacher, the student's initial vector head — and the curves of
the reference's own ``VectorFieldNerfRunner.train_epoch`` (train/vector_field_nerf_train.py:161-292) on it: row 0 the recorded run (reproduced
bit for bit when the fixture was made), rows 1.. the same task under other torch / numpy random streams.  Here the same 12 epochs x 100 steps
run through ``tools/reference_sequence.ReferenceLoop`` — the reference trainer's loop body call for call on the names ``vf_nerf_amd.dropin``
installs, i.e. through the step session — from the same initial weights, over several device random streams (Philox: stratified jitter, the
always-drawn z_add, the supervision points), and with the exact-fp32 kernels (fp32 storages, launch-by-launch autograd) as the control.

The runs are chaotic (a ReLU unit or an argmax on the other side parts two trajectories for good), the reference's own five runs end between
11.4 and 13.5 dB: the comparison is statistical — windowed means of the total loss and of the four active terms, the median clip norm, the
final PSNR against the teacher's targets — with the reference's own run-to-run spread as the yardstick.

    python tools/replay_reference_run.py [--streams 4] [--out profiles/r06/replay_reference_run]      # writes <out>.json and <out>.md
"""
from __future__ import annotations

import ast
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

TERMS = ("rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss")       # the active ones (norm_smaller_than_one starts at epoch 11 000)
WINDOW = 100
TASKS = {"256": "trained_256_run.npz",      # 12 epochs x 100 batches of 64 rays (the run recorded in trained_256.npz)
         "far": "trained_far_run.npz"}      # 80 epochs x 100 batches of 256 rays (trained_far.npz): make_run_golden.py --far


def task_available(task: str = "256") -> bool:
    return os.path.exists(os.path.join(REPO, "tests", "golden", TASKS[task]))


def load_task(task: str = "256"):
    raw = np.load(os.path.join(REPO, "tests", "golden", TASKS[task]))
    recipe = ast.literal_eval(str(raw["train_recipe"]))
    return raw, recipe


def build_student(raw, recipe, dev):
    """Seed + default init x gain + the stored vector head (tests/helpers.build_model: the same recipe the render fixtures use), checked
    against the checksum of the reference model the run started from."""
    import helpers
    fx = dict(seed=recipe["seed"], gain=recipe["gain"], n_samples=recipe["n_samples"], n_importance=recipe["n_importance"], perturb=recipe["perturb"],
              near=recipe["near"], far=recipe["far"], fine_range=recipe["fine_range"], th=recipe["th"], n_window=recipe["n_window"])
    data = {"head_weight": torch.from_numpy(raw["init.head_weight"]), "head_bias": torch.from_numpy(raw["init.head_bias"]),
            "weights_checksum": torch.from_numpy(raw["init.checksum"])}
    model = helpers.build_model(fx, data, device=str(dev))
    sc = model.config.scheduler_config
    assert (sc.lr, sc.clip_norm, sc.lr_decay_steps) == (recipe["lr"], recipe["clip_norm"], recipe["lr_decay_steps"])
    return model


def batches_on(raw, dev):
    n_b, n = raw["batch.uv"].shape[:2]
    out = []
    for t in range(n_b):
        pose = torch.from_numpy(raw["batch.pose"][t]).reshape(1, 4, 4).expand(n, 4, 4).contiguous()
        K = torch.from_numpy(raw["batch.intrinsics"][t]).reshape(1, 4, 4).expand(n, 4, 4).contiguous()
        # (the shapes the reference's DataLoader(batch_size=1) hands train_epoch: a leading 1)
        out.append({"uv": torch.from_numpy(raw["batch.uv"][t]).unsqueeze(0).to(dev), "pose": pose.unsqueeze(0).to(dev), "intrinsics": K.unsqueeze(0).to(dev),
                    "rgb": torch.from_numpy(raw["batch.rgb"][t]).unsqueeze(0).to(dev), "depth": torch.from_numpy(raw["batch.depth"][t]).unsqueeze(0).to(dev)})
    return out


def psnr(model, batches, first=16) -> float:
    """make_trained_golden.train.psnr: deterministic render of the first 16 training batches against their targets."""
    keep = (model.ray_sampler.deterministic, model.fine_sampler.deterministic)
    model.ray_sampler.deterministic = model.fine_sampler.deterministic = True
    se, n = 0.0, 0
    with torch.no_grad():
        for b in batches[:first]:
            out = model.render(b["pose"][0], b["uv"][0], b["intrinsics"][0], 0, False)
            se += float(((out.coarse_rgb_values - b["rgb"][0]).double() ** 2).sum())
            n += b["rgb"][0].numel()
    model.ray_sampler.deterministic, model.fine_sampler.deterministic = keep
    return float(-10.0 * np.log10(se / n))


def replay(stream: int, kernels: str = "default", dev=None, epochs=None, task: str = "256"):
    """One run of the recorded task -> dict(loss[steps], terms[steps,4], clip[steps], psnr_before_after, issued_as, seconds).
    ``kernels``: "default" (f16x3, 16-bit storages, step session) or "fp32" (exact-fp32 MFMA kernels, fp32 storages, launch by launch)."""
    import reference_sequence
    from vf_nerf_amd import loss as vloss, stepengine, supervision, trainer
    dev = dev or torch.device("cuda:0")
    raw, recipe = load_task(task)
    model = build_student(raw, recipe, dev)
    batches = batches_on(raw, dev)
    if kernels == "fp32":
        model.precision, model.activation_storage, model.gradient_storage = "fp32", "fp32", "fp32"
    elif kernels == "fp32_storages":      # the default f16x3 kernels through the session, activations and gradients stored as fp32
        model.activation_storage, model.gradient_storage = "fp32", "fp32"
    elif kernels == "default_lbl":          # the default kernels and storages WITHOUT the step session: launch-by-launch autograd (backward.py), dense colours
        model.step_sessions = False
    elif kernels == "fp32_backward":        # the f16x3 FORWARD kernels with the exact-fp32 BACKWARD kernels (a diagnostic knob: launch by launch)
        model.backward_kernels = "fp32"
        model.vector_field_network.backward_kernels = model.rendering_network.backward_kernels = "fp32"
    elif kernels == "default_dense":        # the step session with the DENSE colour branch
        model.sparse_colour_training = False
    elif kernels.startswith("storages:"):      # "storages:<activations>,<gradients>", e.g. storages:f16,fp32
        model.activation_storage, model.gradient_storage = kernels.split(":", 1)[1].split(",")
    model.rng_seed, model._rng_offset = 9000 + 101 * stream, 0
    supervision.manual_seed(7000 + 53 * stream)
    crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
    loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(recipe["centroid"], recipe["far"], recipe["near"]),
                                            recipe["border_radius"], clip_norm=recipe["clip_norm"], sync_each_step=False)
    psnr0 = psnr(model, batches)
    losses, terms, clips = [], [], []
    n_epochs = recipe["epochs"] if epochs is None else epochs
    took_session = 0
    t0 = time.perf_counter()
    for epoch in range(n_epochs):
        for b in batches:
            loss, td = loop(b, epoch)
            losses.append(loss.detach())
            terms.append([td[k] for k in TERMS])
            clips.append(loop.last_total_norm)
            eng = stepengine.StepEngine.of(model)
            took_session += int(eng.why_not is None and eng.session is not None)
    torch.cuda.synchronize(dev)
    seconds = time.perf_counter() - t0
    steps = len(losses)
    step8 = float(model.optimizer.state[model.vector_field_network.layers[8].weight]["step"])
    assert step8 == 2 * steps, "SURVEY Q4: two Adam updates per step for the aliased vector-field parameters"
    return {"loss": np.array([float(x) for x in losses]), "terms": np.array([[float(v) for v in row] for row in terms]),
            "clip": np.array([float(c) for c in clips]), "psnr_before_after": [psnr0, psnr(model, batches)],
            "issued_as": "step session" if took_session == steps else f"launch by launch ({stepengine.StepEngine.of(model).why_not})" if took_session == 0 else f"mixed ({took_session} of {steps} in a session)",
            "guard_switched_to_fp32": model.f16x3_disabled, "seconds": round(seconds, 1), "steps": steps}


def windows(x: np.ndarray, window: int = WINDOW) -> np.ndarray:
    """[runs, steps(, k)] -> means over consecutive ``window``-step windows along the steps axis."""
    runs, steps = x.shape[:2]
    return x[:, :steps // window * window].reshape(runs, steps // window, window, *x.shape[2:]).mean(axis=2)


def compare(ref: dict, runs: list, window: int = WINDOW) -> dict:
    """Reference curves (``runs.*`` of the fixture) against a family of replays.  Per quantity and 100-step window: the reference runs'
    envelope [min, max] and spread, the replays' mean, and where that mean sits — ``outside`` = distance outside the envelope in units of
    the envelope's width (0 inside)."""
    out = {"quantities": {}, "window": window}
    names = ("loss",) + TERMS
    ref_w = {"loss": windows(ref["loss"], window)}
    hip_w = {"loss": windows(np.stack([r["loss"] for r in runs]), window)}
    rt, ht = windows(ref["terms"], window), windows(np.stack([r["terms"] for r in runs]), window)
    for j, k in enumerate(TERMS):
        ref_w[k], hip_w[k] = rt[:, :, j], ht[:, :, j]
    worst = 0.0
    for k in names:
        r, h = ref_w[k], hip_w[k]
        lo, hi, mean_r, sd_r = r.min(0), r.max(0), r.mean(0), r.std(0, ddof=1)
        mean_h, sd_h = h.mean(0), (h.std(0, ddof=1) if h.shape[0] > 1 else np.zeros(h.shape[1]))
        width = np.maximum(hi - lo, 0.02 * np.abs(mean_r) + 1e-4)       # (a window where the five reference runs happen to coincide still gets 2 %)
        outside = np.maximum(0.0, np.maximum(lo - mean_h, mean_h - hi)) / width
        # difference of the family means in units of its standard error (Welch)
        se = np.sqrt(sd_r ** 2 / r.shape[0] + sd_h ** 2 / max(1, h.shape[0])) + 1e-12
        out["quantities"][k] = {"reference_mean": mean_r.round(5).tolist(), "reference_min": lo.round(5).tolist(), "reference_max": hi.round(5).tolist(),
                                "replay_mean": mean_h.round(5).tolist(), "replay_min": h.min(0).round(5).tolist(), "replay_max": h.max(0).round(5).tolist(),
                                "replay_mean_outside_reference_envelope_in_widths": outside.round(3).tolist(),
                                "difference_of_means_in_standard_errors": ((mean_h - mean_r) / se).round(2).tolist(),
                                "ratio_of_means": (mean_h / mean_r).round(4).tolist()}
        worst = max(worst, float(outside.max()))
    out["worst_window_outside_in_widths"] = round(worst, 3)
    clip_r = np.median(ref["clip"], axis=1)
    clip_h = np.array([np.median(r["clip"]) for r in runs])
    out["clip_norm_median"] = {"reference_runs": clip_r.round(4).tolist(), "replays": clip_h.round(4).tolist(),
                               "ratio_of_family_means": round(float(clip_h.mean() / clip_r.mean()), 4)}
    ps_r = ref["psnr_before_after"]
    ps_h = np.array([r["psnr_before_after"] for r in runs])
    out["psnr_vs_teacher_db"] = {"reference_before": round(float(ps_r[0, 0]), 3), "replay_before": ps_h[:, 0].round(3).tolist(),
                                 "reference_after": ps_r[:, 1].round(3).tolist(), "replay_after": ps_h[:, 1].round(3).tolist(),
                                 "reference_after_mean": round(float(ps_r[:, 1].mean()), 3), "replay_after_mean": round(float(ps_h[:, 1].mean()), 3),
                                 "reference_after_min_max": [round(float(ps_r[:, 1].min()), 3), round(float(ps_r[:, 1].max()), 3)]}
    return out


def reference_curves(raw) -> dict:
    return {"loss": raw["runs.loss"], "terms": raw["runs.terms"][:, :, :4], "clip": raw["runs.clip"], "psnr_before_after": raw["runs.psnr_before_after"]}


def markdown(report: dict) -> str:
    W = report.get("window", WINDOW)
    lines = [f"# The reference trainer's recorded {report.get('steps', 1200)}-step run, replayed on the HIP path\n",
             report["task"] + "\n",
             f"Per {W}-step window: mean over the reference's runs [min .. max of its runs] | mean over the replays [min .. max].\n"]
    for fam in ("default", "fp32_storages", "fp32"):
        if fam not in report:
            continue
        c = report[fam]["comparison"]
        lines.append(f"## {report[fam]['what']} — {report[fam]['runs']} streams, {report[fam]['issued_as']}, {report[fam]['ms_per_step']} ms per step\n")
        for k, q in c["quantities"].items():
            lines.append(f"**{k}**\n")
            lines.append("| window | reference | replay | outside envelope (widths) | ratio of means |")
            lines.append("|---|---|---|---|---|")
            for w in range(len(q["reference_mean"])):
                lines.append(f"| {w * W}-{(w + 1) * W - 1} | {q['reference_mean'][w]:.4f} [{q['reference_min'][w]:.4f} .. {q['reference_max'][w]:.4f}] | "
                             f"{q['replay_mean'][w]:.4f} [{q['replay_min'][w]:.4f} .. {q['replay_max'][w]:.4f}] | "
                             f"{q['replay_mean_outside_reference_envelope_in_widths'][w]:.2f} | {q['ratio_of_means'][w]:.3f} |")
            lines.append("")
        p, cl = c["psnr_vs_teacher_db"], c["clip_norm_median"]
        lines.append(f"PSNR against the teacher's targets: before {p['replay_before'][0]:.3f} dB (reference {p['reference_before']:.3f}); after: reference runs "
                     f"{p['reference_after']} (mean {p['reference_after_mean']}), replays {p['replay_after']} (mean {p['replay_after_mean']}).")
        lines.append(f"Median clip norm: reference runs {cl['reference_runs']}, replays {cl['replays']} (ratio of family means {cl['ratio_of_family_means']}).")
        lines.append(f"Worst window of any quantity: replay mean {c['worst_window_outside_in_widths']} envelope widths outside the reference runs' envelope.\n")
    return "\n".join(lines)


def main() -> None:
    streams = int(sys.argv[sys.argv.index("--streams") + 1]) if "--streams" in sys.argv else 4
    task = sys.argv[sys.argv.index("--task") + 1] if "--task" in sys.argv else "256"
    control = int(sys.argv[sys.argv.index("--control-streams") + 1]) if "--control-streams" in sys.argv else streams
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(REPO, "gpurun_out", "replay_reference_run" + ("" if task == "256" else "_" + task))
    raw, recipe = load_task(task)
    ref = reference_curves(raw)
    steps = recipe["epochs"] * recipe["steps_per_epoch"]
    window = WINDOW if steps <= 2000 else 500
    report = {"steps": steps, "window": window, "task": f"{recipe['epochs']} epochs x {recipe['steps_per_epoch']} fixed batches of {recipe['n_rays']} rays x ({recipe['n_samples']} + {recipe['n_importance']}) "
                      f"samples, 8 orbit views, targets rendered by the reference from a teacher of another seed; the reference's own train_epoch ran it "
                      f"{ref['loss'].shape[0]} times (row 0 = the run recorded in tests/golden/{TASKS[task].replace('_run', '')}, reproduced to {float(raw['runs.reproduces_recorded'][0]):.1e} when the "
                      f"fixture was made; the others under other torch / numpy seeds)."}
    families = [("default", "default kernels (f16x3, 16-bit storages)"), ("fp32", "exact-fp32 kernels, fp32 storages (control)")]
    if "--fp32-storages" in sys.argv:
        families.insert(1, ("fp32_storages", "default f16x3 kernels, activations and gradients stored as fp32"))
    for fam, what in families:
        runs = [replay(s, fam, task=task) for s in range(control if fam == "fp32" else streams)]
        if not runs:
            continue
        report[fam] = {"what": what, "runs": len(runs), "issued_as": runs[0]["issued_as"], "ms_per_step": round(1e3 * sum(r["seconds"] for r in runs) / sum(r["steps"] for r in runs), 3),
                       "guard_switched_to_fp32": [r["guard_switched_to_fp32"] for r in runs], "comparison": compare(ref, runs, window),
                       "loss_window_means_per_run": windows(np.stack([r["loss"] for r in runs]), window).round(5).tolist()}
        print(f"{fam}: {len(runs)} runs, {report[fam]['ms_per_step']} ms/step, final PSNR {report[fam]['comparison']['psnr_vs_teacher_db']['replay_after']}, worst window "
              f"{report[fam]['comparison']['worst_window_outside_in_widths']} widths outside", flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out + ".json", "w") as fh:
        json.dump(report, fh, indent=1)
    with open(out + ".md", "w") as fh:
        fh.write(markdown(report))
    print(f"wrote {out}.json / .md")


if __name__ == "__main__":
    main()
