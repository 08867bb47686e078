#!/usr/bin/env python3
"""HOST time of a training step (CPU time spent issuing work: no device synchronisation inside the timed loop until its end) next to the
wall time of the same steps, for three ways of issuing it: trainer.TrainStep as ONE C call (vfn_train_step, the default), the reference
trainer's OWN call sequence (train/vector_field_nerf_train.py:177-275 restated call for call in tools/reference_sequence.py: render, the
samplers, the two network calls, VFLoss, zero_grad, backward, clip_grad_norm_, optimizer.step — what a reference user drives through
vf_nerf_amd.dropin; "drop_in_sequence": with the loop's per-step loss.item() / running sums, which the HIP path serves as deferred scalars
(vf_nerf_amd/deferred.py: no synchronisation); "_no_item": without those reads; "_sync_item": with plain floats, a synchronisation per step as
in rounds 1-4), and the launch-by-launch autograd path of round 4
("python": model.step_sessions = False, model.one_call_train_step = False).  The reference's batch size (1 024 rays) is the launch-bound
regime.

    python tools/host_profile.py [rays] [steps] [cores]

``cores``: restrict THIS process to that many host cores first (os.sched_setaffinity, before any GPU call) — what one of eight ranks on
a 32-core host gets."""
import json
import os
import sys
import time

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cores = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if cores > 0:
    os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:cores]))      # before torch / HIP are loaded

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import bench  # noqa: E402
from vf_nerf_amd import supervision, trainer  # noqa: E402

if cores > 0:
    torch.set_num_threads(cores)
dev = torch.device("cuda", 0)
out = {"rays": rays, "samples": 128, "steps": steps, "host_cores": len(os.sched_getaffinity(0)), "train_step_streams": int(os.environ.get("VFN_TRAIN_STREAMS", "2"))}
import reference_sequence  # noqa: E402  (tools/)


def measure(step_fn, steps):
    for _ in range(8):
        step_fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    # host time without the device queue pushing back: a synchronisation before every step (the queue is empty when the step is issued)
    iso = 0.0
    for _ in range(min(steps, 30)):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step_fn()
        iso += time.perf_counter() - t1
    torch.cuda.synchronize()
    return {"wall_ms_per_step": round(wall / steps * 1e3, 4), "host_issue_ms_per_step_back_to_back": round(host / steps * 1e3, 4),
            "host_issue_ms_per_step_empty_queue": round(iso / min(steps, 30) * 1e3, 4)}


for path in ("one_call", "drop_in_sequence", "drop_in_sequence_no_item", "drop_in_sequence_sync_item", "python"):
    built = bench.build_trained_scene(dev, rays, 64, 64, seed=0)             # trained weights, targets = the model's own render (bench.training_targets)
    if built is not None:
        model, uv, pose, K, info = built
        model._bench_trained_weights = info
    else:
        model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, seed=0)
    model.one_call_train_step = path == "one_call"
    model.step_sessions = path != "python"
    model.sparse_colour_training = os.environ.get("VFN_SPARSE_COLOURS", "1") != "0"
    model.train_step_streams = int(os.environ.get("VFN_TRAIN_STREAMS", "2"))        # A/B of the side stream inside vfn_train_step
    rgb_gt, depth_gt, centroid, radius = bench.training_targets(model, uv, pose, K, dev, 64, 64)
    supervision.manual_seed(7)
    if path.startswith("drop_in_sequence"):
        from vf_nerf_amd import stepengine
        from types import SimpleNamespace
        from vf_nerf_amd import loss as vloss
        crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
        vloss.DEFERRED_SCALARS = not path.endswith("sync_item")
        loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(centroid, 1.0), radius,
                                                sync_each_step=not path.endswith("no_item"))
        # what the reference's DataLoader hands over: host tensors with a leading batch dimension of one (train.py:172-177) — here already on
        # the device, as a pinned-memory loader with a prefetch stream would leave them (the upload is not what is being measured)
        data = {"uv": uv.unsqueeze(0), "intrinsics": K.unsqueeze(0), "pose": pose.unsqueeze(0), "rgb": rgb_gt.unsqueeze(0), "depth": depth_gt.unsqueeze(0)}
        rec = measure(lambda: loop(data, 0), steps)
        # where the host time of a step goes, stage by stage (empty queue: a synchronisation before every step)
        loop.stage_seconds = {}
        for _ in range(30):
            torch.cuda.synchronize()
            loop(data, 0)
        torch.cuda.synchronize()
        rec["host_ms_by_stage_empty_queue"] = {k: round(v / 30 * 1e3, 4) for k, v in loop.stage_seconds.items()}
        loop.stage_seconds = None
        eng = stepengine.StepEngine.of(model)
        sel = model._last_colour_counts.tolist() if getattr(model, "_last_colour_counts", None) is not None else None
        rec.update({"took_the_step_session": eng.why_not is None and eng.session is not None, "why_not": eng.why_not,
                    "sparse_colour_branch": bool(model.sparse_colour_training) and eng.why_not is None,
                    "samples_with_nonzero_weight": round(sel[0] / sel[1], 4) if sel else None,
                    "loss_item_every_step": loop.sync_each_step, "deferred_scalars": vloss.DEFERRED_SCALARS})
        vloss.DEFERRED_SCALARS = True
        out[path] = rec
        continue
    step = trainer.TrainStep(model, centroid, border_radius=radius, far=1.0)
    rec = measure(lambda: step(pose, uv, K, rgb_gt, depth_gt, epoch=0), steps)
    sel = step.last_colour_counts.tolist() if step.last_colour_counts is not None else None
    rec.update({"took_one_call_path": step.one_call.why_not is None, "why_not": step.one_call.why_not,
                "sparse_colour_branch": bool(model.sparse_colour_training) and step.one_call.why_not is None,
                "samples_with_nonzero_weight": round(sel[0] / sel[1], 4) if sel else None})
    out[path] = rec
print(json.dumps(out))
