#!/usr/bin/env python3
"""The reference trainer's recorded 1 200-step run (``trained_256.npz``: ``curve.*``) as a REPLAYABLE task, plus the reference's own
run-to-run spread on it.

``trained_256.npz`` holds the loss curve of ``VectorFieldNerfRunner.train_epoch`` (train/vector_field_nerf_train.py:161-292) over 12
epochs x 100 fixed 64-ray batches, and the weights it arrived at — but not what a second implementation needs to run the SAME task where
the reference cannot be imported (the GPU box): the batches (pixels, pose, intrinsics, and the rgb / depth targets the reference's
``render()`` produced from the teacher model) and the initial weights.  This script regenerates them with ``make_trained_golden``'s own
functions, proves they are the recorded run's by re-running the reference trainer on them under the recorded seeds (the curve must
reproduce ``trained_256.npz``'s), and then runs the reference trainer ``--streams`` - 1 more times on the same task under OTHER random
streams (torch / numpy seeds: stratified jitter, the always-drawn ``z_add``, the supervision points): the spread of the reference's own
curves is the yardstick a statistical comparison of another implementation's curves needs (the runs are chaotic; tools/train_curve.py).

Runs only in the build container (needs /root/reference, read-only); nothing of the reference's source travels.

    python tests/golden/make_run_golden.py [--streams 4]          # ~2 minutes per stream on 8 cores
    python tests/golden/make_run_golden.py --far [--streams 3]    # the 8 000-step run of trained_far.npz (256-ray batches): ~half an hour per stream
    python tests/golden/make_run_golden.py --far --append 4       # four MORE reference runs appended to the existing fixture (its task is re-checked)

Writes ``trained_256_run.npz`` (``--far``: ``trained_far_run.npz``, the task and reference runs of ``trained_far.npz``'s recorded run):
* ``batch.uv [100,64,2]``, ``batch.pose [100,4,4]``, ``batch.intrinsics [100,4,4]`` (one per batch: every ray of a batch shares them),
  ``batch.rgb [100,64,3]``, ``batch.depth [100,64,1]``;
* ``init.head_weight [3,256]``, ``init.head_bias [3]``: the recentred vector head of the student (everything else of the initial state is
  seed + default init x gain, rebuilt by ``vf_nerf_amd`` and checked against ``init.checksum``);
* ``runs.loss [R,1200]``, ``runs.terms [R,1200,6]``, ``runs.clip [R,1200]``, ``runs.psnr_before_after [R,2]``, ``runs.seeds [R,2]``
  (row 0 = the recorded run of ``trained_256.npz``), ``runs.reproduces_recorded`` (largest |difference| of row 0 to the stored curve);
* ``train_recipe`` (repr of the recipe dict).
"""
from __future__ import annotations

import importlib.util
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(HERE, f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


mt = _load("make_trained_golden")
mtg = mt.mtg
from vf_nerf_amd import synthetic  # noqa: E402

EXTRA_SEEDS = ((4100, 3025), (5100, 4025), (6100, 5025), (7100, 6025), (8100, 7025), (9100, 8025), (10100, 9025), (11100, 10025),
               (12100, 11025), (13100, 12025))     # (torch, numpy) of the further streams


def main() -> None:
    torch.set_num_threads(8)
    streams = int(sys.argv[sys.argv.index("--streams") + 1]) if "--streams" in sys.argv else 4
    far = "--far" in sys.argv
    fx = mt.TRAIN_FAR if far else mt.TRAIN
    recorded = np.load(os.path.join(HERE, "trained_far.npz" if far else "trained_256.npz"))
    assert repr(fx) == str(recorded["train_recipe"]), "the recipe in make_trained_golden.py is not the recorded run's"
    teacher = mtg.build_reference_model(dict(fx, seed=fx["teacher_seed"]))
    batches = mt.make_batches(fx, teacher)
    student0 = mtg.build_reference_model(fx)
    chk = synthetic.weights_checksum({"vf": student0.vector_field_network.state_dict(), "rn": student0.rendering_network.state_dict(),
                                      "density": student0.density.state_dict()})
    head = student0.vector_field_network.layers[8]
    arrays = {"init.head_weight": head.weight[:3].detach().numpy().copy(), "init.head_bias": head.bias[:3].detach().numpy().copy(),
              "init.checksum": np.array([chk["sum"], chk["abs_sum"], chk["count"]], dtype=np.float64)}
    for b in batches:
        assert bool((b["pose"][0] == b["pose"][0][:1]).all()) and bool((b["intrinsics"][0] == b["intrinsics"][0][:1]).all())
    arrays["batch.uv"] = np.stack([b["uv"][0].numpy() for b in batches])
    arrays["batch.pose"] = np.stack([b["pose"][0][0].numpy() for b in batches])
    arrays["batch.intrinsics"] = np.stack([b["intrinsics"][0][0].numpy() for b in batches])
    arrays["batch.rgb"] = np.stack([b["rgb"][0].numpy() for b in batches])
    arrays["batch.depth"] = np.stack([b["depth"][0].numpy() for b in batches])

    path = os.path.join(HERE, "trained_far_run.npz" if far else "trained_256_run.npz")
    append = int(sys.argv[sys.argv.index("--append") + 1]) if "--append" in sys.argv else 0
    runs = {"loss": [], "terms": [], "clip": [], "psnr_before_after": []}
    if append:
        have = np.load(path)
        for k in ("batch.uv", "batch.pose", "batch.intrinsics", "batch.rgb", "batch.depth", "init.head_weight", "init.head_bias", "init.checksum"):
            assert np.array_equal(have[k], arrays[k]), f"the regenerated task differs from the fixture's at {k}"
        done = [tuple(int(v) for v in row) for row in have["runs.seeds"]]
        for k in runs:
            runs[k] = [row for row in have[f"runs.{k}"]]
        arrays["runs.reproduces_recorded"] = have["runs.reproduces_recorded"]
        todo = [sd for sd in EXTRA_SEEDS if sd not in done][:append]
        seeds = done + todo
    else:
        done = []
        seeds = todo = [(fx["torch_seed"], fx["numpy_seed"])] + list(EXTRA_SEEDS[:streams - 1])
    for ts, ns in todo:
        r = len(runs["loss"])
        t0 = time.time()
        _, stats = mt.train(fx, batches=batches, torch_seed=ts, numpy_seed=ns, quiet=True)
        for k in runs:
            runs[k].append(stats[f"curve.{k}"])
        print(f"stream {r} (torch seed {ts}, numpy seed {ns}): loss {stats['curve.loss'][:100].mean():.4f} -> {stats['curve.loss'][-100:].mean():.4f}, "
              f"PSNR vs teacher {stats['curve.psnr_before_after']}, {time.time() - t0:.0f} s", flush=True)
        if r == 0:
            gap = float(np.abs(stats["curve.loss"] - recorded["curve.loss"]).max())
            gap_psnr = float(np.abs(stats["curve.psnr_before_after"] - recorded["curve.psnr_before_after"]).max())
            print(f"  recorded run reproduced: max |loss - recorded| {gap:.3e} over {len(stats['curve.loss'])} steps, PSNR difference {gap_psnr:.3e}", flush=True)
            # the first hundred steps must be the recorded ones to rounding (a different batch or initial weight would show at step 0);
            # later steps may part through thread-order rounding of the CPU GEMMs — reported, not required
            assert float(np.abs(stats["curve.loss"][:100] - recorded["curve.loss"][:100]).max()) < 1e-3, "the regenerated task is not the recorded run's"
            arrays["runs.reproduces_recorded"] = np.array([gap, gap_psnr])
        # (written after every run: a half-hour run is not lost to an interruption)
        snap = dict(arrays)
        for k, v in runs.items():
            snap[f"runs.{k}"] = np.stack(v)
        snap["runs.seeds"] = np.array(seeds[:len(runs["loss"])], dtype=np.int64)
        snap["runs.term_names"] = np.array(["rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss", "directional_derivatives_loss"])
        snap["train_recipe"] = np.array(repr(fx))
        np.savez_compressed(path, **snap)
    for k, v in runs.items():
        arrays[f"runs.{k}"] = np.stack(v)
    arrays["runs.seeds"] = np.array(seeds, dtype=np.int64)
    arrays["runs.term_names"] = np.array(["rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss", "directional_derivatives_loss"])
    arrays["train_recipe"] = np.array(repr(fx))
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), {len(seeds)} reference runs")


if __name__ == "__main__":
    main()
