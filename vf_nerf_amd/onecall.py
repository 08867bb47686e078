"""The trainer's loop body as ONE C call (``vfn_train_step``, csrc/vfn_train.hip) — the host side of ``trainer.TrainStep``.

``trainer.TrainStep`` issues a step as ~60 Python-side operations (autograd functions, ctypes calls, allocations): 2.2 ms of host
time at the reference's 1 024-ray batches (train/vector_field_nerf_train.py:172-260), which is what the device needs for the
step's kernels.  ``OneCallStep`` fills, per step, what changes in the two POD structs the model's ``stepengine.StepEngine`` owns (the
batch's pointers, the random-stream positions, Adam's bias corrections) and makes one call.  The launches, their order and therefore
every value are those of the launch-by-launch path (tests/test_hip_trainer.py::test_one_call_training_step_equals_the_launch_by_launch_step).

It serves the shipped regime only — eval-mode networks on the f16x3 kernels, fragment-ordered workspace, fused device loss, flat
optimizer, ``white = False`` — and says so through ``applicable``; anything else takes ``TrainStep``'s Python path.  (The reference's
OWN loop, which makes these calls one by one, reaches the same kernels through ``stepengine.StepSession``.)
"""
from __future__ import annotations

from typing import Optional

import torch

from . import lib
from . import loss as vloss
from . import supervision
from .render_output import NerfOutput, RepeatedRows
from .stepengine import StepEngine, _p, _round32, checked


class OneCallStep:
    def __init__(self, step) -> None:
        self.step = step                    # the owning trainer.TrainStep
        self.model = step.model
        self.engine = StepEngine.of(step.model)
        self.why_not: Optional[str] = None  # the reason the last call did not take this path (diagnostics / tests)

    @property
    def params(self):
        return self.engine.params

    @property
    def io(self):
        return self.engine.io

    # ---------------------------------------------------------------------------------------------
    def applicable(self, pose, white: bool, n: int) -> bool:
        model, step = self.model, self.step

        def no(reason: str) -> bool:
            self.why_not = reason
            return False

        if not getattr(model, "one_call_train_step", True):
            return no("model.one_call_train_step is off")
        crit = step.criterion
        if pose.is_cuda and not white and torch.is_grad_enabled() and \
                (not isinstance(crit, vloss.VFLoss) or not getattr(crit, "fused", False) or step.compact_selection):
            return no("the loss is not the fused device VFLoss")
        s_t = model.ray_sampler.N_samples + min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        reason = self.engine.model_reason(pose, white, n, _round32(2 * ((n * s_t) // 10)))
        if reason is not None:
            return no(reason)
        # the all-reduce of a multi-rank step runs on the bucket's buffer; phase 2 clips and applies the optimizer's flat gradient: they
        # must be one buffer (a bucket built before the optimizer had its flat storage owns another one)
        bucket = step.bucket
        flat = getattr(bucket, "flat", None)            # (distributed.GradientBucket; a stand-in without a buffer of its own has nothing to diverge)
        if flat is not None:
            f = model.optimizer.flat()
            if f is None or flat.data_ptr() != f["grad"].data_ptr():
                return no("the gradient bucket is not the optimizer's flat gradient buffer")
        self.why_not = None
        return True

    # ---------------------------------------------------------------------------------------------
    def run(self, pose, pixels, intrinsics, rgb_gt, depth_gt, epoch: int, uniforms):
        model, step, eng = self.model, self.step, self.engine
        cfg = model.config
        dev = pose.device
        opt = model.optimizer
        eng.supersede()
        f = eng.bind(dev)
        pr, io = eng.params, eng.io
        # ---- render parameters ----------------------------------------------------------------------------------------------------------
        # streams = 2: the supervision batch's forward and chain on a side stream inside the call, beside the fine pass's (same values)
        n, s_c, n_f, keep_render = eng.fill_render(pose, pixels, intrinsics, epoch, uniforms, int(getattr(model, "train_step_streams", 2)))
        s_t, m = s_c + n_f, n * (s_c + n_f)
        n_sup = m // 10

        # ---- supervision --------------------------------------------------------------------------------------------------------------
        pr.n_sup, pr.border, pr.center = n_sup, int(bool(cfg.border_supervision)), int(bool(cfg.center_supervision))
        pr.sup_rows_reserved = 0
        for i in range(3):
            pr.sup_centroid[i] = step.centroid_host[i]
        pr.sup_radius, pr.border_r_min, pr.border_r_max = step.radius, step.far - 5 * step.radius, step.far
        replay = []
        for on in (pr.border, pr.center):
            u = None
            if on and supervision._replay:
                u = supervision._replay.pop(0)
                if tuple(u.shape) != (n_sup, 3):
                    raise ValueError(f"replayed uniforms have shape {tuple(u.shape)}, the sampler needs {(n_sup, 3)}")
                u = u.to(dev).float().contiguous()
            replay.append(u)
        pr.sup_seed, pr.sup_offset = supervision._seed & (2 ** 64 - 1), supervision._offset & (2 ** 64 - 1)
        supervision._offset += n_sup * sum(1 for on, u in zip((pr.border, pr.center), replay) if on and u is None)

        # ---- loss ---------------------------------------------------------------------------------------------------------------------
        crit = step.criterion
        w = crit.weights
        lp = pr.loss
        has_depth = depth_gt is not None and depth_gt.nelement() > 0
        lp.has_depth, lp.smaller_on = int(has_depth), int(epoch >= crit.config.norm_smaller_than_one_start)
        lp.w_rgb, lp.w_depth, lp.w_unit, lp.w_sup, lp.w_smaller = float(w.rgb), float(w.depth), float(w.unit_norm), float(w.supervision), \
            float(w.norm_smaller_than_one)
        lp.depth_clamp = float(crit.config.depth_loss_clamp)
        lp.ray_center = pr.center
        lp.radius = step.radius
        for i in range(3):
            lp.centroid[i] = step.centroid_host[i]
        rgb_gt = rgb_gt.to(dev).reshape(-1, 3).float().contiguous()
        if rgb_gt.shape[0] != n or (has_depth and depth_gt.numel() != n):
            raise ValueError(f"TrainStep: targets of {rgb_gt.shape[0]} rays (depth: {depth_gt.numel() if has_depth else 0}) for {n} rays")
        depth_gt = depth_gt.to(dev).reshape(-1).float().contiguous() if has_depth else None

        # ---- storage forms, packs, optimizer scalars ----------------------------------------------------------------------------------
        single = eng.fill_storage()
        eng.fill_optimizer(f)

        # ---- buffers ------------------------------------------------------------------------------------------------------------------
        key = (n, s_c, n_f, n_sup, pr.border, pr.center, pr.sparse_colours, str(dev))
        ws = eng.workspace(key, dev)
        views = eng.outputs(n, s_t, dev)
        ray_dirs, z, pts, normals, colors, weights, rgb, depth, out_terms, out_norm = views
        io.sup_u_border, io.sup_u_center = _p(replay[0]), _p(replay[1])
        io.rgb_gt, io.depth_gt = checked(rgb_gt, "rgb ground truth", dev, n), checked(depth_gt, "depth ground truth", dev, n)
        io.workspace = _p(ws)
        io.d_rgb_in = io.d_depth_in = io.d_normals_in = None
        keep_alive = (keep_render, replay, rgb_gt, depth_gt)     # until the call has been issued

        # ---- the call(s) --------------------------------------------------------------------------------------------------------------
        guard = model.range_guard
        watching = guard.active()
        if watching:
            guard.poll()

        def call(phases):
            pr.phases = phases
            lib.train_step(pr, io)

        bucket = step.bucket
        first = lib.TRAIN_FORWARD_BACKWARD if bucket is not None else (lib.TRAIN_FORWARD_BACKWARD | lib.TRAIN_OPTIMIZER)
        if watching:
            with guard.watch(dev):
                call(first)
        else:
            call(first)
        if bucket is not None:
            bucket.all_reduce_mean()
            call(lib.TRAIN_OPTIMIZER)
        del keep_alive
        opt._opt_called = True                   # (what torch's LRScheduler looks at before it warns about the call order)
        opt.finish_step(f)                       # step counters; invalidates the packs' keys (the parameters changed under them) ...
        eng.mark_packs_current(single)           # ... and the call has already re-packed them
        model.scheduler.step()

        step.last_total_norm = out_norm[0]
        step.last_colour_counts = out_norm[2:4]       # device [selected, all]: read (synchronising) only by whoever reports it
        rep_dirs = RepeatedRows(ray_dirs.view(n, 3), s_t)
        step.last_outputs = NerfOutput(points_coarse=pts.view(n, s_t, 3), points_fine=None, coarse_normals=normals.view(n, s_t, 3),
                                       coarse_rgb_values=rgb.view(n, 3), coarse_depth_map=depth.view(n, 1), fine_normals=None, fine_rgb_values=None,
                                       fine_depth_map=None, z_vals=z.view(n, s_t), directional_derivtives=None, ray_dirs=rep_dirs,
                                       coarse_colors=colors.view(m, 3))
        # (a fresh little tensor per step, stepengine.outputs: this step's seven scalars are read back only when somebody needs a value)
        from .deferred import DeviceScalars, as_loss
        holder = DeviceScalars(out_terms)
        return (as_loss(holder.dev[6], holder, 6) if vloss.DEFERRED_SCALARS else holder.dev[6]), vloss._LazyTerms(vloss._NAMES, None, holder=holder)
