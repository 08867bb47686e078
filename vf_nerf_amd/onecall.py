"""The trainer's loop body as ONE C call (``vfn_train_step``, csrc/vfn_train.hip) — the host side.

``trainer.TrainStep`` issues a step as ~60 Python-side operations (autograd functions, ctypes calls, allocations): 2.2 ms of host
time at the reference's 1 024-ray batches (train/vector_field_nerf_train.py:172-260), which is what the device needs for the
step's kernels.  ``OneCallStep`` prepares, once, the two POD structs ``vfn_train_step`` takes (every pointer of the model's
parameters, gradients, optimizer buffers, weight packs; one persistent workspace) and per step only fills in what changes: the
batch's pointers, the random-stream positions, Adam's bias corrections.  The launches, their order and therefore every value are
those of the launch-by-launch path (tests/test_hip_trainer.py::test_one_call_training_step_equals_the_launch_by_launch_step).

It serves the shipped regime only — eval-mode networks on the f16x3 kernels, fragment-ordered workspace, fused device loss, flat
optimizer, ``white = False`` — and says so through ``applicable``; anything else takes ``TrainStep``'s Python path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import lib
from . import loss as vloss
from . import supervision
from .render_output import NerfOutput, RepeatedRows


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class OneCallStep:
    def __init__(self, step) -> None:
        self.step = step                    # the owning trainer.TrainStep
        self.model = step.model
        self._built_for = None
        self._ws: Dict[tuple, torch.Tensor] = {}
        self.params = lib.TrainStepParams()
        self.io = lib.TrainStepIO()
        self._keep = []                     # Python objects whose memory the structs point into
        self.why_not: Optional[str] = None  # the reason the last call did not take this path (diagnostics / tests)

    # ---------------------------------------------------------------------------------------------
    def applicable(self, pose, white: bool, n: int) -> bool:
        model, step = self.model, self.step
        cfg = model.config
        vf, rn = model.vector_field_network, model.rendering_network

        def no(reason: str) -> bool:
            self.why_not = reason
            return False

        if not getattr(model, "one_call_train_step", True):
            return no("model.one_call_train_step is off")
        if not pose.is_cuda or white or not torch.is_grad_enabled():
            return no("host tensors, white background or gradients disabled")
        crit = step.criterion
        if not isinstance(crit, vloss.VFLoss) or not getattr(crit, "fused", False) or step.compact_selection:
            return no("the loss is not the fused device VFLoss")
        if cfg.numerical_jacobian or cfg.rendering != "volsdf" or not cfg.ray_sampler_config.fine_sampling():
            return no("numerical Jacobian / rendering mode / no fine sampling")
        if vf.training or rn.training or rn._batch_statistics() or not (vf.supports_fused() and rn.supports_fused()) or not rn.config.detach_normals:
            return no("a network in training mode, an unsupported geometry or attached normals")
        if not model.uses_f16x3() or model.f16x3_guard == "strict" or getattr(model, "_keep_saved", False):
            return no("not on the f16x3 kernels, strict guard or a test hook")
        if model.workspace_layout != "fragment" or not getattr(model, "shared_step_workspace", True) or not model.reuse_proposal or \
                getattr(model, "backward_kernels", "auto") == "fp32":
            return no("workspace layout / sharing switched off")
        from .backward import StoredFinePass, _direct_ok
        s_c = model.ray_sampler.N_samples
        n_f = min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        if not StoredFinePass.applicable(model, n, s_c, n_f) or (n * (s_c + n_f)) % 32 or n_f < 2:
            return no("sample counts are not whole groups of 32 points")
        n_sup = (n * (s_c + n_f)) // 10
        pad = (2 * n_sup + 31) // 32 * 32
        sparse = bool(getattr(model, "sparse_colour_training", True))
        if n * (s_c + n_f) * (2 if sparse else 1) + pad >= (1 << 21):
            return no("too many points for one fragment-ordered workspace")
        from .optim import FlatAdam
        opt = model.optimizer
        if not isinstance(opt, FlatAdam) or opt.flat() is None or not opt.regions_for(model.parameters()):
            return no("the optimizer is not the flat Adam over exactly model.parameters()")
        if not (_direct_ok(vf, pose.device) and _direct_ok(rn, pose.device)):
            return no("a parameter is frozen, hooked or without a flat gradient view")
        if any(not p.requires_grad for p in model.density.parameters()) or not hasattr(model.density, "scale"):
            return no("density scalars frozen or absent")
        self.why_not = None
        return True

    # ---------------------------------------------------------------------------------------------
    def _build(self, f, dev) -> None:
        """Everything that only changes when a buffer moves: layer tables, gradient targets, flat buffers."""
        from .backward import _layer_table, _head_rows
        model = self.model
        vf, rn = model.vector_field_network, model.rendering_network
        io = self.io
        self._keep = []

        def layer_array(net):
            arr = lib._layer_array(net.geometry(), [{k: v.detach() for k, v in d.items()} for d in net._layer_tensors()])
            self._keep.append(arr)
            return C.cast(arr, C.c_void_p)

        vf_geom, rn_geom = vf.geometry(), rn.geometry()
        self._keep += [vf_geom, rn_geom]
        io.vf_geom, io.rn_geom = C.cast(C.pointer(vf_geom), C.c_void_p), C.cast(C.pointer(rn_geom), C.c_void_p)
        io.vf_layers, io.rn_layers = layer_array(vf), layer_array(rn)
        vf_tab, rn_tab = _layer_table(vf), _layer_table(rn)
        self._keep += [vf_tab, rn_tab]
        io.vf_wgrad, io.rn_wgrad = C.cast(vf_tab, C.c_void_p), C.cast(rn_tab, C.c_void_p)
        io.vf_head_w, io.rn_head_w = _p(_head_rows(vf)), _p(_head_rows(rn))
        d = model.density
        io.beta, io.mean, io.scale = _p(d.beta), _p(d.mean), _p(d.scale)
        io.g_beta, io.g_mean, io.g_scale = _p(d.beta.grad), _p(d.mean.grad), _p(d.scale.grad)
        io.flat_param, io.flat_grad, io.exp_avg, io.exp_avg_sq = _p(f["param"]), _p(f["grad"]), _p(f["exp_avg"]), _p(f["exp_avg_sq"])
        io.n_flat = f["param"].numel()
        io.clip_workspace = _p(f["workspace"])
        pr = self.params
        pr.n_regions = len(f["regions"])
        for i, (start, end, mult) in enumerate(f["regions"]):
            pr.starts[i], pr.ends[i], pr.mults[i] = int(start), int(end), int(mult)
        self._built_for = (id(f), f["param"].data_ptr(), f["grad"].data_ptr(), str(dev))

    def _packs(self, single: bool):
        """The four weight packs, current for the parameters as they are now (re-packed here only when something other than
        vfn_train_step's own re-pack changed them)."""
        from .backward import _packed_bwd16
        vf, rn = self.model.vector_field_network, self.model.rendering_network
        return vf.packed16_weights(), rn.packed16_weights(), _packed_bwd16(vf, single), _packed_bwd16(rn, single)

    def _mark_packs_current(self, single: bool) -> None:
        """vfn_train_step re-packed all four packs from the updated parameters: give the caches the key they would compute."""
        name = "_packed_bwd16r_cache" if single else "_packed_bwd16_cache"
        for net in (self.model.vector_field_network, self.model.rendering_network):
            _, key = net._pack_key()
            net._packed16_cache = (key, net._packed16_cache[1])
            setattr(net, name, (key, getattr(net, name)[1]))

    # ---------------------------------------------------------------------------------------------
    def run(self, pose, pixels, intrinsics, rgb_gt, depth_gt, epoch: int, uniforms):
        model, step = self.model, self.step
        cfg = model.config
        dev = pose.device
        opt = model.optimizer
        f = opt.flat()
        opt._rebind_grads(f)
        if self._built_for != (id(f), f["param"].data_ptr(), f["grad"].data_ptr(), str(dev)):
            self._build(f, dev)
        pr, io = self.params, self.io
        from .backward import _storage, _train_products
        n = pixels.shape[0]
        s_c = model.ray_sampler.N_samples
        n_f = min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        s_t, m = s_c + n_f, n * (s_c + n_f)
        n_sup = m // 10
        uniforms = uniforms or {}

        # ---- render parameters (as VectorFieldNerf._render_one_call fills them) ---------------------------------------------------------
        pose, intrinsics = model._per_ray_camera(pose, intrinsics, n)
        model._anneal(epoch, dev)
        far_c, far_ct = model._far_args(model.ray_sampler.far)
        far_f, far_ft = model._far_args(model.fine_sampler.far)
        rng = float(model.fine_sampler.range)
        perturb_c, perturb_f = not model.ray_sampler.deterministic, not model.fine_sampler.deterministic
        rp = pr.render
        rp.n_rays, rp.n_coarse, rp.n_fine = n, s_c, n_f
        rp.pose_is_quat = int(pose.dim() == 2 and pose.shape[1] == 7)
        rp.perturb_coarse, rp.perturb_fine = int(perturb_c), int(perturb_f)
        rp.near_coarse, rp.near_fine = float(model.ray_sampler.near), float(model.fine_sampler.near)
        rp.far_coarse, rp.far_fine = (0.0 if far_ct is not None else far_c), (0.0 if far_ft is not None else far_f)
        rp.fine_range, rp.window_step = rng, 2 * rng / (n_f - 1)
        rp.span = (far_f - float(model.fine_sampler.near)) if far_ft is None else 0.0
        rp.density = model._density_params()
        # 2: the supervision batch's forward and chain on a side stream inside the call, beside the fine pass's (same values)
        rp.streams = int(getattr(model, "train_step_streams", 2))

        def given(name, needed):
            return uniforms[name].to(dev).float().contiguous() if (needed and name in uniforms) else None

        u_c, u_f, u_a = given("u_coarse", perturb_c), given("u_fine", perturb_f), given("u_add", True)
        generated = (n * s_c if (perturb_c and u_c is None) else 0) + (n * n_f if (perturb_f and u_f is None) else 0) + (n * n_f if u_a is None else 0)
        rp.seed, rp.offset = model.rng_seed & (2 ** 64 - 1), model._rng_offset & (2 ** 64 - 1)
        model._rng_offset += (generated + 3) // 4

        # ---- supervision --------------------------------------------------------------------------------------------------------------
        pr.n_sup, pr.border, pr.center = n_sup, int(bool(cfg.border_supervision)), int(bool(cfg.center_supervision))
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
        rgb_gt = rgb_gt.reshape(-1, 3).float().contiguous()
        if rgb_gt.shape[0] != n or (has_depth and depth_gt.numel() != n):
            raise ValueError(f"TrainStep: targets of {rgb_gt.shape[0]} rays (depth: {depth_gt.numel() if has_depth else 0}) for {n} rays")
        depth_gt = depth_gt.reshape(-1).float().contiguous() if has_depth else None

        # ---- storage forms, packs, optimizer scalars ----------------------------------------------------------------------------------
        f16, frag, dy16 = _storage(model, True)
        single = dy16 == "f16p1"
        if single:
            dy16 = "f16"
        pr.save_flags = (lib.WS_F16 if f16 else 0) | lib.WS_FRAG | (lib.WS_P1 if single else 0)
        pr.dy_flags = lib.DY_FRAG | {None: 0, "bf16": lib.DY_BF16, "f16": lib.DY_F16S}[dy16] | (lib.DY_P1 if single else 0)
        pr.dy_form = {None: lib.DYF_FRAG32, "bf16": lib.DYF_FRAGBF16, "f16": lib.DYF_FRAGF16S}[dy16]
        pr.x_form = lib.XF_FRAG16 if f16 else lib.XF_FRAG32
        pr.forward_products = 1 if single else _train_products(model)
        vf16, rn16, vfb, rnb = self._packs(single)
        io.vf_packed16, io.rn_packed16, io.vf_packed_bwd16, io.rn_packed_bwd16 = _p(vf16), _p(rn16), _p(vfb), _p(rnb)
        group = opt.param_groups[0]
        beta1, beta2, step_size, bc2 = opt.step_scalars(f)
        for i, (a, b) in enumerate(zip(step_size, bc2)):
            pr.step_size[i], pr.bc2_sqrt[i] = a, b
        pr.beta1, pr.beta2, pr.eps, pr.weight_decay = beta1, beta2, group["eps"], group["weight_decay"]
        pr.max_norm = float(cfg.scheduler_config.clip_norm)
        pr.repack = 1
        # the colour branch only where a sample's weight is non-zero (exact: include/vfn.h, vfn_train_step); False: dense, as the Python path
        pr.sparse_colours = int(bool(getattr(model, "sparse_colour_training", True)))

        # ---- buffers ------------------------------------------------------------------------------------------------------------------
        key = (n, s_c, n_f, n_sup, pr.border, pr.center, pr.sparse_colours, str(dev))
        ws = self._ws.get(key)
        if ws is None:
            # ONE workspace stays alive: ~34 KiB per sample dense, ~61 KiB with the sparse colour branch (region 2 is sized for every sample
            # being selected: 17 / 30 GiB at 4096 x 128 of the 288 GB) — the reference's trainer draws batches of one size
            self._ws.clear()
            need = lib.train_step_workspace_bytes(pr, model.vector_field_network.geometry(), model.rendering_network.geometry())
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        # the step's outputs: one allocation, sliced (ray_dirs, z_vals, points, normals, colors, weights, rgb, depth, out_terms, out_norm)
        sizes = (n * 3, n * s_t, m * 3, m * 3, m * 3, n * s_t, n * 3, n, 8, 4)
        flat = torch.empty(sum(sizes), device=dev)
        views, o = [], 0
        for k in sizes:
            views.append(flat[o:o + k])
            o += k
        ray_dirs, z, pts, normals, colors, weights, rgb, depth, out_terms, out_norm = views
        uv = pixels.float().contiguous()
        io.uv, io.pose, io.intrinsics = _p(uv), _p(pose), _p(intrinsics)
        io.t_vals = _p(model._linspace(s_c, dev))
        io.far_coarse_per_ray, io.far_fine_per_ray = _p(far_ct), _p(far_ft)
        io.u_coarse, io.u_fine, io.u_add = _p(u_c), _p(u_f), _p(u_a)
        io.sup_u_border, io.sup_u_center = _p(replay[0]), _p(replay[1])
        io.rgb_gt, io.depth_gt = _p(rgb_gt), _p(depth_gt)
        io.workspace = _p(ws)
        io.ray_dirs, io.z_vals, io.points, io.normals, io.colors, io.weights, io.rgb, io.depth = (_p(t) for t in views[:8])
        io.out_terms, io.out_norm = _p(out_terms), _p(out_norm)
        io.out_counts = out_norm.data_ptr() + 8           # [2:4] of the same little tensor: samples the colour branch ran on, all samples
        keep_alive = (uv, pose, intrinsics, u_c, u_f, u_a, replay, rgb_gt, depth_gt, far_ct, far_ft)     # until the call has been issued

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
        self._mark_packs_current(single)         # ... and the call has already re-packed them
        model.scheduler.step()

        step.last_total_norm = out_norm[0]
        step.last_colour_counts = out_norm[2:4]       # device [selected, all]: read (synchronising) only by whoever reports it
        rep_dirs = RepeatedRows(ray_dirs.view(n, 3), s_t)
        step.last_outputs = NerfOutput(points_coarse=pts.view(n, s_t, 3), points_fine=None, coarse_normals=normals.view(n, s_t, 3),
                                       coarse_rgb_values=rgb.view(n, 3), coarse_depth_map=depth.view(n, 1), fine_normals=None, fine_rgb_values=None,
                                       fine_depth_map=None, z_vals=z.view(n, s_t), directional_derivtives=None, ray_dirs=rep_dirs,
                                       coarse_colors=colors.view(m, 3))
        return out_terms[6], vloss._LazyTerms(vloss._NAMES, out_terms[:6])
