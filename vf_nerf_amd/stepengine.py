"""The step engine: ONE owner per model of ``vfn_train_step``'s two POD structs and its persistent workspace, and the two ways a
training step reaches them.

* ``trainer.TrainStep`` (``onecall.OneCallStep``) hands the whole loop body to C in one call (phases FORWARD_BACKWARD | OPTIMIZER).
* The reference's own trainer makes the calls of that loop body ONE BY ONE (train/vector_field_nerf_train.py:177-260, unchanged, through
  ``vf_nerf_amd.dropin``): ``model.render`` -> ``functions.sample_border_points`` -> ``vector_field_network(points)[:, :3]`` ->
  ``functions.get_center_indices_and_gt`` -> ``functions.sample_center_points`` -> ``vector_field_network(points)[:, :3]`` -> ``VFLoss`` ->
  ``optimizer.zero_grad`` -> ``backward`` -> ``clip_grad_norm_`` -> ``optimizer.step``.  A grad-mode ``render()`` in the shipped regime
  opens a ``StepSession`` on the same workspace: the render is ONE C call (VFN_TRAIN_RENDER: the saving forwards with the sparse colour
  branch), the samplers write their points into the workspace's supervision rows, the vector-field forwards on those points are
  vector-only saving forwards into the same rows (on the engine's side stream, beside the render's launches), and ``backward()`` reaches
  ONE autograd node whose backward is ONE C call (VFN_TRAIN_BACKWARD: per-ray backward, the chains, every weight gradient).  The loss in
  the middle is whatever the trainer calls (``loss.VFLoss`` under the drop-in: two launches).

Every value is what the launch-by-launch autograd path (backward.py) produces up to the order of the sums; nothing here computes.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, List, Optional, Tuple

import torch

from . import lib
from .render_output import LazyColours, NerfOutput, RepeatedRows


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _round32(n: int) -> int:
    return (n + 31) // 32 * 32


def checked(t: Optional[torch.Tensor], name: str, dev, rows: Optional[int] = None, dtype=torch.float32) -> Optional[int]:
    """data_ptr() of a batch tensor the C call will read through a raw pointer: it must live on the step's device, be contiguous and of
    the dtype (and, where given, the row count) the kernels assume — a host tensor or a short batch would otherwise be an out-of-bounds
    device access instead of an exception."""
    if t is None:
        return None
    if not t.is_cuda or t.device != dev:
        raise lib.VfnError(f"{name}: expected a tensor on {dev}, got {t.device}")
    if t.dtype != dtype or not t.is_contiguous():
        raise lib.VfnError(f"{name}: expected a contiguous {dtype} tensor, got {t.dtype}, contiguous={t.is_contiguous()}")
    if rows is not None and (t.dim() == 0 or t.shape[0] != rows):
        raise lib.VfnError(f"{name}: expected {rows} rows, got shape {tuple(t.shape)}")
    return t.data_ptr()


_current: Optional["weakref.ReferenceType[StepSession]"] = None


def current_session() -> Optional["StepSession"]:
    """The step session the last grad-mode render() opened, while it is still open (its backward has not run, nothing replaced it)."""
    s = _current() if _current is not None else None
    return s if (s is not None and s.open) else None


def host_centroid(centroid) -> Optional[Tuple[float, float, float]]:
    """The three coordinates of a centroid WITHOUT a device synchronisation: a CPU tensor / sequence, or a device tensor that carries
    them (``dropin.cached_centroid`` attaches ``_vfn_host`` to what the datasets' ``get_centroid(device)`` returns).  None otherwise."""
    host = getattr(centroid, "_vfn_host", None)
    if host is not None:
        return host
    if isinstance(centroid, torch.Tensor):
        if centroid.is_cuda or centroid.numel() != 3:
            return None
        return tuple(float(v) for v in centroid.reshape(3).tolist())
    try:
        vals = tuple(float(v) for v in centroid)
    except TypeError:
        return None
    return vals if len(vals) == 3 else None


class StepEngine:
    def __init__(self, model) -> None:
        self._model_ref = weakref.ref(model)     # (the model owns the engine: no reference cycle keeps a 30 GB workspace waiting for the cyclic collector)
        self.params = lib.TrainStepParams()
        self.io = lib.TrainStepIO()
        self._built_for = None
        self._keep: list = []               # Python objects whose memory the structs point into
        self._ws: Dict[tuple, torch.Tensor] = {}
        self._layouts: Dict[tuple, List[int]] = {}
        self.session: Optional[StepSession] = None
        self.why_not: Optional[str] = None  # the reason the last step did not take the C path (diagnostics / tests)

    @property
    def model(self):
        return self._model_ref()

    @staticmethod
    def of(model) -> "StepEngine":
        eng = getattr(model, "_step_engine", None)
        if eng is None:
            eng = model._step_engine = StepEngine(model)
        return eng

    # ---------------------------------------------------------------------------------------------
    def model_reason(self, pose, white: bool, n: int, extra_rows: int) -> Optional[str]:
        """Why the model / the batch cannot take the C step (None: it can).  ``extra_rows``: supervision rows beside the N S_t samples."""
        model = self.model
        cfg = model.config
        vf, rn = model.vector_field_network, model.rendering_network
        if not pose.is_cuda or white or not torch.is_grad_enabled():
            return "host tensors, white background or gradients disabled"
        if cfg.numerical_jacobian or cfg.rendering != "volsdf" or not cfg.ray_sampler_config.fine_sampling():
            return "numerical Jacobian / rendering mode / no fine sampling"
        if vf.training or rn.training or rn._batch_statistics() or not (vf.supports_fused() and rn.supports_fused()) or not rn.config.detach_normals:
            return "a network in training mode, an unsupported geometry or attached normals"
        if not model.uses_f16x3() or model.f16x3_guard == "strict" or getattr(model, "_keep_saved", False):
            return "not on the f16x3 kernels, strict guard or a test hook"
        if model.workspace_layout != "fragment" or not getattr(model, "shared_step_workspace", True) or not model.reuse_proposal or \
                getattr(model, "backward_kernels", "auto") == "fp32":
            return "workspace layout / sharing switched off"
        from .backward import StoredFinePass, _direct_ok
        s_c = model.ray_sampler.N_samples
        n_f = min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        if not StoredFinePass.applicable(model, n, s_c, n_f) or (n * (s_c + n_f)) % 32 or n_f < 2:
            return "sample counts are not whole groups of 32 points"
        # (a launch addresses its part of a slot with 32-bit offsets: the samples + the supervision rows of one chain launch < 2^21 points;
        #  the workspace itself — with the sparse colour branch's second region — may hold more since round 5)
        if n * (s_c + n_f) + extra_rows >= (1 << 21):
            return "too many points for one launch over a fragment-ordered slot"
        from .optim import FlatAdam
        opt = model.optimizer
        if not isinstance(opt, FlatAdam) or opt.flat() is None or not opt.regions_for(model.parameters()):
            return "the optimizer is not the flat Adam over exactly model.parameters()"
        if not (_direct_ok(vf, pose.device) and _direct_ok(rn, pose.device)):
            return "a parameter is frozen, hooked or without a flat gradient view"
        if any(not p.requires_grad for p in self._anchors()[2:]) or not hasattr(model.density, "scale"):
            return "density scalars frozen or absent"
        return None

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

    def bind(self, dev):
        """The optimizer's flat buffers, every gradient rebound to its view, the structs' fixed pointers current.  -> the flat dict."""
        opt = self.model.optimizer
        f = opt.flat()
        opt._rebind_grads(f)
        if self._built_for != (id(f), f["param"].data_ptr(), f["grad"].data_ptr(), str(dev)):
            self._build(f, dev)
        if opt.step_engine is None or opt.step_engine() is not self:
            opt.step_engine = weakref.ref(self)
        return f

    def packs(self, single: bool):
        """The four weight packs, current for the parameters as they are now (re-packed here only when something other than
        vfn_train_step's own re-pack changed them)."""
        from .backward import _packed_bwd16
        vf, rn = self.model.vector_field_network, self.model.rendering_network
        return vf.packed16_weights(), rn.packed16_weights(), _packed_bwd16(vf, single), _packed_bwd16(rn, single)

    def mark_packs_current(self, single: bool) -> None:
        """vfn_train_step re-packed all four packs from the updated parameters: give the caches the key they would compute."""
        name = "_packed_bwd16r_cache" if single else "_packed_bwd16_cache"
        for net in (self.model.vector_field_network, self.model.rendering_network):
            _, key = net._pack_key()
            net._packed16_cache = (key, net._packed16_cache[1])
            setattr(net, name, (key, getattr(net, name)[1]))

    # ---------------------------------------------------------------------------------------------
    def fill_render(self, pose, pixels, intrinsics, epoch: int, uniforms, streams: int):
        """The render part of the parameter struct and the batch pointers (as VectorFieldNerf._render_one_call fills them).
        -> (n, s_c, n_f, keep_alive): the tensors the pointers refer to."""
        model, pr, io = self.model, self.params, self.io
        dev = pose.device
        n = pixels.shape[0]
        s_c = model.ray_sampler.N_samples
        n_f = min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        uniforms = uniforms or {}
        if pose.dim() >= 2 and pose.shape[0] not in (1, n) and tuple(pose.shape) != (4, 4):
            raise lib.VfnError(f"pose has {pose.shape[0]} rows for {n} rays (one per ray, or one for the batch)")
        if intrinsics.dim() == 3 and intrinsics.shape[0] not in (1, n):
            raise lib.VfnError(f"intrinsics has {intrinsics.shape[0]} rows for {n} rays (one per ray, or one for the batch)")
        pose, intrinsics = model._per_ray_camera(pose.to(dev), intrinsics.to(dev), n)
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
        rp.streams = int(streams)
        # where this step's f16x3 forwards report operands outside their range: a field of the call's struct (no per-thread setter)
        guard = model.range_guard
        rp.status_word = _p(guard._state(dev)["status"]) if guard.active() else None

        def given(name, needed, shape):
            if not (needed and name in uniforms):
                return None
            u = uniforms[name].to(dev).float().contiguous()
            if tuple(u.shape) != shape:
                raise lib.VfnError(f"uniforms[{name!r}] has shape {tuple(u.shape)}, the sampler needs {shape}")
            return u

        u_c, u_f, u_a = given("u_coarse", perturb_c, (n, s_c)), given("u_fine", perturb_f, (n, n_f)), given("u_add", True, (n, n_f))
        generated = (n * s_c if (perturb_c and u_c is None) else 0) + (n * n_f if (perturb_f and u_f is None) else 0) + (n * n_f if u_a is None else 0)
        rp.seed, rp.offset = model.rng_seed & (2 ** 64 - 1), model._rng_offset & (2 ** 64 - 1)
        model._rng_offset += (generated + 3) // 4
        if far_ct is not None:
            far_ct = far_ct.to(dev)
        if far_ft is not None:
            far_ft = far_ft.to(dev)
        uv = pixels.to(dev).float().contiguous()
        io.uv, io.pose, io.intrinsics = checked(uv, "pixels", dev, n), checked(pose, "pose", dev, n), checked(intrinsics, "intrinsics", dev, n)
        io.t_vals = _p(model._linspace(s_c, dev))
        io.far_coarse_per_ray, io.far_fine_per_ray = checked(far_ct, "far (coarse sampler)", dev, n), checked(far_ft, "far (fine sampler)", dev, n)
        io.u_coarse, io.u_fine, io.u_add = _p(u_c), _p(u_f), _p(u_a)
        return n, s_c, n_f, (uv, pose, intrinsics, u_c, u_f, u_a, far_ct, far_ft)

    def fill_storage(self) -> bool:
        """Storage forms of the training workspace, the four packs.  -> single (the one-product mode)."""
        from .backward import _storage, _train_products
        model, pr, io = self.model, self.params, self.io
        f16, frag, dy16 = _storage(model, True)
        single = dy16 == "f16p1"
        if single:
            dy16 = "f16"
        pr.save_flags = (lib.WS_F16 if f16 else 0) | lib.WS_FRAG | (lib.WS_P1 if single else 0)
        pr.dy_flags = lib.DY_FRAG | {None: 0, "bf16": lib.DY_BF16, "f16": lib.DY_F16S}[dy16] | (lib.DY_P1 if single else 0)
        pr.dy_form = {None: lib.DYF_FRAG32, "bf16": lib.DYF_FRAGBF16, "f16": lib.DYF_FRAGF16S}[dy16]
        pr.x_form = lib.XF_FRAG16 if f16 else lib.XF_FRAG32
        pr.forward_products = 1 if single else _train_products(model)
        self._single = single
        vf16, rn16, vfb, rnb = self.packs(single)
        io.vf_packed16, io.rn_packed16, io.vf_packed_bwd16, io.rn_packed_bwd16 = _p(vf16), _p(rn16), _p(vfb), _p(rnb)
        # the colour branch only where a sample's weight is non-zero (the dense step up to one underflow corner: include/vfn.h, vfn_train_step); False: dense, as the Python path
        pr.sparse_colours = int(bool(getattr(model, "sparse_colour_training", True)))
        return single

    def fill_optimizer(self, f) -> None:
        model, pr = self.model, self.params
        opt = model.optimizer
        group = opt.param_groups[0]
        beta1, beta2, step_size, bc2 = opt.step_scalars(f)
        for i, (a, b) in enumerate(zip(step_size, bc2)):
            pr.step_size[i], pr.bc2_sqrt[i] = a, b
        pr.beta1, pr.beta2, pr.eps, pr.weight_decay = beta1, beta2, group["eps"], group["weight_decay"]
        pr.max_norm = float(model.config.scheduler_config.clip_norm)
        pr.repack = 1

    def adam_step(self, f) -> bool:
        """``optimizer.step()`` of the flat Adam as ONE C call (VFN_TRAIN_ADAM) that also re-packs the four weight packs from the updated
        parameters out of the layer tables the structs already hold — what the whole-step call does at its end — instead of a launch for
        the update and, at the next forward, ~0.5 ms of Python-side re-packing.  False (nothing done) unless the structs describe exactly
        these buffers."""
        io, pr = self.io, self.params
        if self._built_for is None or self._built_for[:3] != (id(f), f["param"].data_ptr(), f["grad"].data_ptr()) or pr.render.n_rays <= 0:
            return False
        single = getattr(self, "_single", None)
        if single is None:
            return False
        vf, rn = self.model.vector_field_network, self.model.rendering_network
        name = "_packed_bwd16r_cache" if single else "_packed_bwd16_cache"
        caches = (getattr(vf, "_packed16_cache", None), getattr(rn, "_packed16_cache", None), getattr(vf, name, None), getattr(rn, name, None))
        if any(c is None for c in caches) or tuple(c[1].data_ptr() for c in caches) != (io.vf_packed16, io.rn_packed16, io.vf_packed_bwd16, io.rn_packed_bwd16):
            return False
        opt = self.model.optimizer
        self.fill_optimizer(f)
        pr.phases = lib.TRAIN_ADAM
        lib.train_step(pr, io)
        opt.finish_step(f)                       # step counters; invalidates the packs' keys (the parameters changed under them) ...
        self.mark_packs_current(single)          # ... and the call has already re-packed them
        return True

    def workspace(self, key: tuple, dev) -> torch.Tensor:
        """ONE workspace stays alive: ~34 KiB per sample dense, ~61 KiB with the sparse colour branch (region 2 is sized for every sample
        being selected: 17 / 30 GiB at 4096 x 128 of the 288 GB) — the reference's trainer draws batches of one size."""
        ws = self._ws.get(key)
        if ws is None:
            self._ws.clear()
            self._layouts.clear()
            model = self.model
            need = lib.train_step_workspace_bytes(self.params, model.vector_field_network.geometry(), model.rendering_network.geometry())
            ws = self._ws[key] = torch.empty(need, dtype=torch.uint8, device=dev)
        return ws

    def layout(self, key: tuple) -> List[int]:
        lay = self._layouts.get(key)
        if lay is None:
            model = self.model
            lay = self._layouts[key] = lib.train_step_workspace_layout(self.params, model.vector_field_network.geometry(),
                                                                        model.rendering_network.geometry())
        return lay

    def outputs(self, n: int, s_t: int, dev):
        """The step's outputs: one allocation, sliced (ray_dirs, z_vals, points, normals, colors, weights, rgb, depth) + (out_terms, out_norm)."""
        m = n * s_t
        sizes = (n * 3, n * s_t, m * 3, m * 3, m * 3, n * s_t, n * 3, n)
        flat = torch.empty(sum(sizes), device=dev)
        views, o = [], 0
        for k in sizes:
            views.append(flat[o:o + k])
            o += k
        # (the step's scalars — loss terms, clip norm, counts — in a little tensor of their own: whoever keeps a step's terms, e.g. the
        #  trainer's running sums, keeps 48 bytes alive and not the step's outputs)
        small = torch.empty(12, device=dev)
        views += [small[:8], small[8:]]
        io = self.io
        io.ray_dirs, io.z_vals, io.points, io.normals, io.colors, io.weights, io.rgb, io.depth = (_p(t) for t in views[:8])
        io.out_terms, io.out_norm = _p(views[8]), _p(views[9])
        io.out_counts = views[9].data_ptr() + 8           # [2:4] of the same little tensor: samples the colour branch ran on, all samples
        return views

    def _anchors(self):
        """One parameter of each net and the density's scalars: what makes a step's outputs differentiable (their gradients are written in
        place by the C call; the module tree is walked once)."""
        a = getattr(self, "_anchor_cache", None)
        if a is None or any(p.device != a[1] for p in a[0][:1]):
            model = self.model
            params = [next(iter(model.vector_field_network.parameters())), next(iter(model.rendering_network.parameters()))] + list(model.density.parameters())
            a = self._anchor_cache = (params, params[0].device)
        return a[0]

    def supersede(self) -> None:
        """The structs are about to describe another step: a session still open on them can no longer run its backward."""
        s = self.session
        if s is not None and s.open:
            s.stale = True
        self.session = None

    # ---------------------------------------------------------------------------------------------
    def open_session(self, pose, pixels, intrinsics, epoch: int, uniforms) -> Optional[NerfOutput]:
        """A grad-mode render() in the shipped regime: the render part of the step from C, outputs tied to one autograd node.  None (with
        ``why_not``) when the regime or the batch does not allow it — the caller then takes the launch-by-launch path."""
        global _current
        model = self.model
        cfg = model.config
        if not getattr(model, "step_sessions", True):
            self.why_not = "model.step_sessions is off"
            return None
        prev = self.session
        if prev is not None and prev.open and not prev.abandoned():
            self.why_not = "an earlier render()'s step is still waiting for its backward"
            return None
        n = pixels.shape[0]
        s_c = model.ray_sampler.N_samples
        n_f = min(model.fine_sampler.N_samples, model.fine_sampler.max_samples)
        m = n * (s_c + n_f)
        # the trainer appends (N S_t) // 10 points per supervision batch (train/vector_field_nerf_train.py:186-214): room for the configured ones
        batches = int(bool(cfg.border_supervision)) + int(bool(cfg.center_supervision))
        sup_rows = batches * _round32(m // 10)
        reason = self.model_reason(pose, False, n, sup_rows)
        if reason is not None:
            self.why_not = reason
            return None
        self.why_not = None
        dev = pose.device
        self.supersede()
        f = self.bind(dev)
        pr, io = self.params, self.io
        n, s_c, n_f, keep_alive = self.fill_render(pose, pixels, intrinsics, epoch, uniforms, int(getattr(model, "train_step_streams", 2)))
        s_t = s_c + n_f
        pr.n_sup, pr.border, pr.center = 0, 0, 0
        pr.sup_rows_reserved = sup_rows
        for i in range(3):
            pr.sup_centroid[i] = 0.0
        self.fill_storage()
        key = (n, s_c, n_f, ("session", sup_rows), pr.sparse_colours, str(dev))
        ws = self.workspace(key, dev)
        lay = self.layout(key)
        views = self.outputs(n, s_t, dev)
        io.workspace = _p(ws)
        io.rgb_gt = io.depth_gt = io.sup_u_border = io.sup_u_center = None
        io.d_rgb_in = io.d_depth_in = io.d_normals_in = None
        session = StepSession(self, n, s_c, n_f, sup_rows, ws, lay, views, f)
        rgb, depth, normals = _SessionRender.apply(session, *self._anchors())
        del keep_alive
        self.session = session
        _current = weakref.ref(session)
        model.vector_field_network._step_ws = None      # (later vector-field forwards that do not join the session stand alone)
        ray_dirs, z, pts, _, colors, weights = views[:6]
        colours_out = colors.view(n * s_t, 3)
        if pr.sparse_colours:
            # the reference returns every sample's colour (vector_field_nerf.py:338); the step evaluated the selected ones.  The others are
            # filled on first access (render_output.LazyColours) by the dense gradient-free launch — with THIS step's weights, so before
            # optimizer.step(); model.eager_session_colours = True fills them here (one more fused forward per step)
            colours_out = LazyColours(colours_out, model._dense_colours_fill(pts, ray_dirs, n, s_t, int(pr.forward_products)))
            if getattr(model, "eager_session_colours", False):
                colours_out = colours_out.materialise()
        return NerfOutput(points_coarse=pts.view(n, s_t, 3), points_fine=None, coarse_normals=normals.view(n, s_t, 3),
                          coarse_rgb_values=rgb.view(n, 3), coarse_depth_map=depth.view(n, 1), fine_normals=None, fine_rgb_values=None,
                          fine_depth_map=None, z_vals=z.view(n, s_t), directional_derivtives=None, ray_dirs=RepeatedRows(ray_dirs.view(n, 3), s_t),
                          coarse_colors=colours_out)


class StepSession:
    """One training step between a grad-mode render() and its backward (see the module docstring)."""

    def __init__(self, engine: StepEngine, n: int, s_c: int, n_f: int, sup_rows: int, ws: torch.Tensor, lay: List[int], views, flat) -> None:
        # (the engine holds its last session; the session reaches the engine — and through it the model — weakly: no cycle keeps a step's
        #  30 GB workspace waiting for the cyclic collector once the model is gone)
        self._engine_ref = weakref.ref(engine)
        self.n, self.s_c, self.n_f, self.s_t, self.m = n, s_c, n_f, s_c + n_f, n * (s_c + n_f)
        self.sup_rows = sup_rows
        self.ws, self.flat = ws, flat
        self.views = views
        self.ray_dirs, self.z, self.points, self.normals, self.colors, self.weights, self.rgb, self.depth, self.out_terms, self.out_norm = views
        self.normals = self.normals.view(self.m, 3)

        # (views of the persistent workspace: the same five tensors step after step, made once per workspace)
        cached = getattr(engine, "_region_views", None)
        if cached is None or cached[0] != (ws.data_ptr(), sup_rows, self.m):
            def rows3(index: int, rows: int) -> torch.Tensor:
                off = lay[index]
                return ws[off:off + rows * 12].view(torch.float32).view(rows, 3)

            cached = engine._region_views = ((ws.data_ptr(), sup_rows, self.m),
                                             (rows3(lib.TWS_SUP_PTS, sup_rows), rows3(lib.TWS_SUP_GT, sup_rows), rows3(lib.TWS_SUP_PRED, sup_rows),
                                              rows3(lib.TWS_D_SUP, sup_rows), rows3(lib.TWS_DN, self.m)))
        self.sup_pts, self.sup_gt, self.sup_pred, self.d_sup, self.dn = cached[1]
        self.next_row = 0
        self.regions: Dict[int, dict] = {}          # row0 -> {count, on_side, forwarded, pending}
        self.ray_centre: Optional[dict] = None      # the deferred centre-ball rows of functions.get_center_indices_and_gt
        self.open, self.stale, self.backward_done = True, False, False
        self.node_ref = None
        self._flush_queued = False

    @property
    def engine(self) -> "StepEngine":
        eng = self._engine_ref()
        if eng is None:
            raise RuntimeError("the model this training step belongs to no longer exists")
        return eng

    @property
    def model(self):
        return self.engine.model

    # -- life cycle -------------------------------------------------------------------------------
    def abandoned(self) -> bool:
        """Nothing can reach this step's backward any more (its outputs, hence its autograd node, are gone)."""
        return self.node_ref is None or self.node_ref() is None

    def _require_live(self, what: str) -> None:
        if self.stale:
            raise RuntimeError(f"{what}: a later render() or training step has reused this step's workspace (one open step per model; "
                               "set model.step_sessions = False to differentiate several renders in one backward)")

    # -- forward side -----------------------------------------------------------------------------
    def render(self) -> None:
        # (VectorFieldNerf.render holds the range guard's watch around this call; the supervision forwards that follow report into the same
        # status word through the struct's status_word field, and the next guarded call's read-back sees what they left there)
        eng = self.engine
        eng.params.phases = lib.TRAIN_RENDER
        lib.train_step(eng.params, eng.io)

    def take(self, count: int) -> Optional[int]:
        """First row of a fresh supervision region of ``count`` points (whole groups of 32 are set aside), or None when there is no room."""
        if not self.open or self.stale or self.backward_done or count <= 0:
            return None
        rows = _round32(count)
        if self.next_row + rows > self.sup_rows:
            return None
        row0 = self.next_row
        self.next_row += rows
        self.regions[row0] = dict(count=count, on_side=False, forwarded=False, pending=False)
        return row0

    def sample(self, inward: bool, r_min: float, r_max: float, centroid, count: int, u: Optional[torch.Tensor], seed: int, offset: int):
        """Supervision points and their ground truth in a fresh region -> (points[count,3], gt[count,3]) views of the workspace, or None."""
        dev = self.ws.device
        host = host_centroid(centroid)
        cdev = None
        if host is None:
            if not isinstance(centroid, torch.Tensor):
                return None
            cdev = centroid.to(dev).float().reshape(3).contiguous()
        if u is not None and (tuple(u.shape) != (count, 3) or u.device != dev or u.dtype != torch.float32 or not u.is_contiguous()):
            return None
        row0 = self.take(count)
        if row0 is None:
            return None
        eng = self.engine
        cx, cy, cz = host if host is not None else (0.0, 0.0, 0.0)
        self.regions[row0]["on_side"] = lib.train_step_supervision_points(eng.params, eng.io, inward, r_min, r_max, cx, cy, cz, cdev, row0, count, u,
                                                                          seed, offset)
        return self.sup_pts[row0:row0 + count], self.sup_gt[row0:row0 + count]

    def region_of(self, points: torch.Tensor) -> Optional[int]:
        """row0 of the region ``points`` is (exactly the view ``sample`` handed out, not yet forwarded), or None."""
        if not self.open or self.stale or self.backward_done or not points.is_cuda or points.dim() != 2 or points.shape[1] != 3 or \
                points.dtype != torch.float32 or not points.is_contiguous():
            return None
        delta = points.data_ptr() - self.sup_pts.data_ptr()
        if delta < 0 or delta % (32 * 12):
            return None
        row0 = delta // 12
        reg = self.regions.get(row0)
        if reg is None or reg["count"] != points.shape[0] or reg["forwarded"]:
            return None
        return row0

    def forward_rows(self, row0: int) -> torch.Tensor:
        reg = self.regions[row0]
        eng = self.engine
        lib.train_step_supervision_forward(eng.params, eng.io, row0, reg["count"], reg["on_side"])
        reg["forwarded"] = True
        return self.sup_pred[row0:row0 + reg["count"]]

    # -- backward side ----------------------------------------------------------------------------
    def supervision_gradient(self, row0: int, d_vec: Optional[torch.Tensor]) -> None:
        """The upstream gradient of a region's predictions.  Before the render node's backward: parked in the workspace's D_SUP rows, which
        that backward's ONE chain walks together with the render's samples; should the backward pass end without reaching that node (a
        loss on the supervision predictions alone), ``flush`` differentiates the parked rows on their own.  After it: right away."""
        self._require_live("backward of a supervision forward")
        if d_vec is None:
            return
        reg = self.regions[row0]
        self.d_sup[row0:row0 + reg["count"]].copy_(d_vec.reshape(reg["count"], 3))
        reg["pending"] = True
        if self.backward_done:
            self.flush()
        elif not self._flush_queued:
            self._flush_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self) -> None:
        self._flush_queued = False
        if self.stale:
            return
        eng = self.engine
        for row0, reg in sorted(self.regions.items()):
            if reg["pending"]:
                reg["pending"] = False
                lib.train_step_supervision_backward(eng.params, eng.io, row0, reg["count"])

    def run_backward(self, d_rgb, d_depth, d_normals) -> None:
        self._require_live("backward of render()")
        if self.backward_done:
            raise RuntimeError("backward of render(): this step has been differentiated already (retain_graph is not supported by the C step)")
        rc = self.ray_centre
        if rc is not None and not rc["consumed"]:
            raise RuntimeError("functions.get_center_indices_and_gt deferred the centre-ball rows to vf_nerf_amd's VFLoss, but the loss that ran did not "
                               "take them: install the drop-in loss (vf_nerf_amd.dropin) or set model.defer_center_rows = False")
        eng = self.engine
        io = eng.io
        dev = self.ws.device
        n, m = self.n, self.m

        def grad(t, shape, name):
            if t is None:
                return None
            t = t.reshape(shape)
            if t.dtype != torch.float32 or not t.is_contiguous():
                t = t.float().contiguous()
            if t.device != dev:
                raise lib.VfnError(f"{name}: upstream gradient on {t.device}, the step lives on {dev}")
            return t

        d_rgb = grad(d_rgb, (n, 3), "d rgb")
        if d_rgb is None:
            d_rgb = torch.zeros(n, 3, device=dev)
        d_depth = grad(d_depth, (n,), "d depth")
        d_normals = grad(d_normals, (m, 3), "d normals")
        if d_normals is None:
            self.dn.zero_()
            d_normals = self.dn
        io.d_rgb_in, io.d_depth_in, io.d_normals_in = _p(d_rgb), _p(d_depth), _p(d_normals)
        # the chain walks the supervision rows that forwards have filled, up to the last such region; a region below it that was sampled
        # but never forwarded (its rows hold no activations) is forwarded now — its upstream gradient is zero, it adds exactly nothing
        used = 0
        for row0, reg in sorted(self.regions.items(), reverse=True):
            if reg["forwarded"]:
                used = max(used, row0 + _round32(reg["count"]))
            elif row0 < used:
                self.forward_rows(row0)
        eng.params.sup_rows_used = used
        eng.params.phases = lib.TRAIN_BACKWARD
        lib.train_step(eng.params, eng.io)
        io.d_rgb_in = io.d_depth_in = io.d_normals_in = None
        for reg in self.regions.values():          # the one chain of the call above has walked every parked row
            reg["pending"] = False
        self.backward_done = True
        self.open = False
        self.model._last_colour_counts = self.out_norm[2:4]
        from . import optim
        if optim.CLIP_INSIDE_STEP:
            # torch.nn.utils.clip_grad_norm_ is PyTorch's own (dropin.install(patch_clip=False)): the gradient stays parked in the flat
            # buffer, invisible to it; optimizer.step() all-reduces, clips with the configured norm and updates (optim.FlatAdam.step)
            for p, _, _, _ in self.flat["entries"]:
                p.grad = None
            self.flat["parked_max_norm"] = float(self.model.config.scheduler_config.clip_norm)


class _SessionRender(torch.autograd.Function):
    """render() of an open step: forward = VFN_TRAIN_RENDER, backward = VFN_TRAIN_BACKWARD (every parameter gradient is ADDED in place into
    the optimizer's flat gradient: the inputs are anchors that make the outputs differentiable, they receive no gradient of their own)."""

    @staticmethod
    def forward(ctx, session: StepSession, *anchors):
        session.render()
        ctx.session = session
        ctx.n_anchors = len(anchors)
        ctx.set_materialize_grads(False)
        session.node_ref = weakref.ref(ctx)
        return session.rgb.view(session.n, 3), session.depth.view(session.n, 1), session.normals

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_normals):
        ctx.session.run_backward(d_rgb, d_depth, d_normals)
        return (None,) * (1 + ctx.n_anchors)


class _SessionVF(torch.autograd.Function):
    """vector_field_network(points)[:, :3] on a supervision region of an open step."""

    @staticmethod
    def forward(ctx, session: StepSession, row0: int, anchor):
        ctx.session, ctx.row0 = session, row0
        ctx.set_materialize_grads(False)
        return session.forward_rows(row0)

    @staticmethod
    def backward(ctx, d_vec):
        ctx.session.supervision_gradient(ctx.row0, d_vec)
        return None, None, None


class _CentreRowsMarker(torch.autograd.Function):
    """What functions.get_center_indices_and_gt returns for an open step: NO rows — the selection (ray samples inside the centre ball), its
    count and its gradient live inside the fused loss kernels (csrc/vfn_loss.hip, ``ray_center``), which ``loss.VFLoss`` switches on when it
    finds this node behind ``pred["supervised_normals"]``.  The reference's boolean-mask indexing (functions.py:137-157) is a device
    synchronisation in the middle of every step."""

    @staticmethod
    def forward(ctx, normals, session: StepSession):
        ctx.session = session
        return normals.new_empty(0, 3)

    @staticmethod
    def backward(ctx, _g):
        return None, None


def centre_rows(session: StepSession, points, normals, centroid, radius: float):
    """-> (prediction rows [0,3] carrying the marker node, ground truth rows [0,3]) or None when the rows cannot be deferred."""
    host = host_centroid(centroid)
    if host is None or session.ray_centre is not None or not normals.requires_grad:
        return None
    if normals.data_ptr() != session.normals.data_ptr() or points.data_ptr() != session.points.data_ptr() or normals.numel() != session.m * 3:
        return None
    session.ray_centre = dict(centroid=host, radius=float(radius), consumed=False)
    return _CentreRowsMarker.apply(normals, session), normals.new_empty(0, 3)


def find_marker(t: Optional[torch.Tensor]) -> Optional[StepSession]:
    """The session whose centre-ball marker sits behind ``t`` in the autograd graph (directly or through torch.cat / views), or None."""
    fn = getattr(t, "grad_fn", None)
    seen, stack = 0, [fn]
    while stack and seen < 64:
        node = stack.pop()
        if node is None:
            continue
        seen += 1
        if type(node).__name__ == "_CentreRowsMarkerBackward":
            return node.session
        name = type(node).__name__
        if name.startswith(("CatBackward", "ViewBackward", "ReshapeAliasBackward", "SliceBackward", "AliasBackward", "UnsafeViewBackward")):
            stack.extend(nf for nf, _ in node.next_functions)
    return None


# ------------------------------------------------------------------------------------------------
# vector_field_network(points) -> [n, 3 + F] whose [:, :3] is a supervision region's vector head
# ------------------------------------------------------------------------------------------------
class LazyVFOutput(torch.Tensor):
    """The [n, 3 + F] result of ``vector_field_network(points)`` on a supervision region of an open step.  The trainer keeps the three vector
    columns (``...[:, :3]``, train/vector_field_nerf_train.py:201,213): that slice is the region's vector-only saving forward, and nothing
    else is evaluated.  ANY other use materialises the full matrix through the stand-alone differentiable forward (its own workspace, the
    feature block included) and proceeds on that — same values, same gradients, one more forward."""

    @staticmethod
    def __new__(cls, vec: torch.Tensor, make_full, cols: int):
        t = torch.Tensor._make_wrapper_subclass(cls, (vec.shape[0], cols), dtype=vec.dtype, device=vec.device, requires_grad=vec.requires_grad)
        t._vec, t._make_full, t._full = vec, make_full, None
        return t

    def materialise(self) -> torch.Tensor:
        if self._full is None:
            self._full = self._make_full()
            self._make_full = None
        return self._full

    @staticmethod
    def _is_vector_columns(key) -> bool:
        if not (isinstance(key, tuple) and len(key) == 2):
            return False
        rows, cols = key
        if not (isinstance(rows, slice) and rows == slice(None)) and rows is not Ellipsis:
            return False
        return isinstance(cols, slice) and cols.start in (None, 0) and cols.stop == 3 and cols.step in (None, 1)

    @classmethod
    def _swap(cls, a):
        if isinstance(a, LazyVFOutput):
            return a.materialise()
        if isinstance(a, (list, tuple)):
            return type(a)(cls._swap(x) for x in a)
        return a

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is torch.Tensor.__getitem__ and isinstance(args[0], LazyVFOutput) and cls._is_vector_columns(args[1]):
            return args[0]._vec
        with torch._C.DisableTorchFunctionSubclass():
            if func in _METADATA:
                return func(*args, **kwargs)
            return func(*cls._swap(args), **{k: cls._swap(v) for k, v in kwargs.items()})

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        return func(*cls._swap(args), **{k: cls._swap(v) for k, v in kwargs.items()})


_T = torch.Tensor
_METADATA = {_T.shape.__get__, _T.size, _T.dim, _T.ndim.__get__, _T.device.__get__, _T.dtype.__get__, _T.numel, _T.nelement, _T.__len__,
             _T.requires_grad.__get__, _T.is_cuda.__get__, _T.ndimension}


def session_vf_forward(net, points: torch.Tensor):
    """``vector_field_network(points)`` when ``points`` is a supervision region of the open step of ``net``'s model -> LazyVFOutput, else None."""
    session = current_session()
    if session is None or session.model.vector_field_network is not net or not torch.is_grad_enabled():
        return None
    row0 = session.region_of(points)
    if row0 is None:
        return None
    vec = _SessionVF.apply(session, row0, session.engine._anchors()[0])
    pts = points

    def make_full():
        from .backward import vf_forward_autograd
        return vf_forward_autograd(net, pts, False)

    return LazyVFOutput(vec, make_full, 3 + net._feature_dims())
