"""Autograd wrappers of the training path.

The reference trains through plain PyTorch autograd over the fine pass (models/nerf/vector_field_nerf.py:294-323;
train/vector_field_nerf_train.py:251-260) with BatchNorm in eval mode (Q8).  Here the same gradients come from HIP
kernels: a training forward that saves each layer's output, a fused dX chain, persistent weight-gradient kernels
(csrc/vfn_mlp_bwd.hip) and the per-ray density/composite backward (csrc/vfn_rays.hip).  What remains in PyTorch is
parameter-sized glue: summing the partial slabs and un-folding BatchNorm / the skip scale back onto the reference's
parameters (Linear weight/bias, BatchNorm weight/bias).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from . import lib

HID = lib.HIDDEN
EPS_BN = 1e-5


# ------------------------------------------------------------------------------------------------
# host-side mirror of the kernel's layer plan (csrc/vfn_plan.h): which reference rows / columns each
# hidden entry covers
# ------------------------------------------------------------------------------------------------
def _entries(net) -> List[dict]:
    geom = net.geometry()
    L = geom.n_layers
    pe = 3 + 6 * geom.multires if geom.multires > 0 else 3
    F = geom.feature_dims
    out = []
    if net._kind == lib.NET_VF:
        skip = geom.skip_layer
        for i in range(L - 1):
            e = dict(layer=i, rows=geom.out_dims[i], row_off=0, scale=1.0, act=None, aux=None)
            if i == 0:
                e["aux"] = (0, pe)
            elif i == skip:
                e["act"] = (0, geom.out_dims[i - 1])
                e["aux"] = (geom.out_dims[i - 1], pe)
                e["scale"] = 1.0 / math.sqrt(2.0)
            else:
                e["act"] = (0, HID)
            out.append(e)
        if F > 0:
            out.append(dict(layer=L - 1, rows=F, row_off=3, scale=1.0, act=(0, HID), aux=None))
    else:
        for i in range(L - 1):
            e = dict(layer=i, rows=geom.out_dims[i], row_off=0, scale=1.0, act=(0, HID), aux=None)
            if i == 0:
                e["act"] = (6 + pe, F)
                e["aux"] = (0, 6 + pe)
            out.append(e)
    return out


def _packed_bwd(net) -> torch.Tensor:
    tensors, key = net._pack_key()
    dev = tensors[0].device
    cache = getattr(net, "_packed_bwd_cache", None)
    if cache is None or cache[0] != key or cache[1].device != dev:
        geom = net.geometry()
        buf = torch.empty(max(1, lib.packed_bwd_size(net._kind, geom)), device=dev)
        with torch.no_grad():
            lib.pack_weights_bwd(net._kind, geom, [{k: v.detach() for k, v in d.items()} for d in net._layer_tensors()],
                                 buf)
        net._packed_bwd_cache = (key, buf)
        cache = net._packed_bwd_cache
    return cache[1]


def _packed_bwd16(net, rounded: bool = False) -> torch.Tensor:
    """Transposed bf16-split pack for the bf16 dX chain (csrc/vfn_bwd16.hip), cached on the parameter versions.  ``rounded``: the
    pack of the single-product chain (hi planes rounded to nearest; its own cache)."""
    tensors, key = net._pack_key()
    dev = tensors[0].device
    name = "_packed_bwd16r_cache" if rounded else "_packed_bwd16_cache"
    cache = getattr(net, name, None)
    if cache is None or cache[0] != key or cache[1].device != dev:
        geom = net.geometry()
        buf = torch.empty(lib.packed_bwd16_size(net._kind, geom), dtype=torch.uint8, device=dev)
        with torch.no_grad():
            lib.pack_weights_bwd16(net._kind, geom, [{k: v.detach() for k, v in d.items()} for d in net._layer_tensors()], buf,
                                   round_hi=rounded)
        cache = (key, buf)
        setattr(net, name, cache)
    return cache[1]


def _head_rows(net) -> torch.Tensor:
    """Rows 0..2 of the last Linear ([3][256], contiguous view): the 3-channel head."""
    return net._linear(net.num_layers - 1).weight.detach()[:3]


def _groups(m: int) -> int:
    return max(1, min(256, m // 256))


def _vf_inputs(vf, saved_slots: List[torch.Tensor]) -> List[Optional[torch.Tensor]]:
    """act input of each VF entry (entry 0 reads only the encoding; entry h reads the output of entry h-1; the feature
    block reads the last plain hidden output) followed by the head's input."""
    entries = _entries(vf)
    n_plain = len(entries) - (1 if vf._feature_dims() > 0 else 0)
    inputs: List[Optional[torch.Tensor]] = [None] + [saved_slots[h - 1] for h in range(1, n_plain)]
    if vf._feature_dims() > 0:
        inputs.append(saved_slots[n_plain - 1])
    inputs.append(saved_slots[n_plain - 1])
    return inputs


def _rn_inputs(feats: torch.Tensor, saved_slots: List[torch.Tensor]) -> List[torch.Tensor]:
    """rendering net: entry 0 reads the features, entry h the output of entry h-1, the head the last output."""
    return [feats] + list(saved_slots)


def _layer_table(net):
    """vfn_wgrad_layer array of the net (parameters and their .grad tensors per reference layer), rebuilt when an address moves."""
    rows = []
    for i in range(net.num_layers):
        lin, bn = net._linear(i), net._bn(i)
        row = [lin.weight, lin.bias, None, None, None, lin.weight.grad, lin.bias.grad, None, None]
        if bn is not None:
            row[2:5] = [bn.weight, bn.running_var, bn.running_mean]
            row[7:9] = [bn.weight.grad, bn.bias.grad]
        rows.append(row)
    key = tuple(0 if t is None else t.data_ptr() for row in rows for t in row)
    cache = getattr(net, "_wgrad_table", None)
    if cache is None or cache[0] != key:
        arr = (lib.WgradLayer * len(rows))()
        k = 0
        for i in range(len(rows)):
            for name, _ in lib.WgradLayer._fields_:
                setattr(arr[i], name, key[k] or None)
                k += 1
        cache = net._wgrad_table = (key, arr)
    return cache[1]


def _wgrad_scratch(net, m: int, dev) -> torch.Tensor:
    cache = getattr(net, "_wgrad_scratch_cache", None)
    if cache is None:
        cache = net._wgrad_scratch_cache = {}
    key = (m, str(dev))
    buf = cache.get(key)
    if buf is None:
        if len(cache) >= 4:
            cache.clear()
        buf = cache[key] = torch.empty(lib.net_weight_grads_scratch_bytes(net._kind, net.geometry(), m), dtype=torch.uint8, device=dev)
    return buf


def _direct_ok(net, dev) -> bool:
    """Every parameter of the net has a gradient tensor the un-fold kernel can add into (the flat optimizer's views)."""
    if not getattr(net, "accumulate_into_grad", True):
        return False
    dev = torch.device(dev)
    plist = getattr(net, "_param_list_cache", None)
    if plist is None:
        plist = net._param_list_cache = list(net.parameters())
    # In-place mode writes every parameter's gradient (and hands autograd ONE anchor per net): only sound when every parameter
    # wants a gradient and nobody listens for it — a frozen parameter must not be touched, a tensor hook / post-accumulate hook
    # would never fire.  Otherwise: the full parameter list through autograd's own accumulation.
    for p in plist:
        if not p.requires_grad or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    # the flat optimizer's views (optim.FlatAdam) are contiguous fp32 slices of one device buffer by construction: when every
    # gradient IS the view it made, nothing else needs checking
    from .optim import owner_of
    owner = owner_of(plist[0]) if plist else None
    if owner is not None and owner._flat is not None and owner._flat["grad"].device == dev:
        views = owner._flat.get("grad_views")
        if views is not None:
            by_id = owner._flat.get("view_of")
            if by_id is None:
                by_id = owner._flat["view_of"] = {id(p): v for (p, _, _, _), v in zip(owner._flat["entries"], views)}
            if all(p.grad is by_id.get(id(p)) and p.grad is not None for p in plist):
                return True
    return all(p.grad is not None and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and p.grad.device == dev and not p.grad.requires_grad
               for p in plist)


def _ensure_grads(net) -> None:
    """A forward that chose in-place accumulation handed autograd ONE anchor parameter per net; should the gradients have been
    dropped since (zero_grad(set_to_none=True)), the sums start from fresh zeros here."""
    for p in net.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)


def _weight_grads(net, inputs, dy_slots, aux, dz_head, m: int, skip=(), fast: bool = False, x_f16: bool = False,
                  x_fp32_entries=(), frag=None, ws_ref=None) -> Dict[torch.nn.Parameter, torch.Tensor]:
    """inputs[h]: [M,256] act input of hidden entry h (None if it has none); inputs[-1]: input of the 3-channel head.
    dy_slots[h]: [M,256] pre-activation gradient of entry h.  aux: [M,40].  dz_head: [M,4].  Runs the persistent
    weight-gradient kernels (``fast``: the 256 x 256 products on the bf16 matrix cores, csrc/vfn_dw16.hip), then ONE
    launch (csrc/vfn_unfold.hip) sums their partial slabs and un-folds BatchNorm / the skip scale onto the parameters:
        W' = s * scale * W,  b' = s (b - mu) + beta_bn,  s = gamma / sqrt(var + eps).
    ``x_f16``: the act inputs are workspace slots in the f16 storage (first 512 bytes of each row), except the
    entries listed in ``x_fp32_entries`` (the rendering net's first layer reads the fp32 feature slot).
    ``frag`` = (dy_form, x_form): the workspace is FRAGMENT-ORDERED (csrc/vfn_dwf.hip: every product, thin ones included, on
    the bf16 matrix cores from 1-KiB-contiguous pieces); inputs / dy_slots are then flat slots, and the entries of
    ``x_fp32_entries`` are row-major fp32 [M,256] (the tanh'ed features)."""
    dev = aux.device
    entries = _entries(net)
    G = _groups(m)
    grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
    unfold = []
    # When every parameter of the net already HAS a gradient tensor (the flat optimizer keeps them as views of one buffer, zeroed
    # by zero_grad), the un-fold kernel adds straight into it and nothing is handed to autograd for these parameters: the ~90
    # AccumulateGrad additions of a training step (one launch each) disappear.  (Tensor hooks on the parameters do not fire then.)
    direct = _direct_ok(net, dev)

    if frag is not None and direct and ws_ref is not None and getattr(net, "one_call_weight_grads", True):
        # fragment-ordered workspace, gradients accumulated in place: the whole launch sequence below from C out of one cached
        # scratch buffer (csrc/vfn_wgrad.hip: same launches, same values; ~45 Python-side operations per backward fewer).
        # ws_ref = (saved [slots, slot_floats], dy [slots, slot_floats], this net's first slot)
        saved, dy_all, first = ws_ref
        lib.net_weight_grads_frag(net._kind, net.geometry(), _layer_table(net), saved, first, dy_all, saved.shape[1], frag[0], frag[1],
                                  inputs[0] if 0 in x_fp32_entries else None, aux, dz_head, m, with_features=not skip,
                                  accumulate=True, scratch=_wgrad_scratch(net, m, dev))
        return {}

    def out_for(p):
        if direct:
            return p.grad
        if p not in grads:
            grads[p] = torch.empty_like(p)
        return grads[p]

    for h, e in enumerate(entries):
        if h in skip:
            continue
        lin, bn = net._linear(e["layer"]), net._bn(e["layer"])
        dy = dy_slots[h]
        db_part = torch.empty(G, HID, device=dev)
        u = dict(db=db_part, w=lin.weight.detach(), b_lin=lin.bias.detach(), g_w=out_for(lin.weight), g_b=out_for(lin.bias),
                 rows=e["rows"], row_off=e["row_off"], in_dim=lin.in_features, slab_rows=HID, scale=e["scale"])
        if e["act"] is not None:
            part = torch.empty(G, HID, HID, device=dev)
            if frag is not None:
                lib.weight_grad_frag(0, dy, frag[0], inputs[h], lib.XF_ROWS32 if h in x_fp32_entries else frag[1], m, G, part, db_part)
            elif fast:
                lib.weight_grad_partials_bf16(dy, inputs[h], m, G, part, db_part, x_f16=x_f16 and h not in x_fp32_entries)
            else:
                lib.weight_grad_partials(0, dy, HID, HID, inputs[h], HID, HID, m, G, part, db_part)
            u.update(dw_act=part, act_c0=e["act"][0], act_nc=e["act"][1])
        if e["aux"] is not None:
            part = torch.empty(G, HID, 64, device=dev)
            if frag is not None:
                lib.weight_grad_frag(1, dy, frag[0], aux, lib.XF_AUX40, m, G, part, db_part if e["act"] is None else None)
            else:
                lib.weight_grad_partials(1, dy, HID, HID, aux, lib.AUX_K, lib.AUX_K, m, G, part,
                                         db_part if e["act"] is None else None)
            u.update(dw_aux=part, aux_c0=e["aux"][0], aux_nc=e["aux"][1])
        if bn is not None:
            u.update(bn_w=bn.weight.detach(), bn_var=bn.running_var.detach(), bn_mean=bn.running_mean.detach(),
                     g_bn_w=out_for(bn.weight), g_bn_b=out_for(bn.bias))
        unfold.append(u)
    # 3-channel head = rows 0..2 of the last Linear (no BatchNorm)
    last = net._linear(net.num_layers - 1)
    part = torch.empty(G, 32, HID, device=dev)
    dbp = torch.empty(G, 32, device=dev)
    if frag is not None:
        lib.weight_grad_frag(2, dz_head, lib.DYF_DZ4, inputs[-1], frag[1], m, G, part, dbp)
    else:
        lib.weight_grad_partials(2, dz_head, 4, 3, inputs[-1], HID, HID, m, G, part, dbp, x_f16=x_f16)
    fresh = not direct and last.weight not in grads          # the feature rows of the last Linear were skipped (vector-only forward)
    unfold.append(dict(dw_act=part, db=dbp, w=last.weight.detach(), b_lin=last.bias.detach(), g_w=out_for(last.weight),
                       g_b=out_for(last.bias), rows=3, row_off=0, in_dim=last.in_features, slab_rows=32, act_c0=0,
                       act_nc=HID, scale=1.0))
    if fresh and last.out_features > 3:
        grads[last.weight][3:].zero_()
        grads[last.bias][3:].zero_()
    lib.unfold_weight_grads(unfold, G, accumulate_mask=(1 << len(unfold)) - 1 if direct else 0)
    return grads


_F16_TRAIN_MAX_POINTS = 1 << 21   # the 16-bit training kernels address a fragment-ordered slot (32 KiB per 32 points) with 32-bit offsets


class _Workspace:
    """What a training forward leaves for the backward.

    16-bit path (``frag``): every slot is FRAGMENT-ORDERED (include/vfn.h) — flat buffers of ceil(M/32) groups x 32 KiB;
    ``f16``: the ReLU slots hold f16 values (``VectorFieldNerf.activation_storage``); slot 8 of a VF net, the tanh'ed features,
    is row-major fp32 [M,256] (``feats``).  ``dy16``: the chain stores the pre-activation gradients in 16 bits —
    "bf16", or "f16" = f16 of the values scaled per lane and tile with the exponents behind each group's pieces (csrc/vfn_dwf.hip,
    "dY form 3") (``VectorFieldNerf.gradient_storage``).  Exact-fp32 path: row-major [slots][M][256] fp32.
    ``dy16 = "f16p1"`` (``VectorFieldNerf.training_products = 1`` on the default storages): the same workspace as "f16", filled and
    walked by the SINGLE-PRODUCT forward and chain (``single``)."""

    def __init__(self, m: int, n_slots: int, dev, f16: bool = False, frag: bool = False, dy16=False) -> None:
        dy16 = {True: "bf16", False: None, None: None, "fp32": None}.get(dy16, dy16) if frag else None
        self.single = dy16 == "f16p1"
        if self.single:
            assert f16 and frag
            dy16 = "f16"
        assert dy16 in (None, "bf16", "f16"), dy16
        self.m, self.n_slots, self.f16, self.frag, self.dy16 = m, n_slots, f16, frag, dy16
        if frag:
            self.slot_floats = lib.frag_groups(m) * lib.GROUP_FLOATS
            self.saved = torch.empty(n_slots, self.slot_floats, device=dev)
        else:
            self.slot_floats = m * HID
            self.saved = torch.empty(n_slots, m, HID, device=dev)
        # sign bits of the saved ReLU outputs (f16x3 training forwards write them, the bf16 chain reads them): 32 B per point and slot
        self.masks = torch.empty(n_slots, m, 2, 4, dtype=torch.int32, device=dev)
        self.aux_vf = torch.empty(m, lib.AUX_K, device=dev)
        self.aux_rn = torch.empty(m, lib.AUX_K, device=dev)

    def slot(self, h: int) -> torch.Tensor:
        return self.saved[h]

    def feats(self, h: int = 8) -> torch.Tensor:
        """The tanh'ed feature slot as the row-major [M,256] fp32 matrix it always is."""
        return self.saved[h].reshape(-1)[: self.m * HID].view(self.m, HID)

    def fwd_flags(self) -> int:
        return (lib.WS_F16 if self.f16 else 0) | (lib.WS_FRAG if self.frag else 0) | (lib.WS_P1 if self.single else 0)

    def dy_flags(self) -> int:
        return (lib.DY_FRAG if self.frag else 0) | {None: 0, "bf16": lib.DY_BF16, "f16": lib.DY_F16S}[self.dy16] | (lib.DY_P1 if self.single else 0)

    def frag_forms(self):
        """(dy_form, x_form) of lib.weight_grad_frag for this workspace, or None for the row-major layouts."""
        if not self.frag:
            return None
        return ({None: lib.DYF_FRAG32, "bf16": lib.DYF_FRAGBF16, "f16": lib.DYF_FRAGF16S}[self.dy16], lib.XF_FRAG16 if self.f16 else lib.XF_FRAG32)

    def new_dy(self) -> torch.Tensor:
        dev = self.saved.device
        return torch.empty(self.n_slots, self.slot_floats, device=dev) if self.frag else torch.empty(self.n_slots, self.m, HID, device=dev)

    def rows(self, h: int, feature_slot: int = 8) -> torch.Tensor:
        """Slot h as a row-major fp32 [M,256] matrix whatever the storage (tests / debugging)."""
        if not self.frag:
            t = self.saved[h]
            return t.view(torch.float16)[:, :HID].float() if (self.f16 and h != feature_slot) else t
        if h == feature_slot:
            return self.feats(h)
        return lib.frag_to_rows(self.saved[h], self.m, torch.float16 if self.f16 else torch.float32)


def _train_products(model) -> int:
    """f16 products per fp32-equivalent product in the colour branch of an activation-saving forward (``training_colour_products``,
    default 3; 2 only while the inference setting is 2 as well — the guard's colour self-check speaks for both)."""
    return 2 if (getattr(model, "training_colour_products", 3) == 2 and getattr(model, "colour_products", 3) == 2) else 3


def _storage(owner, fast: bool):
    """(f16 activations, fragment order, 16-bit gradient form or None) for a 16-bit training forward of ``owner`` (model or VF net)."""
    if not fast:
        return False, False, None
    grads = getattr(owner, "gradient_storage", "fp32")
    f16, frag = getattr(owner, "activation_storage", "fp32") == "f16", getattr(owner, "workspace_layout", "fragment") == "fragment"
    if getattr(owner, "training_products", 3) == 1 and f16 and frag and grads == "f16":
        grads = "f16p1"        # single-product forward and chain on the default storages (_Workspace.single)
    return f16, frag, None if grads == "fp32" else grads


# ------------------------------------------------------------------------------------------------
# fine pass of render()
# ------------------------------------------------------------------------------------------------
class _FinePass(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, pts, z, ray_dirs, *params):
        vf, rn = model.vector_field_network, model.rendering_network
        n, s_t = z.shape
        m = n * s_t
        dev = pts.device
        vf_h, rn_h = len(_entries(vf)), len(_entries(rn))
        fast = model.uses_f16x3() and m < _F16_TRAIN_MAX_POINTS
        # ``backward_kernels = "fp32"`` (diagnostic): the f16x3 forward with a row-major fp32 workspace, which the exact-fp32
        # backward kernels read — the same forward under both backward arithmetics (tests/test_hip_fullsize.py)
        bwd_fast = fast and getattr(model, "backward_kernels", "auto") != "fp32"
        f16, frag, dy16 = _storage(model, bwd_fast)
        ws = _Workspace(m, vf_h + rn_h, dev, f16=f16, frag=frag, dy16=dy16)
        scal = model.density.raw_scalars()
        if fast:                                               # split-half products, fp32-equivalent (csrc/vfn_mlp16.hip)
            normals, colors = lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(),
                                                              rn.packed16_weights(), pts.reshape(-1, 3), ray_dirs, s_t,
                                                              ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags(),
                                                              colour_products=1 if ws.single else _train_products(model))
        else:
            normals, colors = lib.vf_render_fused_fwd_train(vf.geometry(), vf.packed_weights(), rn.geometry(),
                                                            rn.packed_weights(), pts.reshape(-1, 3), ray_dirs, s_t,
                                                            ws.saved, ws.aux_vf, ws.aux_rn)
        dp = model._density_params()
        _, weights, _, rgb, depth = lib.ray_density_weights(dp, normals, ray_dirs, z, scal, colors=colors,
                                                            want_sigma=False)
        if getattr(model, "_keep_saved", False):   # test hook: expose the saved activations
            model._debug_saved = ws.saved
            model._debug_ws = ws
            model._debug_masks = ws.masks if fast else None
        ctx.model, ctx.ws, ctx.dims, ctx.param_order = model, ws, (n, s_t, m, vf_h, rn_h), list(params)
        ctx.fast = bwd_fast        # the backward must run the kernels that match what THIS forward wrote (masks, f16 slots)
        ctx.save_for_backward(normals, colors, z, ray_dirs, scal)
        return normals, colors, rgb, depth, weights

    @staticmethod
    def backward(ctx, d_normals, d_colors_direct, d_rgb, d_depth, d_weights):
        model, ws = ctx.model, ctx.ws
        n, s_t, m, vf_h, rn_h = ctx.dims
        normals, colors, z, ray_dirs, scal = ctx.saved_tensors
        vf, rn = model.vector_field_network, model.rendering_network
        dev = normals.device

        def cont(t, shape):
            return None if t is None else t.reshape(shape).float().contiguous()

        # (1) per-ray backward: d rgb / d depth / d weights -> d colours, d normals (density path), d scalars
        dn = torch.zeros(m, 3, device=dev) if d_normals is None else d_normals.reshape(m, 3).float().clone()
        dc = torch.empty(m, 3, device=dev)
        dscal = torch.zeros(3, device=dev)
        lib.ray_density_weights_bwd(model._density_params(), normals, ray_dirs, z, scal, colors, cont(d_rgb, (n, 3)),
                                    cont(d_depth, (n,)), cont(d_weights, (n, s_t)), dn, dc, dscal)
        if d_colors_direct is not None:
            dc = dc + d_colors_direct.reshape(m, 3)
        # (2) dX chain through the rendering net, the feature hand-off and the VF net
        dy = ws.new_dy()
        dz_rgb = torch.empty(m, 4, device=dev)
        dz_vec = torch.empty(m, 4, device=dev)
        fast = ctx.fast
        if fast:
            lib.mlp_bwd_chain_bf16_ws(vf.geometry(), _packed_bwd16(vf, ws.single), _head_rows(vf), rn.geometry(), _packed_bwd16(rn, ws.single),
                                      _head_rows(rn), ws.feats(vf_h - 1), ws.masks, dy, ws.dy_flags(), dc, colors, dn, normals, None, 3, m,
                                      dz_rgb, dz_vec)
        else:
            lib.mlp_bwd_chain(vf.geometry(), vf.packed_weights(), _packed_bwd(vf), rn.geometry(), rn.packed_weights(),
                              _packed_bwd(rn), ws.saved, dy, dc, colors, dn, normals, None, 3, m, dz_rgb, dz_vec)
        # (3) weight gradients
        feats = ws.feats(vf_h - 1) if fast else ws.saved[vf_h - 1]
        g_vf = _weight_grads(vf, _vf_inputs(vf, [ws.saved[h] for h in range(vf_h)]), [dy[h] for h in range(vf_h)],
                             ws.aux_vf, dz_vec, m, fast=fast, x_f16=ws.f16, frag=ws.frag_forms(), ws_ref=(ws.saved, dy, 0))
        g_rn = _weight_grads(rn, _rn_inputs(feats, [ws.saved[vf_h + h] for h in range(rn_h)]),
                             [dy[vf_h + h] for h in range(rn_h)], ws.aux_rn, dz_rgb, m, fast=fast, x_f16=ws.f16,
                             x_fp32_entries=(0,), frag=ws.frag_forms(), ws_ref=(ws.saved, dy, vf_h))
        # density scalars in density.parameters() order
        by_name = {"beta": dscal[0], "mean": dscal[1], "scale": dscal[2]}
        g_den = {p: by_name[name].reshape(p.shape) for name, p in model.density.named_parameters()}
        ctx.ws = None
        out = []
        for p in ctx.param_order:
            out.append(g_vf.get(p, g_rn.get(p, g_den.get(p))))
        return (None, None, None, None, *out)


# ------------------------------------------------------------------------------------------------
# one workspace per training step, shared by the fine pass of render() and the vector-field forwards that follow it
# ------------------------------------------------------------------------------------------------
def _round32(n: int) -> int:
    return (n + 31) // 32 * 32


class StepWorkspace:
    """The trainer differentiates three vector-field evaluations per step: the fine pass of render() and the two supervision
    batches (train/vector_field_nerf_train.py:177,191,203,215).  Their weight gradients are sums over points of the same
    products, so the later forwards APPEND their points to the workspace the fine pass laid out (whole groups of 32 points;
    a ragged batch is padded with points whose upstream gradient is zero, which contribute exactly nothing), every backward
    runs its own dX chain over its region, and ONE sequence of weight-gradient launches walks everything that was
    differentiated — issued by a callback at the end of the backward pass — instead of one sequence (and one set of partial
    slabs, and one un-fold) per forward.  Needs the flat optimizer's gradient views (results are added in place)."""

    EXTRA = 0.3             # room for the supervision points, as a fraction of the fine pass's points (the trainer uses 0.2)

    def __init__(self, model, m_fine: int, n_slots: int, dev) -> None:
        self.model, self.m_fine = model, m_fine
        fine = _round32(m_fine)
        extra = _round32(int(fine * self.EXTRA)) if m_fine % 32 == 0 else 0
        if fine + extra >= _F16_TRAIN_MAX_POINTS:
            extra = max(0, (_F16_TRAIN_MAX_POINTS - 32 - fine) // 32 * 32)
        self.total = fine + extra
        f16, frag, dy16 = _storage(model, True)
        self.ws = _Workspace(self.total, n_slots, dev, f16=f16, frag=True, dy16=dy16)
        self.storage = (f16, dy16)
        self.next = fine
        self.dy: Optional[torch.Tensor] = None
        self.dz_vec: Optional[torch.Tensor] = None
        self.done: List[tuple] = []           # (first, padded count, has features) of the regions whose chain has run
        self.queued = False
        self.regions, self.finished = 1, 0    # regions handed out (the fine pass is the first) / regions whose weight gradients are done

    def take(self, count: int) -> Optional[int]:
        """First point of a fresh region of ``count`` (padded) points, or None when the workspace is full."""
        if self.ws is None or count % 32 or self.next + count > self.total:
            return None
        first = self.next
        self.next += count
        self.regions += 1
        return first

    def gradients(self):
        if self.dy is None:
            self.dy = self.ws.new_dy()
            self.dz_vec = torch.empty(self.total, 4, device=self.dy.device)
        return self.dy, self.dz_vec

    def mark_done(self, first: int, count: int, features: bool) -> None:
        self.done.append((first, count, features))
        if not self.queued:
            self.queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def flush(self) -> None:
        """End of the backward pass: the vector-field net's weight gradients over every region whose chain has run."""
        # (the flag and the list are reset FIRST: should a launch below raise, the pool's later backward passes queue a flush again
        # instead of silently leaving the vector-field net's weight gradients out)
        self.queued = False
        regions, self.done = sorted(self.done), []
        if not regions:
            return
        vf = self.model.vector_field_network
        ws, (dy, dz_vec) = self.ws, self.gradients()
        dev = dy.device
        forms = ws.frag_forms()
        for p in vf.parameters():             # (gradients dropped between forward and backward, e.g. zero_grad(set_to_none=True):
            if p.grad is None:                #  nothing was handed to autograd for these parameters, so the sums start here)
                p.grad = torch.zeros_like(p)
        table = _layer_table(vf)

        def launch(first, count, parts):
            lib.net_weight_grads_frag(vf._kind, vf.geometry(), table, ws.saved, 0, dy, ws.slot_floats, forms[0], forms[1], None, ws.aux_vf,
                                      dz_vec, count, True, True, _wgrad_scratch(vf, count, dev), first_point=first, parts=parts)

        # maximal runs of adjacent regions; inside a run, maximal sub-runs of regions whose forward included the feature block
        i = 0
        while i < len(regions):
            j = i
            while j + 1 < len(regions) and regions[j][0] + regions[j][1] == regions[j + 1][0]:
                j += 1
            first, count = regions[i][0], regions[j][0] + regions[j][1] - regions[i][0]
            if all(r[2] for r in regions[i:j + 1]):
                launch(first, count, lib.WGRAD_LAYERS | lib.WGRAD_FEATURES | lib.WGRAD_HEAD)
            else:
                launch(first, count, lib.WGRAD_LAYERS | lib.WGRAD_HEAD)
                k = i
                while k <= j:
                    if regions[k][2]:
                        e = k
                        while e + 1 <= j and regions[e + 1][2]:
                            e += 1
                        launch(regions[k][0], regions[e][0] + regions[e][1] - regions[k][0], lib.WGRAD_FEATURES)
                        k = e + 1
                    else:
                        k += 1
            i = j + 1
        # every region differentiated: the step's 20 GB go back to the allocator now, not when the next render replaces this object
        # (a second backward over the same graph is not supported by these functions anyway)
        self.finished += len(regions)
        if self.finished >= self.regions:
            self.ws = self.dy = self.dz_vec = None


# ------------------------------------------------------------------------------------------------
# fine pass of a training render() with ONE vector-field evaluation per distinct sample
# ------------------------------------------------------------------------------------------------
class StoredFinePass:
    """The reference's training render evaluates the VF net on every proposal sample twice: without gradients for the proposal
    weights (vector_field_nerf.py:252-277), then again among the S_c + N_f samples of the differentiated fine pass (:294-318).
    The rendering net is pointwise and a proposal sample's inputs are complete before the fine sampler runs, so here the
    activation-saving fused forward runs on the proposal samples FIRST (its normals feed the proposal weights), after the
    sampler on the N_f new samples, both filling one workspace in storage order [proposal samples | new samples]; the results
    are scattered to their sorted positions for the density / composite, and the backward gathers the per-sample gradients
    back to storage order, where the chain and the weight-gradient kernels walk the workspace once — their sums over points
    do not care about the order.  Same per-sample arithmetic as the fused launch over the sorted samples, so every forward
    value is bit-identical; one 64-sample vector-only launch less per step."""

    def __init__(self, model, n: int, s_c: int, n_f: int, dev) -> None:
        vf, rn = model.vector_field_network, model.rendering_network
        self.model, self.n, self.s_c, self.n_f = model, n, s_c, n_f
        self.m_c, self.m = n * s_c, n * (s_c + n_f)
        self.vf_h, self.rn_h = len(_entries(vf)), len(_entries(rn))
        f16, frag, dy16 = _storage(model, True)
        assert frag
        # the step's shared workspace (StepWorkspace) when the gradients can be added in place, else a private one
        self.pool = None
        if getattr(model, "shared_step_workspace", True) and _direct_ok(vf, dev) and _direct_ok(rn, dev):
            self.pool = StepWorkspace(model, self.m, self.vf_h + self.rn_h, dev)
            self.ws = self.pool.ws
        else:
            self.ws = _Workspace(self.m, self.vf_h + self.rn_h, dev, f16=f16, frag=True, dy16=dy16)
        vf._step_ws = self.pool                                   # (replaces the previous step's; None: later VF forwards stand alone)
        self.ws_points = self.ws.m
        self.normals_s = torch.empty(self.m, 3, device=dev)      # storage order
        self.colors_s = torch.empty(self.m, 3, device=dev)

    @staticmethod
    def applicable(model, n: int, s_c: int, n_f: int) -> bool:
        return bool(getattr(model, "reuse_proposal_training", True) and model.uses_f16x3() and model.workspace_layout == "fragment" and
                    getattr(model, "backward_kernels", "auto") != "fp32" and (n * s_c) % 32 == 0 and
                    0 < n * (s_c + n_f) < _F16_TRAIN_MAX_POINTS and not model.config.numerical_jacobian and
                    model.rendering_network.config.detach_normals)

    def _launch(self, pts, ray_dirs, per_ray: int, first: int, count: int) -> None:
        model, ws = self.model, self.ws
        vf, rn = model.vector_field_network, model.rendering_network
        lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts.reshape(-1, 3),
                                        ray_dirs, per_ray, ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags(),
                                        ws_first=first, ws_points=self.ws_points, normals=self.normals_s[first:first + count],
                                        colors=self.colors_s[first:first + count], colour_products=1 if ws.single else _train_products(model))

    def proposal(self, pts_c, ray_dirs) -> torch.Tensor:
        """Saving forward on the proposal samples (generation order); returns their normals [N*S_c, 3]."""
        self._launch(pts_c, ray_dirs, self.s_c, 0, self.m_c)
        return self.normals_s[: self.m_c]

    def finish(self, new_pts, dst, z, ray_dirs):
        """Saving forward on the new samples, then the differentiable tail: (normals, colors, rgb, depth, weights), sorted."""
        self._launch(new_pts, ray_dirs, self.n_f, self.m_c, self.m - self.m_c)
        self.dst = dst
        model = self.model
        vf, rn = model.vector_field_network, model.rendering_network
        if self.pool is not None:
            # gradients are added in place: autograd only needs an anchor per net to call the backward (90 fewer inputs to track)
            params = [next(iter(vf.parameters())), next(iter(rn.parameters()))] + list(model.density.parameters())
        else:
            params = list(vf.parameters()) + list(rn.parameters()) + list(model.density.parameters())
        return _StoredFinePassFn.apply(self, z, ray_dirs, *params)


class _StoredFinePassFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sp: "StoredFinePass", z, ray_dirs, *params):
        model = sp.model
        n, s_t, m = sp.n, sp.s_c + sp.n_f, sp.m
        dev = z.device
        normals = torch.empty(m, 3, device=dev)
        colors = torch.empty(m, 3, device=dev)
        lib.scatter_rows3(sp.normals_s, sp.colors_s, sp.dst, normals, colors)
        scal = model.density.raw_scalars()
        _, weights, _, rgb, depth = lib.ray_density_weights(model._density_params(), normals, ray_dirs, z, scal, colors=colors, want_sigma=False)
        if getattr(model, "_keep_saved", False):   # test hook
            model._debug_saved, model._debug_ws, model._debug_masks, model._debug_dst = sp.ws.saved, sp.ws, sp.ws.masks, sp.dst
        ctx.sp, ctx.param_order = sp, list(params)
        ctx.save_for_backward(normals, colors, z, ray_dirs, scal)
        return normals, colors, rgb, depth, weights

    @staticmethod
    def backward(ctx, d_normals, d_colors_direct, d_rgb, d_depth, d_weights):
        sp = ctx.sp
        model, ws = sp.model, sp.ws
        n, s_t, m, vf_h, rn_h = sp.n, sp.s_c + sp.n_f, sp.m, sp.vf_h, sp.rn_h
        normals, colors, z, ray_dirs, scal = ctx.saved_tensors
        vf, rn = model.vector_field_network, model.rendering_network
        dev = normals.device

        def cont(t, shape):
            return None if t is None else t.reshape(shape).float().contiguous()

        # (1) per-ray backward on the SORTED samples
        dn = torch.zeros(m, 3, device=dev) if d_normals is None else d_normals.reshape(m, 3).float().clone()
        dc = torch.empty(m, 3, device=dev)
        dscal = torch.zeros(3, device=dev)
        lib.ray_density_weights_bwd(model._density_params(), normals, ray_dirs, z, scal, colors, cont(d_rgb, (n, 3)),
                                    cont(d_depth, (n,)), cont(d_weights, (n, s_t)), dn, dc, dscal)
        if d_colors_direct is not None:
            dc = dc + d_colors_direct.reshape(m, 3)
        # ... gathered to storage order: row r of the workspace is sorted sample dst[r]
        idx = sp.dst.long()
        dn_s, dc_s = dn.index_select(0, idx), dc.index_select(0, idx)
        # (2) dX chain over the fine pass's region of the workspace, (3) weight gradients
        pool = sp.pool
        dz_rgb = torch.empty(m, 4, device=dev)
        if pool is not None:
            dy, dz_vec = pool.gradients()
        else:
            dy, dz_vec = ws.new_dy(), torch.empty(m, 4, device=dev)
        lib.mlp_bwd_chain_bf16_ws(vf.geometry(), _packed_bwd16(vf, ws.single), _head_rows(vf), rn.geometry(), _packed_bwd16(rn, ws.single), _head_rows(rn),
                                  ws.feats(vf_h - 1), ws.masks, dy, ws.dy_flags(), dc_s, sp.colors_s, dn_s, sp.normals_s, None, 3, m, dz_rgb, dz_vec,
                                  ws_first=0, ws_points=sp.ws_points)
        feats = ws.feats(vf_h - 1)
        if pool is not None:
            _ensure_grads(rn)
        g_rn = _weight_grads(rn, _rn_inputs(feats, [ws.saved[vf_h + h] for h in range(rn_h)]),
                             [dy[vf_h + h] for h in range(rn_h)], ws.aux_rn, dz_rgb, m, fast=True, x_f16=ws.f16,
                             x_fp32_entries=(0,), frag=ws.frag_forms(), ws_ref=(ws.saved, dy, vf_h))
        if pool is not None:
            pool.mark_done(0, m, True)          # the vector-field net's products wait for the other regions of this backward pass
                                                # (a ragged m stays its own run: the weight-gradient kernel masks its last group's tail)
            g_vf = {}
        else:
            g_vf = _weight_grads(vf, _vf_inputs(vf, [ws.saved[h] for h in range(vf_h)]), [dy[h] for h in range(vf_h)],
                                 ws.aux_vf, dz_vec, m, fast=True, x_f16=ws.f16, frag=ws.frag_forms(), ws_ref=(ws.saved, dy, 0))
        by_name = {"beta": dscal[0], "mean": dscal[1], "scale": dscal[2]}
        g_den = {p: by_name[name].reshape(p.shape) for name, p in model.density.named_parameters()}
        sp.ws = None
        return (None, None, None, *[g_vf.get(p, g_rn.get(p, g_den.get(p))) for p in ctx.param_order])


def fine_pass_autograd(model, pts, z, ray_dirs):
    params = list(model.vector_field_network.parameters()) + list(model.rendering_network.parameters()) + \
        list(model.density.parameters())
    return _FinePass.apply(model, pts, z, ray_dirs, *params)


# ------------------------------------------------------------------------------------------------
# standalone VF forward (supervision points: train/vector_field_nerf_train.py:191,203,215)
# ------------------------------------------------------------------------------------------------
class _VFForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, points, vector_only, *params):
        pts = points.reshape(-1, 3).float().contiguous()
        m = pts.shape[0]
        dev = pts.device
        has_feat = net._feature_dims() > 0
        vf_h = len(_entries(net))
        cols = 3 if (vector_only or not has_feat) else 3 + net._feature_dims()
        fast = getattr(net, "precision", "fp32") == "f16x3" and net.supports_f16x3() and m < _F16_TRAIN_MAX_POINTS
        bwd_fast = fast and getattr(net, "backward_kernels", "auto") != "fp32"
        f16, frag, dy16 = _storage(net, bwd_fast)
        if dy16 == "f16p1" and cols > 3:
            dy16 = "f16"           # the single-product kernels evaluate the vector columns only: a [vector | features] forward keeps three
        # a render() under autograd earlier in this step left room in its workspace: append (StepWorkspace)
        pool = getattr(net, "_step_ws", None)
        if pool is not None and pool.ws is not None and bwd_fast and frag and m > 0 and pool.storage == (f16, dy16) and \
                pool.ws.saved.device == dev and _direct_ok(net, dev):
            mp = _round32(m)
            first = pool.take(mp)
            if first is not None:
                pts_p = pts if mp == m else torch.cat([pts, pts.new_zeros(mp - m, 3)])      # padding points: their upstream gradient is zero
                out_p = lib.vf_mlp16_fwd_train(net.geometry(), net.packed16_weights(), pts_p, cols > 3, pool.ws.saved, pool.ws.aux_vf,
                                               pool.ws.masks, save_f16=pool.ws.fwd_flags(), ws_first=first, ws_points=pool.total)
                if cols > 3:
                    out_p = torch.cat([out_p, pool.ws.feats(vf_h - 1)[first:first + mp]], dim=1)
                ctx.net, ctx.ws, ctx.dims, ctx.param_order = net, None, (m, vf_h, cols), list(params)
                ctx.region, ctx.fast = (pool, first, mp), True
                ctx.save_for_backward(out_p)
                return out_p[:m]
        ctx.region = None
        ws = _Workspace(m, vf_h, dev, f16=f16, frag=frag, dy16=dy16)
        if fast:
            out = lib.vf_mlp16_fwd_train(net.geometry(), net.packed16_weights(), pts, cols > 3, ws.saved, ws.aux_vf, ws.masks,
                                         save_f16=ws.fwd_flags())
            if cols > 3:   # [vector | features]: the kernel left the features in their workspace slot
                out = torch.cat([out, ws.feats(vf_h - 1)], dim=1)
        else:
            out = lib.vf_mlp_fwd_train(net.geometry(), net.packed_weights(), pts, cols, ws.saved, ws.aux_vf)
        ctx.net, ctx.ws, ctx.dims, ctx.param_order = net, ws, (m, vf_h, cols), list(params)
        ctx.fast = bwd_fast        # not re-derived in backward: net.precision may have changed in between (numerical Jacobian)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, d_out):
        net, ws = ctx.net, ctx.ws
        m, vf_h, cols = ctx.dims
        (out,) = ctx.saved_tensors
        dev = out.device
        d_out = d_out.float().contiguous()
        if ctx.region is not None:
            # this forward's points live in the step's shared workspace: the chain runs over their region, the weight gradients
            # wait for the end of the backward pass (StepWorkspace.flush)
            pool, first, mp = ctx.region
            if mp != m:
                d_out = torch.cat([d_out, d_out.new_zeros(mp - m, cols)])
            dy, dz_all = pool.gradients()
            lib.mlp_bwd_chain_bf16_ws(net.geometry(), _packed_bwd16(net, pool.ws.single), _head_rows(net), None, None, None, pool.ws.feats(vf_h - 1), pool.ws.masks,
                                      dy, pool.ws.dy_flags(), None, None, d_out, out, _offset_view(d_out, 3) if cols > 3 else None, cols, mp, None,
                                      dz_all, ws_first=first, ws_points=pool.total)
            pool.mark_done(first, mp, cols > 3)
            return (None, None, None, *[None for _ in ctx.param_order])
        dy = ws.new_dy()
        dz_vec = torch.empty(m, 4, device=dev)
        d_feats = None
        if cols > 3:
            d_feats = _offset_view(d_out, 3)
        fast = ctx.fast
        if fast:
            lib.mlp_bwd_chain_bf16_ws(net.geometry(), _packed_bwd16(net, ws.single), _head_rows(net), None, None, None, ws.feats(vf_h - 1), ws.masks,
                                      dy, ws.dy_flags(), None, None, d_out, out, d_feats, cols, m, None, dz_vec)
        else:
            lib.mlp_bwd_chain(net.geometry(), net.packed_weights(), _packed_bwd(net), None, None, None, ws.saved, dy,
                              None, None, d_out, out, d_feats, cols, m, None, dz_vec)
        # vector-only forward: the feature block of the last Linear was never evaluated -> no gradient for it
        skip = (vf_h - 1,) if (net._feature_dims() > 0 and cols == 3) else ()
        if len(ctx.param_order) == 1:         # anchor input only (vf_forward_autograd): the gradients are added in place
            _ensure_grads(net)
        grads = _weight_grads(net, _vf_inputs(net, [ws.saved[h] for h in range(vf_h)]), [dy[h] for h in range(vf_h)],
                              ws.aux_vf, dz_vec, m, skip=skip, fast=fast, x_f16=ws.f16, frag=ws.frag_forms(), ws_ref=(ws.saved, dy, 0))
        ctx.ws = None
        return (None, None, None, *[grads.get(p) for p in ctx.param_order])


class _RawPointer:
    """A (tensor, element offset) pair accepted by lib._ptr-style marshalling for strided column views."""

    def __init__(self, base: torch.Tensor, offset: int) -> None:
        self.base, self.offset = base, offset
        self.is_cuda, self.dtype = base.is_cuda, base.dtype

    def is_contiguous(self) -> bool:
        return True

    def data_ptr(self) -> int:
        return self.base.data_ptr() + 4 * self.offset


def _offset_view(t: torch.Tensor, col0: int) -> _RawPointer:
    return _RawPointer(t, col0)


def vf_forward_autograd(net, points, vector_only):
    plist = getattr(net, "_param_list_cache", None)
    if plist is None:
        plist = net._param_list_cache = list(net.parameters())
    pool = getattr(net, "_step_ws", None)
    if pool is not None and points.is_cuda and _direct_ok(net, points.device):
        # (the forward will either join the step's workspace or accumulate in place from its own: one anchor input is enough)
        return _VFForward.apply(net, points, vector_only, plist[0])
    return _VFForward.apply(net, points, vector_only, *plist)


def render_forward_autograd(net, points, normals, view_dirs, feats):
    """Stand-alone rendering-net forward with gradients (eval-mode BatchNorm): the training path differentiates the fused
    fine pass of render() instead; this serves the secondary entry points, layer by layer (batchstat.py)."""
    from .batchstat import render_forward_eval_autograd
    return render_forward_eval_autograd(net, points.reshape(-1, 3), normals.reshape(-1, 3), view_dirs.reshape(-1, 3),
                                        feats.reshape(-1, feats.shape[-1]))
