"""The two MLPs with their BatchNorm layers in TRAINING mode (batch statistics), on the layer-at-a-time HIP kernels of
``csrc/vfn_bstat.hip``.

Reference: ``VectorFieldNerf.train()`` (models/nerf/vector_field_nerf.py:139-150) puts both networks in training mode;
the trainer does so when the directional-derivative loss weight is non-zero (train/vector_field_nerf_train.py:140-141).
Then ``nn.BatchNorm1d`` normalises with the statistics of the batch (and updates the running ones), and the VF forward
appends three ``autograd.grad`` rows (vector_field_network.py:146-173).  Batch statistics couple all rows of a batch:

* forward: per layer Linear -> column sums -> (scale, shift) -> ReLU, one launch each;
* the three "Jacobian" rows are backward passes through those batch statistics (``grad_outputs = 1`` on every row of one
  output column): row m of the result is sum_r d y_c[r] / d p[m], not the per-point Jacobian of an eval-mode network;
* backward: per layer the two column sums of the BatchNorm backward, dZ, dX = dZ W and the weight-gradient partials.

Everything here is orchestration: buffers from torch, arithmetic in the kernels.  Gradients flow to the parameters and to
the feature / normal hand-off between the networks; the Jacobian rows are returned without a graph (the reference builds
one with ``create_graph=True`` but never differentiates it: render() uses them under ``no_grad`` or drops them, Q10).
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import torch

from . import lib
from .lib import Cols

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
# the BatchNorm backward's column sums from the dX product's registers (VFN_FUSE_BN_SUMS=0: the pass of its own, for A/B runs)
FUSE_BACKWARD_SUMS = os.environ.get("VFN_FUSE_BN_SUMS", "1") != "0"
PRESPLIT_W = os.environ.get("VFN_PRESPLIT_W", "1") != "0"      # W's 16-bit planes split once per layer product (vfn_linear_rows_ws) instead of per workgroup
INV_SQRT2 = 1.0 / math.sqrt(2.0)


def _up8(n: int) -> int:
    return (n + 7) & ~7


class _State:
    """What one forward leaves behind for the Jacobian rows and the backward."""

    def __init__(self) -> None:
        self.x: List[torch.Tensor] = []        # input matrix of every Linear (post-activation of the previous layer)
        self.z: List[torch.Tensor] = []        # pre-BatchNorm output of every BatchNorm'ed Linear
        self.coef: List[torch.Tensor] = []     # [4][n]: scale, shift, mean, rstd
        self.y: Optional[torch.Tensor] = None  # activated output of the last Linear
        self.m = 0
        self.batch_stats = True                # False: eval-mode BatchNorm (running statistics), rows are independent


# Round 6: a hidden layer's activated input x = post * relu(BatchNorm(z_prev)) is never written to HBM where both of its readers can form it
# from z_prev themselves — the next layer's forward product (vfn_linear_rows_fold) and that layer's weight-gradient product
# (vfn_weight_grad_partials_bf16_fold): same values bit for bit (the row pass's own expression), one [M, 256] write and two reads less per
# layer.  Applies to the 256-wide hidden layers on the split-f16 forward product: 6 of the vector-field net's 8, 3 of the rendering net's 4;
# the skip layer (six-product form: no registers for the coefficients), the last layer's input (narrow-head weight gradients) and every
# layer of a net on the exact kernels keep the row pass.  False: the row pass everywhere (rounds 3-5).
FOLD_ACTIVATIONS = os.environ.get("VFN_FOLD_ACTIVATIONS", "1") != "0"        # (the environment switch: same-box A/B, tools/ab_fold.sh)


class Folded:
    """The input of a layer as (z_prev, coef_prev, n_prev, post): act(z)[k] = post * max(z[k] * scale[k] + shift[k], 0) for k < n_prev,
    post * z[k] behind them (the skip layer's encoding columns, stored UNscaled in z_prev's buffer)."""

    def __init__(self, z: torch.Tensor, coef: torch.Tensor, n_prev: int, post: float) -> None:
        self.z, self.coef, self.n_prev, self.post = z, coef, int(n_prev), float(post)

    def materialise(self) -> torch.Tensor:
        """The activated matrix after all (a reader that cannot fold)."""
        m, width = self.z.shape
        x = torch.empty(m, width, device=self.z.device)
        lib.bstat_relu_rows(self.z, self.coef, m, self.n_prev, self.post, x)
        if width > self.n_prev:
            x[:, self.n_prev:] = self.z[:, self.n_prev:] * self.post
        return x


def _bn_layers(net) -> int:
    return sum(1 for i in range(net.num_layers) if net._bn(i) is not None)


def _require_shape(net) -> None:
    L = net.num_layers
    if any(net._bn(i) is None for i in range(L - 1)) or net._bn(L - 1) is not None:
        raise NotImplementedError("the training-mode path expects BatchNorm after every Linear but the last (confs/vf_nerf.conf)")


def _running_coef(bn) -> torch.Tensor:
    """[4][n] scale, shift, mean, rstd of an eval-mode BatchNorm1d (a handful of n-element torch ops on parameters)."""
    rstd = torch.rsqrt(bn.running_var.detach() + BN_EPS)
    scale = bn.weight.detach() * rstd
    return torch.stack([scale, bn.bias.detach() - bn.running_mean.detach() * scale, bn.running_mean.detach(), rstd]).contiguous()


def _arith(net, backward: bool, first_layer: bool = False) -> int:
    """Arithmetic of the layer GEMMs (csrc/vfn_bstat.hip): ``net.gemm_arithmetic`` = "split" (default since round 4) or "fp32" (the exact
    fp32 matrix instruction).  Split: the forward GEMMs on three f16 products per product (22 significant bits, like the fused f16x3
    kernels), the backward GEMMs dX = dZ W — operands of any magnitude — on bf16 in THREE parts (six products, 24 bits at fp32's exponent
    range), the full 256 x 256 weight gradients on the bf16 cores.  Both fp32-equivalent: the training-mode fixture's outputs stay inside
    the same bounds (normals 7e-5, directional derivatives 7e-5), the gradients with pinned ReLU masks inside 2e-3.  Measured
    (profiles/r04/linear_rows_microbench.txt): 0.43 ms (f16x3) / 0.53 ms (bf16x6) against 0.81-0.87 ms per 524 288 x 256 x 256 GEMM once the
    A operand goes through LDS in whole cache lines (the first split kernel fetched 32-byte pieces of 1-KiB-strided rows and was bound by
    that at 0.66 ms); the training-mode step 69 ms against 92 ms."""
    if not _split(net):
        return lib.GEMM_EXACT
    if backward:
        # Round 5: dX = dZ W on THREE bf16 products (16 significant bits at fp32's exponent range: the arithmetic of the eval-mode dX chain,
        # csrc/vfn_bwd16.hip) by default; "split24" keeps bf16 in three parts (six products, 24 bits: rounds 4's).  The training-mode gradients
        # against the oracle: worst 9.5e-5 against 8.4e-5 of a tensor's largest entry (bound 2e-3), the layer product 0.31 against 0.42 ms.
        return lib.GEMM_BF16X6 if getattr(net, "gemm_arithmetic", "split") == "split24" else lib.GEMM_SPLIT_BF16
    # The FIRST layer (and the skip layer, which re-reads the encoded input) reads raw inputs — coordinates of any scale beside the
    # encodings: its forward product runs in the form with fp32's exponent range too.  Later layers read BatchNorm'ed activations, inside the split f16 form's |x| < 1 023 (the launch reports
    # operands beyond it to the range guard, csrc/vfn_bstat.hip).
    return lib.GEMM_BF16X6 if (backward or first_layer) else lib.GEMM_SPLIT_F16


def running_stats_snapshot(*nets):
    """Copies of the BatchNorm running statistics (and batch counters) of the nets that are in batch-statistics mode: a guarded call that
    ends FLAGGED (a split-f16 operand out of range; strict mode repeats it on the exact products) must not have advanced them — neither
    with the flagged attempt's clamped values nor twice."""
    snap = []
    for net in nets:
        if net is None or not net._batch_statistics():
            continue
        for i in range(net.num_layers):
            bn = net._bn(i)
            if bn is not None:
                snap.append((net, bn, bn.running_mean.clone(), bn.running_var.clone(), bn.num_batches_tracked.clone()))
    return snap


def running_stats_restore(snap) -> None:
    with torch.no_grad():
        for net, bn, mean, var, count in snap:
            bn.running_mean.copy_(mean)
            bn.running_var.copy_(var)
            bn.num_batches_tracked.copy_(count)
    for net in {id(e[0]): e[0] for e in snap}.values():
        net._invalidate_packs()


def _planes(arith: int, n_out: int, k_in: int, dev):
    """The scratch for W's pre-split planes (lib.wplanes) where the product's arithmetic has two of them; None otherwise."""
    return lib.wplanes(n_out, k_in, dev) if (PRESPLIT_W and arith in (lib.GEMM_SPLIT_F16, lib.GEMM_SPLIT_BF16)) else None


def _padded(m: int, n: int, width: int, dev) -> torch.Tensor:
    """An [m, width] matrix whose columns [n, width) are zero and whose first n columns the caller's kernel writes in full: only the pad
    columns are cleared (a whole-matrix fill is a 0.6 GB write at 524 288 rows, ~1.2 ms of a training-mode step over all layers)."""
    t = torch.empty(m, width, device=dev)
    if width > n:
        t[:, n:].zero_()
    return t


def _split(net) -> bool:
    """Split-operand products on the 16-bit matrix cores?  Not when the facade asks for the exact kernels (``model.precision = "fp32"``,
    also what the range guard switches to) or the net itself does (``net.gemm_arithmetic = "fp32"``)."""
    mode = getattr(net, "gemm_arithmetic", "split")
    if mode not in ("split", "split24", "fp32"):
        raise ValueError(f"gemm_arithmetic must be 'split', 'split24' or 'fp32', got {mode!r}")
    return mode != "fp32" and getattr(net, "precision", "f16x3") != "fp32"


def _forward(net, x0: torch.Tensor, m: int, final_act: int, fill_skip=None, update_running: bool = True,
             batch_stats: bool = True) -> _State:
    """x0[M, ld] = input matrix of layer 0 (pad columns zero).  ``fill_skip(dst: Cols, scale)`` writes the skip layer's
    re-injected input (the positional encoding) next to the previous layer's output.  ``batch_stats=False``: BatchNorm in
    eval mode (running statistics, nothing updated) — the same layer-at-a-time kernels give a differentiable stand-alone
    forward where the fused kernels have no backward of their own."""
    _require_shape(net)
    dev = x0.device
    st = _State()
    st.m = m
    st.batch_stats = batch_stats
    L = net.num_layers
    skip = net._skip_layer()
    x = x0
    # (the test hook reads the activated matrices; the folding product takes W from its pre-split planes)
    fold = FOLD_ACTIVATIONS and PRESPLIT_W and _split(net) and not getattr(net, "_keep_state", False)
    counted = []              # the BatchNorm layers' batch counters: advanced in ONE launch at the end (a launch each was 5 us x 32 per step)
    for i in range(L - 1):
        lin, bn = net._linear(i), net._bn(i)
        n, k = lin.out_features, lin.in_features
        st.x.append(x)
        ar = _arith(net, False, i == 0 or i == skip)
        # will the NEXT layer read this layer's z through the fold?  (then, in front of the skip layer, z's buffer is as wide as that layer's
        # input and takes the re-injected encoding behind its n columns)
        nxt_k = net._linear(i + 1).in_features
        fold_next = fold and i + 1 <= L - 2 and nxt_k == 256 and 128 < net._linear(i + 1).out_features <= 256 and \
            _arith(net, False, i + 1 == skip) == lib.GEMM_SPLIT_F16      # (the skip layer's six-product form keeps the row pass)
        z = _padded(m, n, nxt_k if (fold_next and i + 1 == skip) else _up8(n), dev)

        def product(stats_part):
            if isinstance(x, Folded):
                lib.linear_rows_fold(x.z, x.coef, x.n_prev, x.post, lin.weight.detach(), lin.bias.detach(), m, n, k, z, stats_part=stats_part,
                                     arith=ar, planes=_planes(ar, n, k, dev))
            else:
                lib.linear_rows(x, lin.weight.detach(), lin.bias.detach(), m, n, k, z, stats_part=stats_part, arith=ar, planes=_planes(ar, n, k, dev))

        if batch_stats:
            parts = lib.linear_rows_stat_parts(m)
            part = torch.empty(parts, 2, n, device=dev)
            product(part)
            sums = torch.empty(2, n, dtype=torch.float64, device=dev)
            lib.colsum_finish(part, parts, 2 * n, sums)
            coef = torch.empty(4, n, device=dev)
            lib.bstat_finalize(sums, m, n, bn.weight.detach(), bn.bias.detach(), BN_EPS, BN_MOMENTUM,
                               bn.running_mean if update_running else None, bn.running_var if update_running else None, coef)
            if update_running:
                counted.append(bn.num_batches_tracked)
        else:
            product(None)
            coef = _running_coef(bn)
        st.z.append(z)
        st.coef.append(coef)
        if fold_next:            # the next product (and its weight gradients) form the activation from z themselves
            if i + 1 == skip:
                fill_skip(Cols(z, n), 1.0)                   # (unscaled: the fold multiplies every column by 1 / sqrt(2))
            x = Folded(z, coef, n, INV_SQRT2 if i + 1 == skip else 1.0)
        elif i + 1 == skip:        # next input = cat([h, pe]) / sqrt(2)  (vector_field_network.py:192-193)
            x = _padded(m, nxt_k, _up8(nxt_k), dev)      # (columns [0, n) by the row pass, [n, nxt_k) by fill_skip)
            lib.bstat_relu_rows(z, coef, m, n, INV_SQRT2, x)
            fill_skip(Cols(x, n), INV_SQRT2)
        else:
            x = _padded(m, n, _up8(nxt_k), dev)          # (the row pass writes columns [0, n))
            lib.bstat_relu_rows(z, coef, m, n, 1.0, x)
    last = net._linear(L - 1)
    st.x.append(x)
    y = _padded(m, last.out_features, _up8(last.out_features), dev)
    ar = _arith(net, False, L == 1)
    lib.linear_rows(x, last.weight.detach(), last.bias.detach(), m, last.out_features, last.in_features, y, act=final_act, arith=ar,
                    planes=_planes(ar, last.out_features, last.in_features, dev))
    st.y = y
    if counted:
        torch._foreach_add_(counted, 1)
    if batch_stats and update_running:
        # vfn_bstat_finalize advanced running_mean / running_var through raw pointers: their _version did not move, so the
        # folded-BatchNorm packs keyed on (data_ptr, _version) would survive into a later eval()/render() with stale statistics
        net._invalidate_packs()
    if getattr(net, "_keep_state", False):     # test hook: expose the activations of the latest forward
        net._debug_state = st
    return st


def _groups(m: int) -> int:
    return max(1, min(256, m // 256))


class _ParamGrads:
    """Weight-gradient partial slabs of every layer, un-folded (summed) in one launch at the end."""

    def __init__(self, net, m: int) -> None:
        self.net, self.m, self.G = net, m, _groups(m)
        self.grads: Dict[torch.nn.Parameter, torch.Tensor] = {}
        self.unfold: List[dict] = []

    def _out(self, p):
        if p not in self.grads:
            self.grads[p] = torch.empty_like(p)
        return self.grads[p]

    def linear(self, i: int, dz: torch.Tensor, x: torch.Tensor) -> None:
        """dW_i = dZ^T X, db_i = column sums of dZ, in pieces the weight-gradient kernel has shapes for: output rows in
        blocks of 256 (or one 32-row block for a narrow head), input columns in blocks of 256 with a tail of <= 64."""
        lin = self.net._linear(i)
        n, k = lin.out_features, lin.in_features
        dev, G, m = dz.device, self.G, self.m
        split = _split(self.net)
        row_blocks = []
        if n % 256 != 0 and n % 256 <= 32 and n > 32 and split:
            # 259 = 256 + 3: any partition of the rows is valid; this one keeps the wide block at column 0 of dZ (16-byte aligned rows: the
            # bf16 product), the last three rows go through the narrow-head shape
            row_blocks += [(r, 256, 0) for r in range(0, n - n % 256, 256)]
            row_blocks.append((n - n % 256, n % 256, 2))
        elif n % 256 != 0 and n % 256 <= 32 and n > 32:   # 259 = 3 + 256: rows 0..2 as the narrow head, the rest as one block
            row_blocks.append((0, n % 256, 2))
            row_blocks += [(r, 256, 0) for r in range(n % 256, n, 256)]
        elif n <= 32:
            row_blocks.append((0, n, 2))
        else:
            row_blocks += [(r, min(256, n - r), 0) for r in range(0, n, 256)]
        # input columns: a 256-wide "act" block when there is one, and a <= 64-wide "aux" block for the rest
        if k <= 64:
            col_blocks = [("aux", 0, k)]
        elif k == 256:
            col_blocks = [("act", 0, 256)]
        elif 256 < k <= 320 and split:
            col_blocks = [("act", 0, 256), ("aux", 256, k - 256)]          # (any partition of the columns is valid: the aligned one)
        elif 256 < k <= 320:
            col_blocks = [("aux", 0, k - 256), ("act", k - 256, 256)]      # rendering net: [p, PE(d), n | 256 features]
        else:
            raise NotImplementedError(f"weight gradients for in_features={k}")
        folded = x if isinstance(x, Folded) else None
        if folded is not None:
            foldable = split and k == 256 and all(not head for _, _, head in row_blocks) and folded.z.shape[1] >= 256
            if not foldable:
                x, folded = folded.materialise(), None
            else:
                x = folded.z
        for r0, rows, head in row_blocks:
            slab_rows = 32 if head else 256
            u = dict(w=lin.weight.detach(), b_lin=lin.bias.detach(), g_w=self._out(lin.weight), g_b=self._out(lin.bias),
                     rows=rows, row_off=r0, in_dim=k, slab_rows=slab_rows, scale=1.0)
            db_part = torch.empty(G, slab_rows, device=dev)
            u["db"] = db_part
            first = True
            for kind, c0, nc in col_blocks:
                dzv, xv = Cols(dz, r0), Cols(x, c0)
                if head:
                    if kind != "act":
                        raise NotImplementedError("a narrow head reading fewer than 256 inputs")
                    part = torch.empty(G, 32, 256, device=dev)
                    lib.weight_grad_partials(2, dzv, dzv.ld, rows, xv, xv.ld, nc, m, G, part, db_part if first else None)
                    u.update(dw_act=part, act_c0=c0, act_nc=nc)
                elif kind == "act":
                    part = torch.empty(G, 256, 256, device=dev)
                    # 256 columns of dZ from r0 on: all valid, or the last block of a layer whose dZ was allocated 256 wide with zero pad
                    # columns (`_backward`: 217 outputs in front of the skip layer)
                    dz_ok = r0 % 4 == 0 and r0 + 256 <= dzv.ld and (rows == 256 or r0 + rows == n)
                    if folded is not None:
                        if not (dz_ok and c0 == 0):
                            raise lib.VfnError(f"layer {i}: a folded input needs the aligned 256-column weight-gradient product")
                        lib.weight_grad_partials_bf16_fold(dzv, xv, folded.coef, folded.n_prev, folded.post, m, G, part, db_part if first else None)
                    elif split and dz_ok and nc == 256 and c0 % 4 == 0 and c0 + 256 <= xv.ld:
                        # a 256 x 256 product over whole 16-byte pieces of both matrices' rows: the bf16 matrix cores (three products on split
                        # operands, csrc/vfn_dw16.hip — what the fused path's row-major backward uses), 4x the fp32 matrix instruction's rate
                        lib.weight_grad_partials_bf16_cols(dzv, xv, m, G, part, db_part if first else None)
                    else:
                        lib.weight_grad_partials(0, dzv, dzv.ld, rows, xv, xv.ld, nc, m, G, part, db_part if first else None)
                    u.update(dw_act=part, act_c0=c0, act_nc=nc)
                else:
                    part = torch.empty(G, 256, 64, device=dev)
                    lib.weight_grad_partials(1, dzv, dzv.ld, rows, xv, xv.ld, nc, m, G, part, db_part if first else None)
                    u.update(dw_aux=part, aux_c0=c0, aux_nc=nc)
                first = False
            self.unfold.append(u)

    def batchnorm(self, i: int, sums: torch.Tensor) -> None:
        bn = self.net._bn(i)
        self.grads[bn.bias] = sums[0].float()        # d beta = sum g', d gamma = sum g' x_hat
        self.grads[bn.weight] = sums[1].float()

    def finish(self) -> Dict[torch.nn.Parameter, torch.Tensor]:
        if self.unfold:
            lib.unfold_weight_grads(self.unfold, self.G)
        return self.grads


def _backward(net, st: _State, dz_last: torch.Tensor, pg: Optional[_ParamGrads], want_dx0: bool):
    """dz_last[M, up8(n_L)]: gradient wrt the pre-activation of the last Linear (pad columns zero).
    -> (dX_0 or None, skip piece (Cols of the gradient wrt the re-injected encoding, scale) or None)."""
    m, L = st.m, net.num_layers
    dev = dz_last.device
    skip = net._skip_layer()
    dz = dz_last
    skip_piece = None
    for i in range(L - 1, -1, -1):
        lin = net._linear(i)
        n, k = lin.out_features, lin.in_features
        if pg is not None:
            pg.linear(i, dz, st.x[i])
        if i == 0 and not want_dx0:
            return None, skip_piece
        g = _padded(m, k, _up8(k), dev)
        if i == 0:
            lib.linear_rows(dz, lin.weight.detach(), None, m, k, n, g, transpose_w=True, arith=_arith(net, True),
                            planes=_planes(_arith(net, True), k, n, dev))                                         # dX = dZ W
            return g, skip_piece
        # BatchNorm + ReLU of layer i-1, whose (scaled) output is columns [0, n_prev) of x[i]
        n_prev = net._linear(i - 1).out_features
        post = INV_SQRT2 if i == skip else 1.0
        z, coef = st.z[i - 1], st.coef[i - 1]
        if FUSE_BACKWARD_SUMS and _arith(net, True) in (lib.GEMM_SPLIT_BF16, lib.GEMM_BF16X6) and n_prev <= 256:
            # dX = dZ W and, from the product while it is in registers, the two column sums of the BatchNorm backward (the separate pass
            # over g and z that computed them was 16 % of a training-mode step)
            parts = lib.linear_rows_stat_parts(m)
            part = torch.empty(parts, 2, n_prev, device=dev)
            lib.linear_rows_dx_sums(dz, lin.weight.detach(), m, k, n, g, z, coef, n_prev, post, part, arith=_arith(net, True),
                                    planes=_planes(_arith(net, True), k, n, dev))
        else:
            lib.linear_rows(dz, lin.weight.detach(), None, m, k, n, g, transpose_w=True, arith=_arith(net, True),
                            planes=_planes(_arith(net, True), k, n, dev))
            parts = lib.bstat_row_parts(m)
            part = torch.empty(parts, 2, n_prev, device=dev)
            lib.bstat_relu_bwd_sums(g, z, coef, m, n_prev, post, part)
        if i == skip:
            skip_piece = (Cols(g, n_prev), INV_SQRT2)
        sums = torch.empty(2, n_prev, dtype=torch.float64, device=dev)
        lib.colsum_finish(part, parts, 2 * n_prev, sums)
        if pg is not None:
            pg.batchnorm(i - 1, sums)
        # (zero pad columns — up to 256 where the bf16 weight-gradient product reads whole blocks)
        dz = _padded(m, n_prev, 256 if (192 < n_prev < 256 and _split(net)) else _up8(n_prev), dev)
        # batch statistics: dz = gamma rstd (g' - mean g' - x_hat mean(g' x_hat)); running statistics: dz = gamma rstd g'
        lib.bstat_relu_bwd_rows(g, z, coef, sums if st.batch_stats else torch.zeros_like(sums), m, n_prev, post, dz)
    raise AssertionError("unreachable")


# ------------------------------------------------------------------------------------------------
# vector-field network
# ------------------------------------------------------------------------------------------------
def _vf_forward_state(net, pts: torch.Tensor, batch_stats: bool = True) -> _State:
    m = pts.shape[0]
    L_pe = net._multires()
    k0 = net._linear(0).in_features
    x0 = torch.zeros(m, _up8(k0), device=pts.device)
    lib.embed_rows(pts, m, L_pe, x0)
    return _forward(net, x0, m, lib.ACT_TANH, fill_skip=lambda dst, scale: lib.embed_rows(pts, m, L_pe, dst, scale),
                    batch_stats=batch_stats)


def _vf_point_grads(net, st: _State, pts: torch.Tensor, dz_last: torch.Tensor, pg: Optional[_ParamGrads]) -> torch.Tensor:
    """Backward from dz_last down to the points -> [M,3]."""
    m = st.m
    dx0, piece = _backward(net, st, dz_last, pg, want_dx0=True)
    d_pts = torch.empty(m, 3, device=pts.device)
    lib.embed_rows_bwd(pts, m, net._multires(), dx0, 1.0, piece[0] if piece else None, piece[1] if piece else 0.0, d_pts)
    return d_pts


def vf_jacobian_rows(net, st: _State, pts: torch.Tensor) -> torch.Tensor:
    """[M,9] = [d y_0 / d p, d y_1 / d p, d y_2 / d p] with grad_outputs = 1 on every row (vector_field_network.py:150-172)."""
    m = st.m
    n_out = net._linear(net.num_layers - 1).out_features
    rows = []
    for c in range(3):
        dz = torch.zeros(m, _up8(n_out), device=pts.device)
        lib.act_bwd_rows(lib.ACT_TANH, None, st.y, m, n_out, dz, onehot_col=c)
        rows.append(_vf_point_grads(net, st, pts, dz, None))
    return torch.cat(rows, dim=1)


class _VFTrainMode(torch.autograd.Function):
    """points[M,3] -> [M, 3 + F (+ 9)] with batch-statistics BatchNorm; gradients to the parameters (and the points)."""

    @staticmethod
    def forward(ctx, net, points, want_jacobian, *params):
        pts = points.detach().reshape(-1, 3).float().contiguous()
        m = pts.shape[0]
        st = _vf_forward_state(net, pts, batch_stats=net.training)
        n_out = net._linear(net.num_layers - 1).out_features
        cols = [st.y[:, :n_out]]
        if want_jacobian:
            cols.append(vf_jacobian_rows(net, st, pts))
        out = torch.cat(cols, dim=1) if len(cols) > 1 else st.y[:, :n_out].contiguous()
        ctx.net, ctx.st, ctx.pts, ctx.n_out, ctx.param_order = net, st, pts, n_out, list(params)
        ctx.points_grad = points.requires_grad
        return out

    @staticmethod
    def backward(ctx, d_out):
        net, st, pts, n_out = ctx.net, ctx.st, ctx.pts, ctx.n_out
        m = st.m
        if d_out.shape[1] > n_out and bool((d_out[:, n_out:] != 0).any()):
            raise NotImplementedError("gradients through the Jacobian columns of the train-mode VF forward (a double backward) "
                                      "are not implemented; the reference's render() never requests them (Q10)")
        dy = d_out[:, :n_out].float().contiguous()
        dz = torch.zeros(m, _up8(n_out), device=dy.device)
        lib.act_bwd_rows(lib.ACT_TANH, dy, st.y, m, n_out, dz)
        pg = _ParamGrads(net, m)
        d_pts = None
        if ctx.points_grad:
            d_pts = _vf_point_grads(net, st, pts, dz, pg)
        else:
            _backward(net, st, dz, pg, want_dx0=False)
        grads = pg.finish()
        ctx.st = None
        return (None, d_pts, None, *[grads.get(p) for p in ctx.param_order])


def vf_forward_train_mode(net, points: torch.Tensor, want_jacobian: bool = True) -> torch.Tensor:
    """``VectorFieldNetwork.forward`` in training mode: [M, 3 + F + 9] (or [M, 3 + F] with ``want_jacobian=False``, for
    callers that drop the Jacobian columns).  Unlike the reference (vector_field_network.py:148) the caller's ``points`` are
    not switched to ``requires_grad`` in place: the gradient wrt the points is produced only when they already ask for it."""
    if not points.is_cuda:
        raise lib.VfnError("the training-mode forward runs on the device (no CPU fallback)")
    return _VFTrainMode.apply(net, points, want_jacobian, *list(net.parameters()))


def vf_forward_eval_rows(net, points: torch.Tensor) -> torch.Tensor:
    """``VectorFieldNetwork.forward`` in eval mode for a geometry the fused kernels are not specialised for (hidden widths
    other than 256, e.g. a narrow checkpoint): the same layer-at-a-time row kernels with the running statistics -> [M, 3 + F].
    Differentiable where the weight-gradient kernels have the layer's shape (``_ParamGrads.linear``)."""
    if not points.is_cuda:
        raise lib.VfnError("the layer-at-a-time forward runs on the device (no CPU fallback)")
    if net.training:
        raise ValueError("vf_forward_eval_rows is the eval-mode path")
    if torch.is_grad_enabled() and (points.requires_grad or any(p.requires_grad for p in net.parameters())):
        return _VFTrainMode.apply(net, points, False, *list(net.parameters()))
    pts = points.detach().reshape(-1, 3).float().contiguous()
    st = _vf_forward_state(net, pts, batch_stats=False)
    return st.y[:, :net._linear(net.num_layers - 1).out_features].contiguous()


# ------------------------------------------------------------------------------------------------
# rendering network
# ------------------------------------------------------------------------------------------------
class _RenderTrainMode(torch.autograd.Function):
    """cat[p, PE(d), n, feat] -> ReLU MLP with batch-statistics BatchNorm -> sigmoid (rendering_network.py:62-108, mode idr);
    gradients to the parameters and the features."""

    @staticmethod
    def forward(ctx, net, batch_stats, points, normals, view_dirs, feats, *params):
        m = points.shape[0]
        dev = points.device
        pe = 3 + 6 * net._multires() if net._multires() > 0 else 3
        k0 = net._linear(0).in_features
        f = feats.shape[1]
        if k0 != 3 + pe + 3 + f:
            raise lib.VfnError(f"rendering net: first layer reads {k0} columns, inputs give {3 + pe + 3 + f}")
        x0 = torch.zeros(m, _up8(k0), device=dev)
        lib.embed_rows(points.detach().reshape(-1, 3).float().contiguous(), m, 0, Cols(x0, 0))
        lib.embed_rows(view_dirs.detach().reshape(-1, 3).float().contiguous(), m, net._multires(), Cols(x0, 3))
        lib.embed_rows(normals.detach().reshape(-1, 3).float().contiguous(), m, 0, Cols(x0, 3 + pe))
        x0[:, 6 + pe:6 + pe + f].copy_(feats.detach())
        st = _forward(net, x0, m, lib.ACT_SIGMOID, batch_stats=batch_stats)
        ctx.net, ctx.st, ctx.cols, ctx.param_order = net, st, (6 + pe, f), list(params)
        # rendering_network.py:76-77: with detach_normals=False the colours' gradient also reaches the normals
        ctx.normal_cols = (3 + pe) if (ctx.needs_input_grad[3] and not net.config.detach_normals) else None
        return st.y[:, :3].contiguous()

    @staticmethod
    def backward(ctx, d_colors):
        net, st = ctx.net, ctx.st
        m = st.m
        c0, f = ctx.cols
        dz = torch.zeros(m, 8, device=d_colors.device)
        lib.act_bwd_rows(lib.ACT_SIGMOID, d_colors.float().contiguous(), st.y, m, 3, dz)
        pg = _ParamGrads(net, m)
        dx0, _ = _backward(net, st, dz, pg, want_dx0=True)
        grads = pg.finish()
        ctx.st = None
        d_normals = None if ctx.normal_cols is None else dx0[:, ctx.normal_cols:ctx.normal_cols + 3]
        return (None, None, None, d_normals, None, dx0[:, c0:c0 + f], *[grads.get(p) for p in ctx.param_order])


def render_forward_train_mode(net, points, normals, view_dirs, feats) -> torch.Tensor:
    if not points.is_cuda:
        raise lib.VfnError("the training-mode forward runs on the device (no CPU fallback)")
    return _RenderTrainMode.apply(net, True, points, normals, view_dirs, feats.float(), *list(net.parameters()))


def render_forward_eval_autograd(net, points, normals, view_dirs, feats) -> torch.Tensor:
    """Stand-alone ``RenderingNetwork.forward`` under autograd with eval-mode BatchNorm (the secondary entry points
    get_colors / get_weights_and_color of the facade with gradients enabled): the layer-at-a-time kernels with the running
    statistics, gradients to the parameters and the features."""
    if not points.is_cuda:
        raise lib.VfnError("the differentiable rendering-net forward runs on the device (no CPU fallback)")
    return _RenderTrainMode.apply(net, False, points, normals, view_dirs, feats.float(), *list(net.parameters()))


# ------------------------------------------------------------------------------------------------
# per-ray density / weights / composite under autograd, for render() paths that call the networks one by one
# ------------------------------------------------------------------------------------------------
class _RayComposite(torch.autograd.Function):
    """(normals[M,3], colors[M,3]) -> (rgb[N,3], depth[N,1], weights[N,S]) through vfn_ray_density_weights; gradients to the
    normals, the colours and the three density scalars (vector_field_nerf.py:308-323,442-474)."""

    @staticmethod
    def forward(ctx, model, normals, colors, z, ray_dirs, *density_params):
        scal = model.density.raw_scalars()
        normals = normals.detach().float().contiguous()
        colors = colors.detach().float().contiguous()
        _, weights, _, rgb, depth = lib.ray_density_weights(model._density_params(), normals, ray_dirs, z, scal, colors=colors,
                                                            want_sigma=False)
        ctx.model, ctx.names = model, [n for n, _ in model.density.named_parameters()]
        ctx.save_for_backward(normals, colors, z, ray_dirs, scal)
        return rgb, depth, weights

    @staticmethod
    def backward(ctx, d_rgb, d_depth, d_weights):
        normals, colors, z, ray_dirs, scal = ctx.saved_tensors
        n, s = z.shape
        dev = z.device

        def cont(t, shape):
            return None if t is None else t.reshape(shape).float().contiguous()

        dn = torch.zeros(n * s, 3, device=dev)
        dc = torch.empty(n * s, 3, device=dev)
        dscal = torch.zeros(3, device=dev)
        lib.ray_density_weights_bwd(ctx.model._density_params(), normals, ray_dirs, z, scal, colors, cont(d_rgb, (n, 3)),
                                    cont(d_depth, (n,)), cont(d_weights, (n, s)), dn, dc, dscal)
        by_name = {"beta": dscal[0], "mean": dscal[1], "scale": dscal[2]}
        dens = [by_name[name].reshape(p.shape) for name, p in ctx.model.density.named_parameters()]
        return (None, dn, dc, None, None, *dens)


def ray_composite(model, normals, colors, z, ray_dirs):
    return _RayComposite.apply(model, normals, colors, z, ray_dirs, *list(model.density.parameters()))
