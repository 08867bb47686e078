"""Range guard of the f16x3 kernels (include/vfn.h, "Range guard of the f16x3 kernels").

The reference's MLPs are plain fp32 (models/vector_field/vector_field_network.py:177-208): no activation, input or weight
magnitude is out of bounds.  The f16x3 kernels carry every operand as two f16 halves, which covers |activation| < ~937,
|input coordinate| < ~937 and folded weights whose largest entry per layer is neither down in the f16 denormals nor beyond
the f16 range.  Default-initialised and normally trained networks sit well inside; a network that does not must not come back
with silently degraded values.  So:

* the kernels OR a bit into a device status word when the ReLU clamp or the input clamp acted (``vfn_f16x3_set_status``);
* ``vfn_pack16_weights`` leaves max |folded weight| per layer behind the pack;
* this guard reads both and, when either is out of range, switches its model to the exact-fp32 kernels (a warning names the
  reason; ``model.f16x3_disabled`` keeps it).  ``mode``:
    "lazy"    (default) the status is read back asynchronously, every ``LAZY_EVERY`` guarded calls and whenever the weights
              were re-packed; the switch happens at the start of the first call after the report arrived.  No
              synchronisation on the hot path; the flagged calls themselves have already returned clamped values.
              One exception (round 3): the FIRST gradient-free render on a (re)packed set of weights is checked like a strict
              call — one synchronisation per weight change, none per chunk of a view — so that a model whose two-product
              colours are out of tolerance never returns them, not even for the calls an asynchronous report would take to arrive.
    "strict"  every guarded call ends with a synchronising read of the status word and is REPEATED on the fp32 kernels when
              flagged: no call ever returns values the clamp touched.
    "off"     no reporting.

Colour self-check.  With ``model.colour_products == 2`` the colour branch of gradient-free renders uses its weights as their f16
roundings (csrc/vfn_mlp16.hip, M16_C2): a relative error of 2^-12 per weight that averages down to ~2e-5 on the colours of
networks with O(1) activations, but grows with the activations' magnitude — nothing a range check could bound a priori.  So it is
MEASURED: on the calls whose status is read back anyway (every call in "strict" mode, every ``LAZY_EVERY``-th and after every
re-pack in "lazy" mode) ``COLOUR_CHECK_RAYS`` rays of the call, spread over the batch, are evaluated a second time with three
products and the largest colour difference travels with the status word.  Above ``COLOUR_CHECK_TOL`` the model goes back to
``colour_products = 3`` (a warning says so; in "strict" mode the call is repeated).  One launch of <= 128 workgroups: 0.12 ms
where it runs, i.e. 0.2 % of a 4096-ray chunk in "lazy" mode.
"""
from __future__ import annotations

import warnings
from typing import Optional

import torch

from . import lib

WEIGHT_MAX_LO = 2.0 ** -9      # a layer whose largest folded weight is below this has its low halves (and soon its high
                               # halves) in the f16 denormals: < ~16 significant bits left instead of 22
WEIGHT_MAX_HI = 3.0e4          # f16 overflows at 65 504
LAZY_EVERY = 32
LAZY_REPACK_EVERY = 8          # ... and at most every 8th call while the weights change under every call (training)
COLOUR_CHECK_TOL = 5.0e-5      # two-product colours may differ from three-product colours by this much (contract: 1e-4 of the reference;
                               # in-family networks measure 1.4e-5 .. 2.0e-5)
COLOUR_CHECK_RAYS = 128


class _Watch:
    """Re-entrant: a guarded call made inside another guarded call (a vector-field forward issued by a render() that is itself
    under watch) joins the outer watch — only the OUTERMOST exit clears the kernels' status pointer, counts the call and evaluates
    the report; an inner watch reports ``flagged = False`` and leaves the verdict (and a strict-mode repeat) to the outer one."""

    def __init__(self, guard: "RangeGuard", dev) -> None:
        self.guard, self.dev, self.flagged, self.outermost = guard, dev, False, False

    def __enter__(self) -> "_Watch":
        g = self.guard
        self.outermost = g._depth == 0
        g._depth += 1
        if self.outermost:
            lib.f16x3_set_status(g._state(self.dev)["status"])
        return self

    def sample(self, out) -> None:
        """Colour self-check of a finished gradient-free render (see the module docstring); call inside the ``with`` block."""
        self.guard._colour_sample(self.dev, out)

    def __exit__(self, *exc) -> bool:
        g = self.guard
        g._depth -= 1
        if self.outermost:
            lib.f16x3_set_status(None)
            if exc[0] is None:
                self.flagged = g._after_call(self.dev)
        return False


class RangeGuard:
    def __init__(self, model) -> None:
        self.model = model
        self.mode = "lazy"
        self.colour_products_reason: Optional[str] = None     # why the guard moved the model back to three-product colours
        self._st: Optional[dict] = None
        self._calls = 0
        self._depth = 0            # nesting of watches (_Watch)

    # -- public ---------------------------------------------------------------------------------
    def active(self) -> bool:
        return self.mode != "off" and self.model.f16x3_disabled is None

    def poll(self) -> None:
        """Consume a finished asynchronous read-back, if any (start of a guarded call)."""
        st = self._st
        if st is not None and st["event"] is not None and st["event"].query():
            st["event"] = None
            self._evaluate(st)

    def watch(self, dev) -> _Watch:
        return _Watch(self, dev)

    def check_now(self, dev) -> Optional[str]:
        """Synchronising check (tests, diagnostics): the reason the f16x3 path was / is being switched off, or None."""
        st = self._state(dev)
        self._read_back(st, dev)
        st["event"].synchronize()
        st["event"] = None
        self._evaluate(st)
        return self.model.f16x3_disabled

    # -- internals --------------------------------------------------------------------------------
    def _state(self, dev) -> dict:
        st = self._st
        if st is None or st["status"].device != torch.device(dev):
            st = self._st = dict(status=torch.zeros(4, dtype=torch.int32, device=dev),
                                 host=torch.zeros(4 + 2 * lib.PACK16_STATS_WORDS, dtype=torch.int32).pin_memory(),
                                 event=None, pack_keys=None)
        return st

    def _nets(self):
        return (self.model.vector_field_network, self.model.rendering_network)

    def _read_back(self, st, dev) -> None:
        host = st["host"]
        host[:4].copy_(st["status"], non_blocking=True)
        for i, net in enumerate(self._nets()):
            cache = getattr(net, "_packed16_cache", None)
            lo = 4 + i * lib.PACK16_STATS_WORDS
            if cache is not None and cache[1].device == torch.device(dev):
                host[lo:lo + lib.PACK16_STATS_WORDS].copy_(cache[1][-4 * lib.PACK16_STATS_WORDS:].view(torch.int32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))      # the stream of `dev` the copies above were issued on (not the current device's)
        st["event"] = ev

    def _pack_keys(self):
        return tuple(getattr(net, "_packed16_cache", (None,))[0] for net in self._nets())

    def _reads_back(self, st) -> bool:
        """Will the call in flight end with a read-back of the status?  (same decision as _after_call, before it)"""
        if self.mode == "strict":
            return True
        nxt = self._calls + 1
        return st["event"] is None and (nxt % LAZY_EVERY == 0 or
                                        (self._pack_keys() != st["pack_keys"] and nxt - st.get("read_at", -LAZY_REPACK_EVERY) >= LAZY_REPACK_EVERY))

    def _colour_sample(self, dev, out) -> None:
        model = self.model
        vf, rn = self._nets()
        if model.colour_products != 2 or not model.uses_f16x3() or model._needs_grad() or vf.training or rn.training or \
                out is None or getattr(out, "coarse_colors", None) is None or out.coarse_colors.numel() == 0:
            return
        st = self._state(dev)
        if not self._reads_back(st):
            return
        with torch.no_grad():
            n = out.z_vals.shape[0]
            s_t = out.z_vals.shape[1]
            idx = torch.arange(0, n, max(1, n // COLOUR_CHECK_RAYS), device=dev)[:COLOUR_CHECK_RAYS]
            pts = out.points_coarse.reshape(n, s_t, 3)[idx].reshape(-1, 3).contiguous()
            dirs = out.ray_dirs.reshape(n, s_t, 3)[idx, 0].contiguous()
            two = out.coarse_colors.reshape(n, s_t, 3)[idx].reshape(-1, 3)
            _, three = lib.vf_render_fused16_fwd(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, s_t,
                                                 colour_products=3)
            diff = (three - two).abs()
            if getattr(model, "sparse_colours", False):        # colours exist only where the sample's weight is non-zero
                diff = diff * (two != 0).any(dim=1, keepdim=True)
            slot = st["status"][1:2].view(torch.float32)
            torch.maximum(slot, diff.max().reshape(1), out=slot)
            st["sampled"] = True

    def _after_call(self, dev) -> bool:
        st = self._state(dev)
        keys = self._pack_keys()
        repacked = keys != st["pack_keys"]
        st["pack_keys"] = keys
        self._calls += 1
        sampled = st.pop("sampled", False)
        if self.mode == "strict" or (repacked and sampled):      # (lazy: the first colour-checked render on these weights, see above)
            if st["event"] is not None:
                st["event"].synchronize()
            self._read_back(st, dev)
            st["event"].synchronize()
            st["event"] = None
            return self._evaluate(st)
        # lazy: an asynchronous read-back every LAZY_EVERY guarded calls and after a re-pack — but a TRAINING loop re-packs on every step
        # (the optimizer moved the weights a little): there, every LAZY_REPACK_EVERY-th call is early enough and spares each step the three
        # copies, the event and the evaluation (0.1 ms of host time at a 1 ms step)
        due = self._calls % LAZY_EVERY == 0 or (repacked and self._calls - st.get("read_at", -LAZY_REPACK_EVERY) >= LAZY_REPACK_EVERY)
        if st["event"] is None and due:
            st["read_at"] = self._calls
            self._read_back(st, dev)
        return False

    def _evaluate(self, st) -> bool:
        host = st["host"]
        as_float = host.view(torch.float32).tolist()      # one conversion: per-element reads of a tensor cost ~2 us each, ~30 of them per call
        flags = int(host[0])
        colour_diff = as_float[1]
        colour_flag = False
        if colour_diff > COLOUR_CHECK_TOL and self.model.colour_products == 2:
            warnings.warn(f"vf_nerf_amd: two-product colours differ from three-product colours by {colour_diff:.2e} (> {COLOUR_CHECK_TOL:.0e}) on "
                          "this model's data; colour_products is now 3", RuntimeWarning, stacklevel=3)
            self.model.colour_products = 3
            self.colour_products_reason = f"two-product colours were {colour_diff:.2e} off"
            st["status"][1:2].zero_()
            colour_flag = True
        reasons = []
        if flags & lib.STATUS_ACT_SATURATED:
            reasons.append("a hidden activation exceeded the split-f16 range (|x| > ~937)")
        if flags & lib.STATUS_INPUT_SATURATED:
            reasons.append("an input coordinate exceeded the split-f16 range (|p| > ~937)")
        for i, (net, tag) in enumerate(zip(self._nets(), ("vector-field", "rendering"))):
            lo = 4 + i * lib.PACK16_STATS_WORDS
            stats = as_float[lo:lo + lib.PACK16_STATS_WORDS]
            n_entries = len(_pack_entries(net))
            for e in range(n_entries):
                w = stats[e]
                if w == 0.0:
                    continue            # nothing packed yet (or an all-zero layer: exact in any representation)
                if w < WEIGHT_MAX_LO or w > WEIGHT_MAX_HI:
                    reasons.append(f"the {tag} net's pack entry {e} has max |folded weight| = {w:.3g}, outside "
                                   f"[{WEIGHT_MAX_LO:.2g}, {WEIGHT_MAX_HI:.2g}]")
        if not reasons:
            return colour_flag
        reason = "; ".join(reasons)
        warnings.warn("vf_nerf_amd: the f16x3 kernels left the range their split-f16 operands represent to fp32 accuracy (" + reason +
                      "); this model now runs on the exact-fp32 kernels", RuntimeWarning, stacklevel=3)
        self.model.f16x3_disabled = reason
        self.model.precision = "fp32"
        st["status"].zero_()
        return True


def _pack_entries(net):
    from .backward import _entries
    return list(_entries(net)) + [None]        # hidden entries in plan order, then the 3-channel head
