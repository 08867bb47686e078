"""The two MLPs of the hot path as parameter containers whose ``forward`` runs the fused HIP kernels.

``VectorFieldNetwork`` mirrors ``models/vector_field/vector_field_network.py:14-208`` and
``RenderingNetwork`` mirrors ``models/vector_field/rendering_network.py:13-108`` of the reference:
same constructor argument (the config dataclass), same ``nn.ModuleList`` layout and therefore the same
``state_dict`` keys (``layers.{i}.0.weight``, ``layers.{i}.1.running_mean``, ``layers.8.weight`` ...), same
RNG consumption at construction (identical default initialisation for a given ``torch.manual_seed``).
The arithmetic lives in ``csrc/vfn_mlp.hip``; the live parameters are re-packed (BatchNorm folded,
MFMA fragment order) whenever one of them changed.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import lib


def _pe_dim(multires: int, base: int = 3) -> int:
    return base + 2 * base * multires if multires > 0 else base


class _PackedMLP(nn.Module):
    """Shared machinery: layer construction, geometry descriptor, packed-weight cache."""

    _kind: int = lib.NET_VF

    def _build_layers(self, dims: List[int], out_dims: List[int], batch_norm: bool, weight_norm: bool) -> None:
        if weight_norm:
            raise NotImplementedError("weight_norm=True is not supported by the HIP path (the shipped "
                                      "confs/vf_nerf.conf uses batch_norm)")
        self.num_layers = len(out_dims)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            lin = nn.Linear(dims[i], out_dims[i])
            if batch_norm and i < self.num_layers - 1:
                self.layers.append(nn.Sequential(lin, nn.BatchNorm1d(out_dims[i])))
            else:
                self.layers.append(lin)
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._geom = None

    # -- parameter access -----------------------------------------------------------------
    def _linear(self, i: int) -> nn.Linear:
        layer = self.layers[i]
        return layer[0] if isinstance(layer, nn.Sequential) else layer

    def _bn(self, i: int) -> Optional[nn.BatchNorm1d]:
        layer = self.layers[i]
        return layer[1] if isinstance(layer, nn.Sequential) else None

    def _layer_tensors(self) -> List[dict]:
        out = []
        for i in range(self.num_layers):
            lin, bn = self._linear(i), self._bn(i)
            d = dict(weight=lin.weight, bias=lin.bias)
            if bn is not None:
                d.update(bn_weight=bn.weight, bn_bias=bn.bias, bn_mean=bn.running_mean, bn_var=bn.running_var)
            out.append(d)
        return out

    def geometry(self) -> lib.NetGeom:
        if self._geom is None:
            self._geom = lib.make_geom(
                self.num_layers, self._multires(), self._skip_layer(), self._feature_dims(),
                [self._linear(i).in_features for i in range(self.num_layers)],
                [self._linear(i).out_features for i in range(self.num_layers)],
                [self._bn(i) is not None for i in range(self.num_layers)])
        return self._geom

    def _invalidate_packs(self) -> None:
        """Mark every pack built from the parameters / running statistics stale (they are re-packed, into the same buffers, on
        next use).  Needed where a kernel wrote one of those tensors through its raw pointer, which does not advance
        ``tensor._version`` (the flat optimizer step, the running statistics of a training-mode forward)."""
        self._pack_epoch = getattr(self, "_pack_epoch", 0) + 1

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda() / .float() replace buffer tensors: drop the cached tensor list and the packs built from it."""
        out = super()._apply(fn, *args, **kwargs)
        for name in ("_pack_tensors", "_packed16_cache", "_packed_bwd_cache", "_packed_bwd16_cache", "_packed_bwd16r_cache"):
            if hasattr(self, name):
                delattr(self, name)
        self._packed, self._packed_key = None, None
        return out

    def _pack_key(self):
        """(data_ptr, version) of every tensor the packs are built from; the tensor list itself is collected once (the
        module tree is fixed after construction — walking it on every call costs more than a kernel launch)."""
        tensors = getattr(self, "_pack_tensors", None)
        if tensors is None:
            tensors = self._pack_tensors = [t for d in self._layer_tensors() for t in d.values()]
        return tensors, (getattr(self, "_pack_epoch", 0),) + tuple([(t.data_ptr(), t._version) for t in tensors])

    def packed_weights(self) -> torch.Tensor:
        """Packed (BN-folded, fragment-ordered) weights for the current parameter values."""
        tensors, key = self._pack_key()
        dev = tensors[0].device
        if self._packed is None or self._packed.device != dev or key != self._packed_key:
            geom = self.geometry()
            if self._packed is None or self._packed.device != dev:
                self._packed = torch.empty(lib.packed_size(self._kind, geom), device=dev)
            with torch.no_grad():
                lib.pack_weights(self._kind, geom, [{k: v.detach() for k, v in d.items()}
                                                    for d in self._layer_tensors()], self._packed)
            self._packed_key = key
        return self._packed

    def packed16_weights(self) -> torch.Tensor:
        """f16x3 pack (hi/lo halves, fragment order) for the current parameter values."""
        tensors, key = self._pack_key()
        dev = tensors[0].device
        cache = getattr(self, "_packed16_cache", None)
        if cache is None or cache[0] != key or cache[1].device != dev:
            geom = self.geometry()
            buf = cache[1] if (cache is not None and cache[1].device == dev) else \
                torch.empty(lib.pack16_size(self._kind, geom), dtype=torch.uint8, device=dev)
            with torch.no_grad():
                lib.pack16_weights(self._kind, geom, [{k: v.detach() for k, v in d.items()}
                                                      for d in self._layer_tensors()], buf)
            self._packed16_cache = (key, buf)
        return self._packed16_cache[1]

    def supports_f16x3(self) -> bool:
        """True when the f16x3 kernels are specialised for this geometry (the shipped layer shapes); otherwise the
        facade uses the exact-fp32 HIP kernels (never a CPU path)."""
        ok = getattr(self, "_f16x3_ok", None)
        if ok is None:
            try:
                lib.pack16_size(self._kind, self.geometry())
                ok = True
            except lib.VfnError:
                ok = False
            self._f16x3_ok = ok
        return ok

    def supports_fused(self) -> bool:
        """True when the fused kernels (fp32 or f16x3) are specialised for this geometry: hidden width 256 everywhere
        (include/vfn.h).  Other geometries — e.g. a narrow checkpoint — run layer by layer on the generic row kernels of
        ``csrc/vfn_bstat.hip`` with eval-mode BatchNorm (``batchstat.py``); still HIP, never a CPU path."""
        ok = getattr(self, "_fused_ok", None)
        if ok is None:
            try:
                lib.packed_size(self._kind, self.geometry())
                ok = True
            except lib.VfnError:
                ok = False
            self._fused_ok = ok
        return ok

    def _batch_statistics(self) -> bool:
        """True in training mode with BatchNorm layers: they normalise with the statistics of the batch, which the fused
        kernels (BatchNorm folded into the weights) cannot express — ``batchstat.py`` runs the layers one at a time."""
        return self.training and any(self._bn(i) is not None for i in range(self.num_layers))


class VectorFieldNetwork(_PackedMLP):
    _kind = lib.NET_VF

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        # arithmetic of gradient-free vector-only queries: "f16x3" (split-half f16 MFMA, fp32-equivalent) or "fp32";
        # the facade's ``precision`` setter writes it.  Full-width and gradient-carrying forwards are always fp32.
        self.precision = "f16x3"
        pe = _pe_dim(config.embedder_multires, config.input_dims)
        dims = [pe if config.embedder_multires > 0 else config.input_dims] + list(config.dimensions) + \
               [config.output_dims + config.feature_vector_dims]
        self.skip_connection_in: List[int] = list(config.skip_connection_in or [])
        outs = []
        for i in range(len(dims) - 1):
            outs.append(dims[i + 1] - dims[0] if (i + 1) in self.skip_connection_in else dims[i + 1])
        if config.xavier_init:
            raise NotImplementedError("xavier_init=True is not supported (the shipped conf sets it False)")
        if config.dropout and config.dropout_probability > 0.0:
            raise NotImplementedError("dropout is not supported on the HIP path (the shipped conf disables it)")
        if len(self.skip_connection_in) > 1:
            raise NotImplementedError("at most one skip connection is supported")
        self._build_layers(dims[:-1], outs, config.batch_norm, config.weight_norm)
        assert (config.init in ["center", "exterior", ""]) or ("exterior" in config.init), \
            "init must be one of [center, exterior, '']"

    def _multires(self) -> int:
        return int(self.config.embedder_multires)

    def _skip_layer(self) -> int:
        return self.skip_connection_in[0] if self.skip_connection_in else -1

    def _feature_dims(self) -> int:
        return int(self.config.feature_vector_dims)

    @property
    def init(self) -> str:
        return self.config.init

    @init.setter
    def init(self, value: str) -> None:
        self.config.init = value

    def load_init(self, init_path: str, device: torch.device = torch.device('cpu')) -> None:
        """Same file conventions as vector_field_network.py:109-138."""
        emb = "embedding" if self.config.embedder_multires > 0 else "no_embedding"
        if self.config.init == "center":
            path = f'exps_vf_nerf/point_to_center/{emb}.pth'
        elif self.config.init == "exterior":
            path = f'exps_vf_nerf/point_exterior/{emb}.pth'
        elif "exterior" in self.config.init:
            path = init_path
        else:
            raise ValueError(f"load_init called with init={self.config.init!r}")
        self.load_state_dict(torch.load(path, map_location=torch.device('cpu')))
        self.to(device)

    def forward(self, points: torch.Tensor, vector_only: bool = False, jacobian: bool = True) -> torch.Tensor:
        """points[M,3] -> [M, 3+F] (tanh'ed).  ``vector_only`` returns just the 3 vector columns and
        skips the feature block of the last Linear (grid queries / proposal pass).
        In training mode (vector_field_network.py:146-173): batch-statistics BatchNorm and nine more columns, the three
        ``autograd.grad`` rows -> [M, 3+F+9]; the reference marks ``points`` as requiring grad in place, which is not done
        here.  ``jacobian=False`` skips those columns (callers that drop them)."""
        if self.training:           # like the reference: the Jacobian columns belong to training mode, BatchNorm or not
            from .batchstat import vf_forward_train_mode
            out = vf_forward_train_mode(self, points, want_jacobian=jacobian and not vector_only)
            return out[:, :3].contiguous() if vector_only else out
        if points.is_cuda and torch.is_grad_enabled():
            # points a sampler wrote into the open training step's supervision rows (stepengine.StepSession): the vector-only saving forward on
            # those rows; [:, :3] of the result is all the trainer reads (train/vector_field_nerf_train.py:201,213)
            from .stepengine import session_vf_forward
            out = session_vf_forward(self, points)
            if out is not None:
                return out[:, :3] if vector_only else out
        from .autograd import vf_forward  # local import: autograd wrappers depend on this module
        return vf_forward(self, points, vector_only)

    def get_outputs(self, input_tensor: torch.Tensor):
        out = self.forward(input_tensor)
        return out[:, :3], out[:, 3:]


class RenderingNetwork(_PackedMLP):
    _kind = lib.NET_RENDER

    def __init__(self, config) -> None:
        super().__init__()
        self.config = config
        self.mode = config.mode
        if self.mode != "idr":
            raise NotImplementedError(f"rendering mode {self.mode!r}: only 'idr' (the shipped conf) is on the HIP path")
        pe = _pe_dim(config.embedder_multires, 3)
        in0 = 3 + pe + 3 + config.feature_vector_dims
        dims = [in0] + list(config.dimensions) + [config.output_dims]
        self._build_layers(dims[:-1], dims[1:], config.batch_norm, config.weight_norm)

    def _multires(self) -> int:
        return int(self.config.embedder_multires)

    def _skip_layer(self) -> int:
        return -1

    def _feature_dims(self) -> int:
        return int(self.config.feature_vector_dims)

    def forward(self, points: torch.Tensor, normals: torch.Tensor, view_dirs: torch.Tensor,
                feature_vectors: torch.Tensor) -> torch.Tensor:
        if self.config.detach_normals:
            normals = normals.detach()
        if self._batch_statistics():
            from .batchstat import render_forward_train_mode
            return render_forward_train_mode(self, points, normals, view_dirs, feature_vectors)
        from .autograd import render_forward
        return render_forward(self, points, normals, view_dirs, feature_vectors)
