"""ctypes binding of ``libvfn.so`` (the C ABI declared in ``include/vfn.h``).

There is no CPU fallback: if the shared library is missing, or a tensor is not a contiguous fp32
CUDA(HIP) tensor, the call raises.  ``import torch`` must come first so that the HIP runtime the
library binds to is the one PyTorch already loaded (same ``libamdhip64.so.7`` SONAME).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VFN_LIB") or os.path.join(_HERE, "csrc", "libvfn.so")     # VFN_LIB: an experimental build (tools/)

MAX_LAYERS = 16
HIDDEN = 256
NET_VF, NET_RENDER = 0, 1

EXPORTS = (
    "vfn_last_error", "vfn_abi_version", "vfn_abi_struct_bytes", "vfn_packed_size", "vfn_pack_weights", "vfn_raygen_uniform",
    "vfn_vf_mlp_fwd", "vfn_render_mlp_fwd", "vfn_vf_render_fused_fwd", "vfn_ray_density_weights",
    "vfn_range_fine_sample", "vfn_range_fine_sample_indexed", "vfn_fill_uniform", "vfn_sample_sphere_shell", "vfn_packed_bwd_size", "vfn_pack_weights_bwd",
    "vfn_vf_mlp_fwd_train", "vfn_vf_render_fused_fwd_train", "vfn_mlp_bwd_chain", "vfn_weight_grad_partials",
    "vfn_ray_density_weights_bwd", "vfn_pack16_size", "vfn_pack16_weights", "vfn_vf_mlp16_fwd",
    "vfn_vf_render_fused16_fwd", "vfn_vf_mlp16_fwd_train", "vfn_vf_render_fused16_fwd_train",
    "vfn_vf_feat16_fwd", "vfn_render16_from_blocks", "vfn_grid_divergence", "vfn_grid_smooth_axis",
    "vfn_grid_unify_direction", "vfn_grid_comb_format", "vfn_grid_unify_direction_sides", "vfn_grid_comb_format_sides", "vfn_weight_grad_partials_bf16", "vfn_unfold_weight_grads", "vfn_packed_bwd16_size", "vfn_pack_weights_bwd16", "vfn_pack_weights_bwd16_mode", "vfn_mlp_bwd_chain_bf16",
    "vfn_linear_rows", "vfn_linear_rows_stat_parts", "vfn_bstat_row_parts", "vfn_colsum_finish", "vfn_bstat_finalize",
    "vfn_bstat_relu_rows", "vfn_bstat_relu_bwd_sums", "vfn_bstat_relu_bwd_rows", "vfn_act_bwd_rows", "vfn_embed_rows",
    "vfn_embed_rows_bwd", "vfn_vf_render_fused16_scatter", "vfn_vf_render_fused16_products", "vfn_net_weight_grads_frag", "vfn_net_weight_grads_frag_part", "vfn_net_weight_grads_scratch_bytes", "vfn_vf_render_fused16_fwd_train_at",
    "vfn_vf_mlp16_fwd_train_at", "vfn_mlp_bwd_chain_bf16_ws_at",
    "vfn_weight_grad_groups", "vfn_scatter_rows3", "vfn_uniform_sample", "vfn_rows_argmax", "vfn_merge_sort_depths",
    "vfn_ray_density_sigma_bwd", "vfn_weight_grad_frag", "vfn_mlp_bwd_chain_bf16_ws", "vfn_f16x3_set_status", "vfn_f16x3_set_clock_probe", "vfn_flat_clip_workspace_bytes", "vfn_flat_clip_grad_norm", "vfn_flat_adam_step", "vfn_unfold_weight_grads_acc", "vfn_render_fwd", "vfn_render_fwd_workspace_bytes",
    "vfn_vf_loss_workspace_bytes", "vfn_vf_loss_fwd", "vfn_vf_loss_bwd", "vfn_train_step", "vfn_train_step_workspace_bytes",
    "vfn_train_step_workspace_layout", "vfn_train_step_supervision_points", "vfn_train_step_supervision_forward", "vfn_train_step_supervision_backward",
    "vfn_linear_rows_dx_sums", "vfn_weight_grad_partials_bf16_ld", "vfn_linear_rows_ws", "vfn_linear_rows_wplanes_bytes",
    "vfn_select_samples", "vfn_grid_lattice_points", "vfn_linear_rows_fold", "vfn_weight_grad_partials_bf16_fold",
)


class VfnError(RuntimeError):
    pass


class NetGeom(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("multires", C.c_int32), ("skip_layer", C.c_int32),
                ("feature_dims", C.c_int32), ("in_dims", C.c_int32 * MAX_LAYERS),
                ("out_dims", C.c_int32 * MAX_LAYERS), ("has_bn", C.c_int32 * MAX_LAYERS)]


class LayerParams(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("bn_weight", C.c_void_p),
                ("bn_bias", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_var", C.c_void_p)]


class RaygenParams(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("pose_is_quat", C.c_int32),
                ("near", C.c_float), ("far", C.c_float)]


class DensityParams(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("n_window", C.c_int32),
                ("normalize", C.c_int32), ("dir_to_normal_th", C.c_float),
                ("beta_min", C.c_float), ("beta_max", C.c_float), ("mean_min", C.c_float),
                ("mean_max", C.c_float), ("scale_min", C.c_float), ("cutoff", C.c_float)]


class FineParams(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_coarse", C.c_int32), ("n_fine", C.c_int32),
                ("near", C.c_float), ("far", C.c_float), ("half_range", C.c_float),
                ("window_step", C.c_float), ("span", C.c_float)]


class RenderParams(C.Structure):
    """mirrors vfn_render_params"""
    _fields_ = [("n_rays", C.c_int32), ("n_coarse", C.c_int32), ("n_fine", C.c_int32), ("pose_is_quat", C.c_int32),
                ("perturb_coarse", C.c_int32), ("perturb_fine", C.c_int32), ("near_coarse", C.c_float), ("near_fine", C.c_float),
                ("far_coarse", C.c_float), ("far_fine", C.c_float), ("fine_range", C.c_float), ("window_step", C.c_float), ("span", C.c_float),
                ("density", DensityParams), ("seed", C.c_uint64), ("offset", C.c_uint64), ("colour_products", C.c_int32), ("separate_launches", C.c_int32), ("streams", C.c_int32), ("sparse_colours", C.c_int32),
                ("timing_events", C.c_void_p * 4), ("status_word", C.c_void_p), ("clock_stamps", C.c_void_p), ("clock_slots", C.c_int64)]


class LossParams(C.Structure):
    """mirrors vfn_loss_params"""
    _fields_ = [("n_rays", C.c_int64), ("n_normals", C.c_int64), ("n_sup", C.c_int64 * 3), ("has_depth", C.c_int32), ("smaller_on", C.c_int32),
                ("ray_center", C.c_int32), ("reserved", C.c_int32), ("w_rgb", C.c_float), ("w_depth", C.c_float), ("w_unit", C.c_float),
                ("w_sup", C.c_float), ("w_smaller", C.c_float), ("depth_clamp", C.c_float), ("radius", C.c_float), ("centroid", C.c_float * 3)]


class TrainStepParams(C.Structure):
    """mirrors vfn_train_step_params"""
    _fields_ = [("render", RenderParams), ("loss", LossParams), ("n_sup", C.c_int64), ("border", C.c_int32), ("center", C.c_int32),
                ("sup_centroid", C.c_float * 3), ("sup_radius", C.c_float), ("border_r_min", C.c_float), ("border_r_max", C.c_float), ("phases", C.c_int32),
                ("sup_seed", C.c_uint64), ("sup_offset", C.c_uint64), ("save_flags", C.c_int32), ("dy_flags", C.c_int32), ("dy_form", C.c_int32),
                ("x_form", C.c_int32), ("forward_products", C.c_int32), ("repack", C.c_int32), ("n_regions", C.c_int32), ("mults", C.c_int32 * 4),
                ("starts", C.c_int64 * 4), ("ends", C.c_int64 * 4), ("step_size", C.c_double * 8), ("bc2_sqrt", C.c_double * 8),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("weight_decay", C.c_double), ("max_norm", C.c_float),
                ("sparse_colours", C.c_int32), ("sup_rows_reserved", C.c_int64), ("sup_rows_used", C.c_int64)]


class TrainStepIO(C.Structure):
    """mirrors vfn_train_step_io (every member a pointer except n_flat)"""
    _fields_ = [(n, C.c_void_p) for n in ("vf_geom", "rn_geom", "vf_packed16", "rn_packed16", "vf_packed_bwd16", "rn_packed_bwd16", "vf_layers", "rn_layers",
                                          "vf_wgrad", "rn_wgrad", "vf_head_w", "rn_head_w", "beta", "mean", "scale", "g_beta", "g_mean", "g_scale",
                                          "flat_param", "flat_grad", "exp_avg", "exp_avg_sq")] + [("n_flat", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("clip_workspace", "uv", "pose", "intrinsics", "t_vals", "far_coarse_per_ray", "far_fine_per_ray", "u_coarse", "u_fine",
                                          "u_add", "sup_u_border", "sup_u_center", "rgb_gt", "depth_gt", "workspace", "ray_dirs", "z_vals", "points", "normals",
                                          "colors", "weights", "rgb", "depth", "out_terms", "out_norm", "out_counts", "d_rgb_in", "d_depth_in", "d_normals_in")]


TRAIN_FORWARD_BACKWARD, TRAIN_OPTIMIZER, TRAIN_RENDER, TRAIN_BACKWARD, TRAIN_CLIP, TRAIN_ADAM = 1, 2, 4, 8, 16, 32
# indices of vfn_train_step_workspace_layout's output (VFN_TWS_* of include/vfn.h)
TWS_SUP_PTS, TWS_SUP_GT, TWS_SUP_PRED, TWS_D_SUP, TWS_DN, TWS_SUP_ROWS, TWS_TOTAL_ROWS, TWS_BYTES, TWS_COUNT = range(9)

_lib: Optional[C.CDLL] = None

def _find_header() -> str:
    """include/vfn.h of the repository, or the copy the build leaves inside the package (``vf_nerf_amd/vfn_abi.h``, written by
    csrc/build.sh) for a package that was copied / installed without the repository around it; ``VFN_HEADER`` overrides both."""
    for cand in (os.environ.get("VFN_HEADER"), os.path.join(os.path.dirname(_HERE), "include", "vfn.h"), os.path.join(_HERE, "vfn_abi.h")):
        if cand and os.path.exists(cand):
            return cand
    raise VfnError(f"include/vfn.h not found next to the package ({os.path.dirname(_HERE)}/include) nor as {_HERE}/vfn_abi.h: the binding's "
                   "signatures are generated from it; set VFN_HEADER or rebuild with vf_nerf_amd/csrc/build.sh")


_header_cache: dict = {}


def _header_raw() -> str:
    if "raw" not in _header_cache:
        path = _find_header()
        with open(path) as fh:
            _header_cache["raw"], _header_cache["path"] = fh.read(), path
    return _header_cache["raw"]


def _header_text() -> str:
    import re
    text = re.sub(r"/\*.*?\*/", "", _header_raw(), flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def header_abi_version() -> int:
    import re
    return int(re.search(r"#define\s+VFN_ABI_VERSION\s+(\d+)", _header_raw()).group(1))


def __getattr__(name: str):
    # read on first use, not at import: the CPU-only parts of the package (loss, optimizer, host logic) import this module too
    if name == "ABI_VERSION":            # include/vfn.h is the single source: the library must report the same number
        return header_abi_version()
    if name == "HEADER_PATH":
        _header_raw()
        return _header_cache["path"]
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def header_prototypes() -> dict:
    """{name: (return type, [parameter types])} of every ``vfn_*`` entry point declared in include/vfn.h, as C type strings
    without the parameter names (``"const float*"``, ``"int64_t"``, ...)."""
    import re
    out = {}
    for ret, name, params in re.findall(r"^\s*([A-Za-z_][\w\s\*]*?)\b(vfn_\w+)\s*\(([^;{}]*?)\)\s*;", _header_text(), flags=re.M | re.S):
        plist = []
        for prm in params.replace("\n", " ").split(","):
            prm = re.sub(r"\s+", " ", prm).strip()
            if prm in ("void", ""):
                continue
            if "[" in prm or "(" in prm:        # array parameters / function pointers: not a form this generator understands
                raise VfnError(f"{name}: parameter '{prm}' of include/vfn.h is an array or a function pointer; the binding generator "
                               "handles scalars and plain pointers only")
            plist.append(re.match(r"(.*?)\s*\w+$", prm).group(1).replace(" *", "*").strip())
        out[name] = (re.sub(r"\s+", " ", ret).strip(), plist)
    return out


_INT_TYPES = {"int": C.c_int, "int32_t": C.c_int32, "int64_t": C.c_int64, "uint32_t": C.c_uint32, "uint64_t": C.c_uint64}
_FLOAT_TYPES = {"float": C.c_float, "double": C.c_double}
_C_INTS = (C.c_int8, C.c_int16, C.c_int32, C.c_int64, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_bool)


class _IntArg:
    """argtypes entry of an integer parameter: Python ints and ctypes integers of any width pass (converted to the declared
    width); floats, pointers and None do not — a call whose arguments were reordered fails here instead of reaching the kernel."""

    def __init__(self, ctype, what):
        self.ctype, self.what = ctype, what

    def from_param(self, v):
        if isinstance(v, _C_INTS):
            v = v.value
        if isinstance(v, bool):
            v = int(v)
        if not isinstance(v, int):
            raise TypeError(f"{self.what}: expected an integer, got {type(v).__name__}")
        return self.ctype(v)


class _FloatArg:
    def __init__(self, ctype, what):
        self.ctype, self.what = ctype, what

    def from_param(self, v):
        if isinstance(v, (C.c_float, C.c_double)):
            v = v.value
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise TypeError(f"{self.what}: expected a real number, got {type(v).__name__}")
        return self.ctype(float(v))


class _PtrArg:
    """argtypes entry of a pointer parameter: None, c_void_p, byref()/pointer() results and ctypes arrays / structures pass."""

    def __init__(self, what):
        self.what = what

    def from_param(self, v):
        if v is None:
            return C.c_void_p(0)
        if isinstance(v, (C.c_void_p, C.c_char_p, C.Array, C._Pointer)) or type(v).__name__ == "CArgObject":
            return v
        if isinstance(v, C.Structure):
            return C.byref(v)
        raise TypeError(f"{self.what}: expected a pointer (None, c_void_p, byref(struct), ctypes array), got {type(v).__name__}")


def _declare(lib: C.CDLL) -> None:
    """restype / argtypes of EVERY export, generated from the prototypes of include/vfn.h (the header is the single source; a
    signature edited there reaches the binding without a second hand-written list)."""
    protos = header_prototypes()
    for name in EXPORTS:
        fn = getattr(lib, name)               # raises AttributeError if the ABI lost a symbol
        if name not in protos:
            raise VfnError(f"{name} is exported by the binding but not declared in {_header_cache['path']}")
        ret, params = protos[name]
        fn.restype = C.c_char_p if ret == "const char*" else _INT_TYPES[ret]
        args = []
        for i, t in enumerate(params):
            what = f"{name} argument {i} ({t})"
            args.append(_PtrArg(what) if t.endswith("*") else
                        _IntArg(_INT_TYPES[t], what) if t in _INT_TYPES else _FloatArg(_FLOAT_TYPES[t], what))
        fn.argtypes = args


def struct_mirrors():
    """The ctypes mirrors of the header's POD structs in the order of ``vfn_abi_struct_bytes``."""
    return (NetGeom, LayerParams, RaygenParams, DensityParams, FineParams, RenderParams, UnfoldEntry, WgradLayer, LossParams, TrainStepParams,
            TrainStepIO)


def load() -> C.CDLL:
    """Load libvfn.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VfnError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or vf_nerf_amd/csrc/build.sh (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    _declare(lib)
    if lib.vfn_abi_version() != header_abi_version():
        raise VfnError(f"libvfn.so reports ABI {lib.vfn_abi_version()}, include/vfn.h says {header_abi_version()}: rebuild the library")
    for i, mirror in enumerate(struct_mirrors()):
        if lib.vfn_abi_struct_bytes(i) != C.sizeof(mirror):
            raise VfnError(f"{mirror.__name__}: {C.sizeof(mirror)} bytes in the binding, {lib.vfn_abi_struct_bytes(i)} in libvfn.so")
    _lib = lib
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise VfnError(f"{what} failed (status {rc}): {load().vfn_last_error().decode()}")


def _ptr(t: Optional[torch.Tensor], name: str, dtype=torch.float32) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise VfnError(f"{name}: expected a CUDA/HIP tensor, got {t.device} (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise VfnError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise VfnError(f"{name}: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def _stream() -> C.c_void_p:
    # (the raw handle of the current stream of the current device: torch.cuda.current_stream() builds a Stream object around it, 13 us a call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


# ------------------------------------------------------------------------------------------------
# geometry / packing
# ------------------------------------------------------------------------------------------------
def make_geom(n_layers: int, multires: int, skip_layer: int, feature_dims: int, in_dims: Sequence[int],
              out_dims: Sequence[int], has_bn: Sequence[bool]) -> NetGeom:
    if n_layers > MAX_LAYERS:
        raise VfnError(f"{n_layers} layers > {MAX_LAYERS}")
    g = NetGeom()
    g.n_layers, g.multires, g.skip_layer, g.feature_dims = n_layers, multires, skip_layer, feature_dims
    for i in range(n_layers):
        g.in_dims[i], g.out_dims[i], g.has_bn[i] = int(in_dims[i]), int(out_dims[i]), int(bool(has_bn[i]))
    return g


def packed_size(kind: int, geom: NetGeom) -> int:
    n = load().vfn_packed_size(kind, C.byref(geom))
    if n < 0:
        raise VfnError(f"unsupported network geometry: {load().vfn_last_error().decode()}")
    return int(n)


def _layer_array(geom: NetGeom, layers: Sequence[dict]):
    arr = (LayerParams * geom.n_layers)()
    for i, lp in enumerate(layers):
        arr[i].weight = _ptr(lp["weight"], f"layer{i}.weight")
        arr[i].bias = _ptr(lp["bias"], f"layer{i}.bias")
        for k in ("bn_weight", "bn_bias", "bn_mean", "bn_var"):
            setattr(arr[i], k, _ptr(lp.get(k), f"layer{i}.{k}"))
    return arr


def pack_weights(kind: int, geom: NetGeom, layers: Sequence[dict], packed: torch.Tensor) -> None:
    """layers[i] = dict(weight=, bias=, bn_weight=, bn_bias=, bn_mean=, bn_var=) of live tensors."""
    _check(load().vfn_pack_weights(kind, C.byref(geom), _layer_array(geom, layers), _ptr(packed, "packed"), _stream()),
           "vfn_pack_weights")


def packed_bwd_size(kind: int, geom: NetGeom) -> int:
    n = load().vfn_packed_bwd_size(kind, C.byref(geom))
    if n < 0:
        raise VfnError(f"unsupported network geometry: {load().vfn_last_error().decode()}")
    return int(n)


def pack_weights_bwd(kind: int, geom: NetGeom, layers: Sequence[dict], packed_bwd: torch.Tensor) -> None:
    _check(load().vfn_pack_weights_bwd(kind, C.byref(geom), _layer_array(geom, layers), _ptr(packed_bwd, "packed_bwd"),
                                       _stream()), "vfn_pack_weights_bwd")


# ------------------------------------------------------------------------------------------------
# kernels
# ------------------------------------------------------------------------------------------------
def raygen_uniform(uv, pose, intrinsics, t_vals, n_samples, near, far, far_per_ray=None, u_coarse=None):
    n = uv.shape[0]
    dev = uv.device
    p = RaygenParams(n, n_samples, int(pose.dim() == 2 and pose.shape[1] == 7), float(near), float(far))
    directions = torch.empty(n, 3, device=dev)
    ray_dirs = torch.empty(n, 3, device=dev)
    cam_loc = torch.empty(n, 3, device=dev)
    z = torch.empty(n, n_samples, device=dev)
    pts = torch.empty(n, n_samples, 3, device=dev)
    _check(load().vfn_raygen_uniform(C.byref(p), _ptr(uv, "uv"), _ptr(pose, "pose"), _ptr(intrinsics, "intrinsics"),
                                     _ptr(t_vals, "t_vals"), _ptr(far_per_ray, "far_per_ray"),
                                     _ptr(u_coarse, "u_coarse"), _ptr(directions, "directions"),
                                     _ptr(ray_dirs, "ray_dirs"), _ptr(cam_loc, "cam_loc"), _ptr(z, "z"),
                                     _ptr(pts, "points"), _stream()), "vfn_raygen_uniform")
    return directions, ray_dirs, cam_loc, z, pts


def uniform_sample(directions, cam_loc, t_vals, n_samples: int, near: float, far: float, far_per_ray=None, u=None,
                   want_points: bool = True):
    """UniformSampler.get_z_vals / RaySampler.sample on given directions[N,3] / cam_loc[N,3] -> (z[N,S], points[N,S,3] | None)."""
    n = directions.shape[0]
    dev = directions.device
    z = torch.empty(n, n_samples, device=dev)
    pts = torch.empty(n, n_samples, 3, device=dev) if want_points else None
    _check(load().vfn_uniform_sample(C.c_int32(n), C.c_int32(n_samples), C.c_float(near), C.c_float(far),
                                     _ptr(directions, "directions"), _ptr(cam_loc, "cam_loc"), _ptr(t_vals, "t_vals"),
                                     _ptr(far_per_ray, "far_per_ray"), _ptr(u, "u"), _ptr(z, "z_vals"), _ptr(pts, "points"),
                                     _stream()), "vfn_uniform_sample")
    return z, pts


def _ptr3(tensors, name: str):
    arr = (C.c_void_p * 3)()
    for k in range(3):
        t = tensors[k] if k < len(tensors) else None
        arr[k] = _ptr(t, f"{name}[{k}]").value if t is not None else None
    return arr


def vf_loss_workspace(dev) -> torch.Tensor:
    return torch.empty(int(load().vfn_vf_loss_workspace_bytes()) // 4, device=dev)


def vf_loss_fwd(lp: LossParams, rgb, rgb_gt, depth, depth_gt, normals, points, sup_pred, sup_gt, workspace) -> torch.Tensor:
    """-> out[8] on the device: five terms, 0, weighted total, supervision rows (csrc/vfn_loss.hip)."""
    out = torch.empty(8, device=workspace.device)
    _check(load().vfn_vf_loss_fwd(C.byref(lp), _ptr(rgb, "rgb"), _ptr(rgb_gt, "rgb_gt"), _ptr(depth, "depth"), _ptr(depth_gt, "depth_gt"),
                                  _ptr(normals, "normals"), _ptr(points, "points"), _ptr3(sup_pred, "sup_pred"), _ptr3(sup_gt, "sup_gt"),
                                  _ptr(workspace, "workspace"), _ptr(out, "out_terms"), _stream()), "vfn_vf_loss_fwd")
    return out


def vf_loss_bwd(lp: LossParams, rgb, rgb_gt, depth, depth_gt, normals, points, sup_pred, sup_gt, workspace, grad_out, d_rgb, d_depth,
                d_normals, d_sup) -> None:
    _check(load().vfn_vf_loss_bwd(C.byref(lp), _ptr(rgb, "rgb"), _ptr(rgb_gt, "rgb_gt"), _ptr(depth, "depth"), _ptr(depth_gt, "depth_gt"),
                                  _ptr(normals, "normals"), _ptr(points, "points"), _ptr3(sup_pred, "sup_pred"), _ptr3(sup_gt, "sup_gt"),
                                  _ptr(workspace, "workspace"), _ptr(grad_out, "grad_out"), _ptr(d_rgb, "d_rgb"), _ptr(d_depth, "d_depth"),
                                  _ptr(d_normals, "d_normals"), _ptr3(d_sup, "d_sup"), _stream()), "vfn_vf_loss_bwd")


def merge_sort_depths(z: torch.Tensor, extra: torch.Tensor, directions=None, cam_loc=None):
    """sort(cat(z[N,S], extra[N,E]), dim=1) per ray on the device (+ the points along the rays when directions / cam_loc are given)
    -> (z_out[N,S+E], points[N,S+E,3] | None)."""
    n, s = z.shape
    e = extra.shape[1]
    out = torch.empty(n, s + e, device=z.device)
    pts = torch.empty(n, s + e, 3, device=z.device) if directions is not None else None
    _check(load().vfn_merge_sort_depths(_ptr(z, "z_vals"), _ptr(extra, "extra"), C.c_int32(n), C.c_int32(s), C.c_int32(e),
                                        _ptr(directions, "directions"), _ptr(cam_loc, "cam_loc"), _ptr(out, "z_out"), _ptr(pts, "points"),
                                        _stream()), "vfn_merge_sort_depths")
    return out, pts


def rows_argmax(w: torch.Tensor) -> torch.Tensor:
    """First-maximum index of every row (torch.argmax(w, dim=-1) semantics) -> int64 [rows]."""
    rows, cols = w.shape
    out = torch.empty(rows, dtype=torch.int64, device=w.device)
    _check(load().vfn_rows_argmax(_ptr(w, "w"), C.c_int32(rows), C.c_int32(cols), _ptr(out, "out", torch.int64), _stream()),
           "vfn_rows_argmax")
    return out


def ray_density_sigma_bwd(dp: DensityParams, normals, ray_dirs, z_vals, scalars, d_sigma, d_normals, d_scalars) -> None:
    n, s = z_vals.shape
    dp.n_rays, dp.n_samples = n, s
    _check(load().vfn_ray_density_sigma_bwd(C.byref(dp), _ptr(normals, "normals"), _ptr(ray_dirs, "ray_dirs"),
                                            _ptr(z_vals, "z_vals"), _ptr(scalars, "scalars"), _ptr(d_sigma, "d_sigma"),
                                            _ptr(d_normals, "d_normals"), _ptr(d_scalars, "d_scalars"), _stream()),
           "vfn_ray_density_sigma_bwd")


def render_workspace_bytes(rp: RenderParams) -> int:
    n = load().vfn_render_fwd_workspace_bytes(C.byref(rp))
    if n < 0:
        raise VfnError("vfn_render_fwd_workspace_bytes: bad sizes")
    return int(n)


def render_fwd(rp: RenderParams, vf_geom, vf_packed16, rn_geom, rn_packed16, uv, pose, intrinsics, t_vals, far_c, far_f, scalars,
               u_coarse, u_fine, u_add, workspace):
    """The whole gradient-free render() in one call (csrc/vfn_render.hip) -> dict of output tensors."""
    n, s_t = rp.n_rays, rp.n_coarse + rp.n_fine
    dev = uv.device
    out = dict(ray_dirs=torch.empty(n, 3, device=dev), z_vals=torch.empty(n, s_t, device=dev), points=torch.empty(n, s_t, 3, device=dev),
               normals=torch.empty(n * s_t, 3, device=dev), colors=torch.empty(n * s_t, 3, device=dev),
               weights=torch.empty(n, s_t, device=dev), rgb=torch.empty(n, 3, device=dev), depth=torch.empty(n, 1, device=dev))
    _check(load().vfn_render_fwd(C.byref(rp), C.byref(vf_geom), _ptr(vf_packed16, "vf_packed16", torch.uint8), C.byref(rn_geom),
                                 _ptr(rn_packed16, "rn_packed16", torch.uint8), _ptr(uv, "uv"), _ptr(pose, "pose"),
                                 _ptr(intrinsics, "intrinsics"), _ptr(t_vals, "t_vals"), _ptr(far_c, "far_coarse_per_ray"),
                                 _ptr(far_f, "far_fine_per_ray"), _ptr(scalars, "density_scalars"), _ptr(u_coarse, "u_coarse"),
                                 _ptr(u_fine, "u_fine"), _ptr(u_add, "u_add"), _ptr(workspace, "workspace", torch.uint8),
                                 _ptr(out["ray_dirs"], "ray_dirs"), _ptr(out["z_vals"], "z_vals"), _ptr(out["points"], "points"),
                                 _ptr(out["normals"], "normals"), _ptr(out["colors"], "colors"), _ptr(out["weights"], "weights"),
                                 _ptr(out["rgb"], "rgb"), _ptr(out["depth"], "depth"), _stream()), "vfn_render_fwd")
    return out


def vf_mlp_fwd(geom: NetGeom, packed, points, out_cols: int):
    m = points.shape[0]
    out = torch.empty(m, out_cols, device=points.device)
    _check(load().vfn_vf_mlp_fwd(C.byref(geom), _ptr(packed, "packed"), _ptr(points, "points"), C.c_int64(m),
                                 C.c_int32(out_cols), _ptr(out, "out"), _stream()), "vfn_vf_mlp_fwd")
    return out


def render_mlp_fwd(geom: NetGeom, packed, points, normals, view_dirs, feats):
    m = points.shape[0]
    colors = torch.empty(m, 3, device=points.device)
    _check(load().vfn_render_mlp_fwd(C.byref(geom), _ptr(packed, "packed"), _ptr(points, "points"),
                                     _ptr(normals, "normals"), _ptr(view_dirs, "view_dirs"), _ptr(feats, "feats"),
                                     C.c_int64(m), _ptr(colors, "colors"), _stream()), "vfn_render_mlp_fwd")
    return colors


def vf_render_fused_fwd(vf_geom, vf_packed, rn_geom, rn_packed, points, ray_dirs, samples_per_ray: int,
                        want_feats: bool = False):
    m = points.shape[0]
    dev = points.device
    normals = torch.empty(m, 3, device=dev)
    colors = torch.empty(m, 3, device=dev)
    feats = torch.empty(m, HIDDEN, device=dev) if want_feats else None
    _check(load().vfn_vf_render_fused_fwd(C.byref(vf_geom), _ptr(vf_packed, "vf_packed"), C.byref(rn_geom),
                                          _ptr(rn_packed, "rn_packed"), _ptr(points, "points"),
                                          _ptr(ray_dirs, "ray_dirs"), C.c_int64(m), C.c_int32(samples_per_ray),
                                          _ptr(normals, "normals"), _ptr(colors, "colors"), _ptr(feats, "feats"),
                                          _stream()), "vfn_vf_render_fused_fwd")
    return normals, colors, feats


def ray_density_weights(dp: DensityParams, normals, ray_dirs, z_vals, scalars, colors=None, want_sigma=True,
                        want_weights=True, want_argmax=False):
    n, s = z_vals.shape
    dev = z_vals.device
    dp.n_rays, dp.n_samples = n, s
    sigma = torch.empty(n, s, device=dev) if want_sigma else None
    weights = torch.empty(n, s, device=dev) if want_weights else None
    argmax = torch.empty(n, dtype=torch.int64, device=dev) if want_argmax else None
    rgb = torch.empty(n, 3, device=dev) if colors is not None else None
    depth = torch.empty(n, 1, device=dev) if colors is not None else None
    _check(load().vfn_ray_density_weights(C.byref(dp), _ptr(normals, "normals"), _ptr(ray_dirs, "ray_dirs"),
                                          _ptr(z_vals, "z_vals"), _ptr(scalars, "density_scalars"),
                                          _ptr(colors, "colors"), _ptr(sigma, "sigma"), _ptr(weights, "weights"),
                                          _ptr(argmax, "argmax", torch.int64), _ptr(rgb, "rgb"),
                                          _ptr(depth, "depth"), _stream()), "vfn_ray_density_weights")
    return sigma, weights, argmax, rgb, depth


def range_fine_sample(z_coarse, argmax, directions, cam_loc, n_fine, near, far, fine_range, u_add, u_fine=None,
                      far_per_ray=None):
    n, sc = z_coarse.shape
    dev = z_coarse.device
    step = 2 * fine_range / (n_fine - 1)            # Python double arithmetic, as ray_sampler.py:279
    span = (far - near) if far_per_ray is None else 0.0
    p = FineParams(n, sc, n_fine, float(near), float(far) if far_per_ray is None else 0.0, float(fine_range),
                   float(step), float(span))
    z = torch.empty(n, sc + n_fine, device=dev)
    pts = torch.empty(n, sc + n_fine, 3, device=dev)
    _check(load().vfn_range_fine_sample(C.byref(p), _ptr(z_coarse, "z_coarse"), _ptr(argmax, "argmax", torch.int64),
                                        _ptr(directions, "directions"), _ptr(cam_loc, "cam_loc"),
                                        _ptr(far_per_ray, "far_per_ray"), _ptr(u_fine, "u_fine"), _ptr(u_add, "u_add"),
                                        _ptr(z, "z_vals"), _ptr(pts, "points"), _stream()), "vfn_range_fine_sample")
    return z, pts


def block_rows(m: int) -> int:
    """Rows a block buffer needs for m stored rows: whole groups of 32 (vfn_vf_feat16_fwd)."""
    return (m + 31) & ~31


def range_fine_sample_indexed(z_coarse, argmax, directions, cam_loc, n_fine, near, far, fine_range, u_add, u_fine=None,
                              far_per_ray=None, new_row0: Optional[int] = None, want_dst: bool = False):
    """range_fine_sample plus src[N,S_t] (int32: the stored row every sorted sample comes from; the new samples' rows start
    at ``new_row0``, default N*S_c) and new_points[N,N_f,3]; with ``want_dst`` also the inverse map dst[new_row0 + N*N_f]
    (int32, -1 for rows no sample comes from)."""
    n, sc = z_coarse.shape
    dev = z_coarse.device
    step = 2 * fine_range / (n_fine - 1)
    span = (far - near) if far_per_ray is None else 0.0
    p = FineParams(n, sc, n_fine, float(near), float(far) if far_per_ray is None else 0.0, float(fine_range),
                   float(step), float(span))
    z = torch.empty(n, sc + n_fine, device=dev)
    pts = torch.empty(n, sc + n_fine, 3, device=dev)
    src = torch.empty(n, sc + n_fine, dtype=torch.int32, device=dev)
    new_pts = torch.empty(n, n_fine, 3, device=dev)
    row0 = n * sc if new_row0 is None else int(new_row0)
    dst = None
    if want_dst:
        rows = row0 + n * n_fine
        dst = torch.empty(rows, dtype=torch.int32, device=dev) if row0 == n * sc else \
            torch.full((rows,), -1, dtype=torch.int32, device=dev)
    _check(load().vfn_range_fine_sample_indexed(C.byref(p), _ptr(z_coarse, "z_coarse"), _ptr(argmax, "argmax", torch.int64),
                                                _ptr(directions, "directions"), _ptr(cam_loc, "cam_loc"),
                                                _ptr(far_per_ray, "far_per_ray"), _ptr(u_fine, "u_fine"),
                                                _ptr(u_add, "u_add"), _ptr(z, "z_vals"), _ptr(pts, "points"),
                                                _ptr(src, "src", torch.int32), _ptr(new_pts, "new_points"),
                                                _ptr(dst, "dst", torch.int32), C.c_int64(row0), _stream()),
           "vfn_range_fine_sample_indexed")
    return (z, pts, src, new_pts, dst) if want_dst else (z, pts, src, new_pts)


def fill_uniform(out: torch.Tensor, seed: int, offset: int) -> torch.Tensor:
    _check(load().vfn_fill_uniform(_ptr(out, "out"), C.c_int64(out.numel()), C.c_uint64(seed & (2 ** 64 - 1)),
                                   C.c_uint64(offset & (2 ** 64 - 1)), _stream()), "vfn_fill_uniform")
    return out


def sample_sphere_shell(n: int, r_min: float, r_max: float, centroid: torch.Tensor, inward: bool, seed: int = 0, offset: int = 0,
                        u: Optional[torch.Tensor] = None):
    dev = centroid.device
    pts = torch.empty(n, 3, device=dev)
    gt = torch.empty(n, 3, device=dev)
    _check(load().vfn_sample_sphere_shell(C.c_int64(n), C.c_float(r_min), C.c_float(r_max), _ptr(centroid, "centroid"),
                                          C.c_int32(1 if inward else 0), _ptr(u, "u"), C.c_uint64(seed & (2 ** 64 - 1)),
                                          C.c_uint64(offset & (2 ** 64 - 1)), _ptr(pts, "points"), _ptr(gt, "gt"), _stream()),
           "vfn_sample_sphere_shell")
    return pts, gt


# ------------------------------------------------------------------------------------------------
# training path
# ------------------------------------------------------------------------------------------------
AUX_K = 40


def vf_mlp_fwd_train(geom: NetGeom, packed, points, out_cols: int, saved, aux_vf):
    m = points.shape[0]
    out = torch.empty(m, out_cols, device=points.device)
    _check(load().vfn_vf_mlp_fwd_train(C.byref(geom), _ptr(packed, "packed"), _ptr(points, "points"), C.c_int64(m),
                                       C.c_int32(out_cols), _ptr(out, "out"), _ptr(saved, "saved"),
                                       _ptr(aux_vf, "aux_vf"), _stream()), "vfn_vf_mlp_fwd_train")
    return out


def vf_render_fused_fwd_train(vf_geom, vf_packed, rn_geom, rn_packed, points, ray_dirs, samples_per_ray, saved, aux_vf,
                              aux_rn):
    m = points.shape[0]
    dev = points.device
    normals = torch.empty(m, 3, device=dev)
    colors = torch.empty(m, 3, device=dev)
    _check(load().vfn_vf_render_fused_fwd_train(C.byref(vf_geom), _ptr(vf_packed, "vf_packed"), C.byref(rn_geom),
                                                _ptr(rn_packed, "rn_packed"), _ptr(points, "points"),
                                                _ptr(ray_dirs, "ray_dirs"), C.c_int64(m), C.c_int32(samples_per_ray),
                                                _ptr(normals, "normals"), _ptr(colors, "colors"), _ptr(saved, "saved"),
                                                _ptr(aux_vf, "aux_vf"), _ptr(aux_rn, "aux_rn"), _stream()),
           "vfn_vf_render_fused_fwd_train")
    return normals, colors


def mlp_bwd_chain(vf_geom, vf_packed, vf_packed_bwd, rn_geom, rn_packed, rn_packed_bwd, saved, dy, d_colors, colors,
                  d_vec, vec, d_feats, vec_stride: int, n_points: int, dz_rgb, dz_vec):
    rn = C.byref(rn_geom) if rn_geom is not None else None
    _check(load().vfn_mlp_bwd_chain(C.byref(vf_geom), _ptr(vf_packed, "vf_packed"), _ptr(vf_packed_bwd, "vf_packed_bwd"),
                                    rn, _ptr(rn_packed, "rn_packed"), _ptr(rn_packed_bwd, "rn_packed_bwd"),
                                    _ptr(saved, "saved"), _ptr(dy, "dy"), _ptr(d_colors, "d_colors"),
                                    _ptr(colors, "colors"), _ptr(d_vec, "d_vec"), _ptr(vec, "vec"),
                                    _ptr(d_feats, "d_feats"), C.c_int32(vec_stride), C.c_int64(n_points),
                                    _ptr(dz_rgb, "dz_rgb"), _ptr(dz_vec, "dz_vec"), _stream()), "vfn_mlp_bwd_chain")


def weight_grad_partials(shape: int, dy, ld_dy: int, n_valid: int, x, ld_x: int, k_valid: int, n_points: int,
                         groups: int, dw_part, db_part=None, x_f16: bool = False):
    _check(load().vfn_weight_grad_partials(C.c_int32(shape), _ptr(dy, "dy"), C.c_int32(ld_dy), C.c_int32(n_valid),
                                           _ptr(x, "x"), C.c_int32(ld_x), C.c_int32(k_valid), C.c_int64(n_points),
                                           C.c_int32(groups), _ptr(dw_part, "dw_part"), _ptr(db_part, "db_part"),
                                           C.c_int32(int(x_f16)), _stream()), "vfn_weight_grad_partials")


def packed_bwd16_size(kind: int, geom: NetGeom) -> int:
    n = load().vfn_packed_bwd16_size(kind, C.byref(geom))
    if n < 0:
        raise VfnError(f"vfn_packed_bwd16_size failed (status {n}): {load().vfn_last_error().decode()}")
    return int(n)


def pack_weights_bwd16(kind: int, geom: NetGeom, layers: Sequence[dict], packed: torch.Tensor, round_hi: bool = False) -> None:
    """``round_hi``: hi planes rounded to nearest instead of truncated — the pack of the single-product chain (DY_P1)."""
    _check(load().vfn_pack_weights_bwd16_mode(kind, C.byref(geom), _layer_array(geom, layers), C.c_int32(int(bool(round_hi))),
                                              _ptr(packed, "packed", torch.uint8), _stream()), "vfn_pack_weights_bwd16")


def relu_sign_words(saved: torch.Tensor) -> torch.Tensor:
    """saved[slots, M, 256] -> the sign-bit words the f16x3 training forwards write next to it: int32 [slots, M, 2, 4] (per slot,
    point and lane half g: tile t -> half t & 1 of dword t >> 1, bit r <-> column 32 t + (r & 3) + 8 (r >> 2) + 4 g).  A torch
    restatement for callers that build a workspace by hand (tests)."""
    dev = saved.device
    r = torch.arange(16, device=dev)
    col = (32 * torch.arange(8, device=dev)[None, :, None] + (r & 3)[None, None, :] + 8 * (r >> 2)[None, None, :]
           + 4 * torch.arange(2, device=dev)[:, None, None])                     # [g, t, r]
    bits = (saved[:, :, col.reshape(-1)] > 0).view(saved.shape[0], saved.shape[1], 2, 8, 16).to(torch.int64)
    half = (bits << r).sum(-1)                                                   # [slots, M, g, t] 16-bit fields
    words = half[..., 0::2] | (half[..., 1::2] << 16)                           # [slots, M, g, 4]
    return torch.where(words >= 2 ** 31, words - 2 ** 32, words).to(torch.int32).contiguous()


def unpack_sign_words(words: torch.Tensor) -> torch.Tensor:
    """The inverse view of relu_sign_words: int32 [slots, M, 2, 4] -> bool [slots, M, 256] (True where the saved ReLU output
    was positive) — exactly the masks the bf16 chain applies."""
    dev = words.device
    w = words.to(torch.int64) & 0xffffffff
    r = torch.arange(16, device=dev)
    half = torch.stack([w & 0xffff, w >> 16], dim=-1).reshape(*words.shape[:3], 8)       # [slots, M, g, t]
    bits = ((half[..., None] >> r) & 1).bool()                                            # [slots, M, g, t, r]
    col = (32 * torch.arange(8, device=dev)[None, :, None] + (r & 3)[None, None, :] + 8 * (r >> 2)[None, None, :]
           + 4 * torch.arange(2, device=dev)[:, None, None]).reshape(-1)                  # [g, t, r] -> column
    out = torch.zeros(words.shape[0], words.shape[1], 256, dtype=torch.bool, device=dev)
    out[:, :, col] = bits.reshape(words.shape[0], words.shape[1], -1)
    return out


def mlp_bwd_chain_bf16(vf_geom, vf_packed_bwd16, vf_head_w, rn_geom, rn_packed_bwd16, rn_head_w, saved, masks, dy, d_colors, colors,
                       d_vec, vec, d_feats, vec_stride: int, n_points: int, dz_rgb, dz_vec):
    rn = C.byref(rn_geom) if rn_geom is not None else None
    _check(load().vfn_mlp_bwd_chain_bf16(C.byref(vf_geom), _ptr(vf_packed_bwd16, "vf_packed_bwd16", torch.uint8),
                                         _ptr(vf_head_w, "vf_head_w"), rn,
                                         _ptr(rn_packed_bwd16, "rn_packed_bwd16", torch.uint8), _ptr(rn_head_w, "rn_head_w"),
                                         _ptr(saved, "saved"), _ptr(masks, "masks", torch.int32), _ptr(dy, "dy"),
                                         _ptr(d_colors, "d_colors"), _ptr(colors, "colors"), _ptr(d_vec, "d_vec"), _ptr(vec, "vec"),
                                         _ptr(d_feats, "d_feats"), C.c_int32(vec_stride), C.c_int64(n_points),
                                         _ptr(dz_rgb, "dz_rgb"), _ptr(dz_vec, "dz_vec"), _stream()), "vfn_mlp_bwd_chain_bf16")


# ------------------------------------------------------------------------------------------------
# fragment-ordered training workspace (include/vfn.h, "FRAGMENT-ORDERED training workspace")
# ------------------------------------------------------------------------------------------------
WS_F16, WS_FRAG, WS_P1 = 1, 2, 4             # flags of the f16x3 training forwards (save_f16 argument); WS_P1: single-product arithmetic
DY_FRAG, DY_BF16, DY_F16S, DY_P1 = 2, 4, 8, 16   # flags of the bf16 chain (dy_flags argument): fragment order, bf16, scaled f16, single product
DYF_FRAG32, DYF_FRAGBF16, DYF_DZ4, DYF_FRAGF16S = 0, 1, 2, 3  # operand forms of weight_grad_frag (3: tile-scaled f16)
XF_FRAG32, XF_FRAG16, XF_ROWS32, XF_AUX40 = 0, 1, 2, 3
GROUP_FLOATS = 8192                          # one group of 32 points = 32 KiB


def frag_groups(m: int) -> int:
    return (m + 31) // 32


def frag_to_rows(slot: torch.Tensor, m: int, dtype=torch.float32) -> torch.Tensor:
    """One fragment-ordered slot (flat fp32 buffer of frag_groups(m) * 8192 floats) -> the row-major [m, 256] matrix it
    holds.  ``dtype``: float32, float16 (activations stored as f16) or bfloat16 (gradients stored as bf16) — the 16-bit forms
    use the first half of every group.  Test / debugging helper: the kernels never need the row-major form."""
    g = frag_groups(m)
    flat = slot.reshape(-1)[: g * GROUP_FLOATS].view(g, GROUP_FLOATS)
    if dtype != torch.float32:
        flat = flat.view(dtype)[:, :GROUP_FLOATS]               # [g, 8192] 16-bit values = the first 16 KiB of each group
    # [group][tile t][quad q][lane half gg][point i][4 values]  ->  [group][point i][t][q][gg][4]
    rows = flat.reshape(g, 8, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(g * 32, 256)
    return rows[:m].float()


def rows_to_frag(rows: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    """The inverse of frag_to_rows: [m, 256] -> one fragment-ordered slot (flat fp32 buffer, padded points zero)."""
    m = rows.shape[0]
    g = frag_groups(m)
    pad = torch.zeros(g * 32, 256, device=rows.device, dtype=torch.float32)
    pad[:m] = rows.float()
    frag = pad.reshape(g, 32, 8, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).reshape(g, GROUP_FLOATS)
    out = torch.zeros(g, GROUP_FLOATS, device=rows.device, dtype=torch.float32)
    if dtype == torch.float32:
        out.copy_(frag)
    else:
        out.view(dtype)[:, :GROUP_FLOATS] = frag.to(dtype)
    return out.reshape(-1)


F16S_EXP_OFF = 16384     # byte offset of the 8 x 64 exponent bytes inside a group of a scaled f16 gradient slot (dy form 3)


def rows_to_frag_f16s(rows: torch.Tensor) -> torch.Tensor:
    """[m, 256] fp32 -> one fragment-ordered slot in the scaled f16 form the bf16 chain writes with DY_F16S (csrc/vfn_dwf.hip,
    "dY form 3"): per group, 32-column tile t and lane (point i, column half gg: the 16 columns 8 q + 4 gg + c of the tile),
    f16(v * 2^k) with max |v| * 2^k in [2^14, 2^15) and the byte k + 113 (255: all 16 values zero) at F16S_EXP_OFF + 64 t + lane.
    Test helper (the chain is the producer)."""
    m = rows.shape[0]
    g = frag_groups(m)
    pad = torch.zeros(g * 32, 256, device=rows.device, dtype=torch.float32)
    pad[:m] = rows.float()
    v = pad.reshape(g, 32, 8, 4, 2, 4)                                       # [group][point i][tile t][quad q][half gg][c]
    mx = v.abs().amax(dim=(3, 5))                                            # [g, i, t, gg]
    biased = (mx.view(torch.int32) >> 23) & 0xff
    b = torch.where(biased == 0, torch.full_like(biased, 255), (254 - biased).clamp(0, 239))
    scale = ((b.clamp(max=239) + 14) << 23).view(torch.float32)
    scale = torch.where(b == 255, torch.zeros_like(scale), scale)
    scaled = v * scale[:, :, :, None, :, None]
    frag = scaled.permute(0, 2, 3, 4, 1, 5).reshape(g, GROUP_FLOATS)          # [group][t][q][gg][i][c]
    out = torch.zeros(g, GROUP_FLOATS, device=rows.device, dtype=torch.float32)
    out.view(torch.float16)[:, :GROUP_FLOATS] = frag.to(torch.float16)
    out.view(torch.uint8)[:, F16S_EXP_OFF:F16S_EXP_OFF + 512] = b.permute(0, 2, 3, 1).reshape(g, 512).to(torch.uint8)   # [t][lane = 32 gg + i]
    return out.reshape(-1)


def frag_f16s_to_rows(slot: torch.Tensor, m: int) -> torch.Tensor:
    """The row-major [m, 256] fp32 values a scaled f16 gradient slot holds (test / debugging helper)."""
    g = frag_groups(m)
    flat = slot.reshape(-1)[: g * GROUP_FLOATS].view(g, GROUP_FLOATS)
    b = flat.view(torch.uint8)[:, F16S_EXP_OFF:F16S_EXP_OFF + 512].to(torch.int32).reshape(g, 8, 2, 32)     # [t][gg][i]
    inv = ((240 - b.clamp(max=239)) << 23).view(torch.float32)
    inv = torch.where(b == 255, torch.zeros_like(inv), inv)
    vals = flat.view(torch.float16)[:, :GROUP_FLOATS].float().reshape(g, 8, 4, 2, 32, 4) * inv[:, :, None, :, :, None]
    rows = vals.permute(0, 4, 1, 2, 3, 5).reshape(g * 32, 256)
    return rows[:m]


def weight_grad_frag(shape: int, dy, dy_form: int, x, x_form: int, n_points: int, groups: int, dw_part, db_part=None) -> None:
    _check(load().vfn_weight_grad_frag(C.c_int32(shape), _ptr(dy, "dy"), C.c_int32(dy_form), _ptr(x, "x"), C.c_int32(x_form),
                                       C.c_int64(n_points), C.c_int32(groups), _ptr(dw_part, "dw_part"), _ptr(db_part, "db_part"),
                                       _stream()), "vfn_weight_grad_frag")


class WgradLayer(C.Structure):
    """mirrors vfn_wgrad_layer"""
    _fields_ = [(n, C.c_void_p) for n in ("weight", "bias", "bn_weight", "bn_var", "bn_mean", "g_weight", "g_bias", "g_bn_weight", "g_bn_bias")]


def weight_grad_groups(n_points: int) -> int:
    return int(load().vfn_weight_grad_groups(C.c_int64(n_points)))


def net_weight_grads_scratch_bytes(kind: int, geom, n_points: int) -> int:
    n = int(load().vfn_net_weight_grads_scratch_bytes(C.c_int32(kind), C.byref(geom), C.c_int64(n_points)))
    if n < 0:
        _check(n, "vfn_net_weight_grads_scratch_bytes")
    return n


WGRAD_LAYERS, WGRAD_FEATURES, WGRAD_HEAD = 1, 2, 4


def net_weight_grads_frag(kind: int, geom, layer_table, saved, slot_index: int, dy, slot_floats: int, dy_form: int, x_form: int, feats,
                          aux, dz_head, n_points: int, with_features: bool, accumulate: bool, scratch, first_point: int = 0,
                          parts: Optional[int] = None) -> None:
    """All weight-gradient launches of one net + the un-fold, from C (csrc/vfn_wgrad.hip).  ``layer_table``: a WgradLayer array
    (parameter and gradient pointers per reference layer); ``saved`` / ``dy``: [slots, slot_floats] workspaces, ``slot_index`` the
    net's first slot in both."""
    for t, name in ((saved, "saved"), (dy, "dy")):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == slot_floats):
            raise VfnError(f"{name}: expected a contiguous CUDA float32 [slots, {slot_floats}] workspace")
    # ``first_point`` (a multiple of 32): the products run over points first_point .. first_point + n_points - 1 of the workspace;
    # ``parts``: WGRAD_* mask (default: everything, or everything but the feature block when ``with_features`` is False)
    if first_point % 32:
        raise VfnError(f"first_point = {first_point} is not a multiple of 32")
    off = slot_index * slot_floats * 4 + (first_point // 32) * GROUP_FLOATS * 4
    if parts is None:
        parts = (WGRAD_LAYERS | WGRAD_FEATURES | WGRAD_HEAD) if with_features else (WGRAD_LAYERS | WGRAD_HEAD)

    def at(t, row_floats):
        return None if t is None else C.c_void_p(t.data_ptr() + first_point * row_floats * 4)

    for t, name in ((feats, "feats"), (aux, "aux"), (dz_head, "dz_head")):
        if t is not None and not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise VfnError(f"{name}: expected a contiguous CUDA float32 tensor")
    _check(load().vfn_net_weight_grads_frag_part(C.c_int32(kind), C.byref(geom), layer_table, C.c_void_p(saved.data_ptr() + off),
                                                 C.c_void_p(dy.data_ptr() + off), C.c_int64(slot_floats * 4), C.c_int32(dy_form), C.c_int32(x_form),
                                                 at(feats, HIDDEN), at(aux, AUX_K), at(dz_head, 4), C.c_int64(n_points),
                                                 C.c_uint32(parts), C.c_int32(int(accumulate)), _ptr(scratch, "scratch", torch.uint8),
                                                 _stream()), "vfn_net_weight_grads_frag")


def mlp_bwd_chain_bf16_ws(vf_geom, vf_packed_bwd16, vf_head_w, rn_geom, rn_packed_bwd16, rn_head_w, feats, masks, dy, dy_flags: int,
                          d_colors, colors, d_vec, vec, d_feats, vec_stride: int, n_points: int, dz_rgb, dz_vec,
                          ws_first: int = 0, ws_points: Optional[int] = None):
    """``ws_first`` / ``ws_points``: the launch's points are points ws_first .. of a workspace (feats, masks, dy, dz_rgb, dz_vec)
    sized for ws_points; the per-sample inputs (d_colors, colors, d_vec, vec, d_feats) stay launch-local."""
    rn = C.byref(rn_geom) if rn_geom is not None else None
    _check(load().vfn_mlp_bwd_chain_bf16_ws_at(C.byref(vf_geom), _ptr(vf_packed_bwd16, "vf_packed_bwd16", torch.uint8),
                                               _ptr(vf_head_w, "vf_head_w"), rn,
                                               _ptr(rn_packed_bwd16, "rn_packed_bwd16", torch.uint8), _ptr(rn_head_w, "rn_head_w"),
                                               _ptr(feats, "feats"), _ptr(masks, "masks", torch.int32), _ptr(dy, "dy"), C.c_int32(dy_flags),
                                               _ptr(d_colors, "d_colors"), _ptr(colors, "colors"), _ptr(d_vec, "d_vec"), _ptr(vec, "vec"),
                                               _ptr(d_feats, "d_feats"), C.c_int32(vec_stride), C.c_int64(n_points),
                                               _ptr(dz_rgb, "dz_rgb"), _ptr(dz_vec, "dz_vec"), C.c_int64(ws_first),
                                               C.c_int64(n_points if ws_points is None else ws_points), _stream()), "vfn_mlp_bwd_chain_bf16_ws")


# ------------------------------------------------------------------------------------------------
# optimizer side over one flat buffer (csrc/vfn_adam.hip)
# ------------------------------------------------------------------------------------------------
def _regions(regions):
    n = len(regions)
    starts = (C.c_int64 * n)(*[int(r[0]) for r in regions])
    ends = (C.c_int64 * n)(*[int(r[1]) for r in regions])
    mults = (C.c_int32 * n)(*[int(r[2]) for r in regions])
    return n, starts, ends, mults


def flat_clip_workspace(device) -> torch.Tensor:
    return torch.zeros(int(load().vfn_flat_clip_workspace_bytes()), dtype=torch.uint8, device=device)


def flat_clip_grad_norm(flat_grad: torch.Tensor, regions, max_norm: float, workspace: torch.Tensor, out2: torch.Tensor) -> None:
    """regions: [(start, end, mult)]; out2[0] <- total norm, out2[1] <- clip coefficient; flat_grad scaled in place."""
    n, starts, ends, mults = _regions(regions)
    _check(load().vfn_flat_clip_grad_norm(_ptr(flat_grad, "flat_grad"), C.c_int64(flat_grad.numel()), C.c_int32(n), starts, ends, mults,
                                          C.c_float(max_norm), _ptr(workspace, "workspace", torch.uint8), _ptr(out2, "out2"), _stream()),
           "vfn_flat_clip_grad_norm")


def flat_adam_step(param, grad, exp_avg, exp_avg_sq, regions, step_size, bc2_sqrt, beta1, beta2, eps, weight_decay) -> None:
    n, starts, ends, mults = _regions(regions)
    ss = (C.c_double * (2 * n))(*[float(x) for x in step_size])
    bc = (C.c_double * (2 * n))(*[float(x) for x in bc2_sqrt])
    _check(load().vfn_flat_adam_step(_ptr(param, "param"), _ptr(grad, "grad"), _ptr(exp_avg, "exp_avg"), _ptr(exp_avg_sq, "exp_avg_sq"),
                                     C.c_int64(param.numel()), C.c_int32(n), starts, ends, mults, ss, bc, C.c_double(beta1), C.c_double(beta2),
                                     C.c_double(eps), C.c_double(weight_decay), _stream()), "vfn_flat_adam_step")


class UnfoldEntry(C.Structure):
    """mirrors vfn_unfold_entry"""
    _fields_ = [(n, C.c_void_p) for n in ("dw_act", "dw_aux", "db", "w", "b_lin", "bn_w", "bn_var", "bn_mean", "g_w", "g_b",
                                          "g_bn_w", "g_bn_b")] + \
               [(n, C.c_int32) for n in ("rows", "row_off", "in_dim", "slab_rows", "act_c0", "act_nc", "aux_c0", "aux_nc")] + \
               [("scale", C.c_float)] + [(n, C.c_int32) for n in ("groups_act", "groups_aux", "groups_db")]


def unfold_weight_grads(entries: Sequence[dict], groups: int, accumulate_mask: int = 0) -> None:
    """entries: dicts with the tensors / ints of vfn_unfold_entry (missing tensors = NULL); bit i of ``accumulate_mask``: entry i
    adds to its g_* tensors instead of overwriting them."""
    arr = (UnfoldEntry * len(entries))()
    for i, e in enumerate(entries):
        for name, _ in UnfoldEntry._fields_[:12]:
            setattr(arr[i], name, _ptr(e.get(name), name))
        for name, _ in UnfoldEntry._fields_[12:20]:
            setattr(arr[i], name, int(e.get(name, 0)))
        arr[i].scale = float(e.get("scale", 1.0))
    _check(load().vfn_unfold_weight_grads_acc(arr, C.c_int32(len(entries)), C.c_int32(groups), C.c_uint32(accumulate_mask), _stream()),
           "vfn_unfold_weight_grads")


def weight_grad_partials_bf16(dy, x, n_points: int, groups: int, dw_part, db_part=None, x_f16: bool = False):
    """shape-0 weight_grad_partials (256 x 256, every column valid) on the bf16 matrix cores (split operands)."""
    _check(load().vfn_weight_grad_partials_bf16(_ptr(dy, "dy"), _ptr(x, "x"), C.c_int64(n_points), C.c_int32(groups),
                                                _ptr(dw_part, "dw_part"), _ptr(db_part, "db_part"), C.c_int32(int(x_f16)), _stream()),
           "vfn_weight_grad_partials_bf16")


def weight_grad_partials_bf16_cols(dy, x, n_points: int, groups: int, dw_part, db_part=None):
    """The same over 256 columns of wider matrices (``Cols`` views: 16-byte aligned column offsets, leading dimensions >= 256)."""
    dy, x = _cols(dy), _cols(x)
    _check(load().vfn_weight_grad_partials_bf16_ld(dy.ptr, C.c_int32(dy.ld), x.ptr, C.c_int32(x.ld), C.c_int64(n_points), C.c_int32(groups),
                                                   _ptr(dw_part, "dw_part"), _ptr(db_part, "db_part"), C.c_int32(0), _stream()),
           "vfn_weight_grad_partials_bf16_ld")


def weight_grad_partials_bf16_fold(dy, z_prev, coef_prev: torch.Tensor, n_prev: int, post_prev: float, n_points: int, groups: int, dw_part, db_part=None):
    """dW = dY^T act(z_prev) over 256 columns with the activation formed in the operand read (vfn_weight_grad_partials_bf16_fold)."""
    dy, z_prev = _cols(dy), _cols(z_prev)
    if coef_prev.numel() < 2 * n_prev:
        raise VfnError(f"weight_grad_partials_bf16_fold: coefficients of {coef_prev.numel()} floats for n_prev = {n_prev} (needs scale | shift rows)")
    _check(load().vfn_weight_grad_partials_bf16_fold(dy.ptr, C.c_int32(dy.ld), z_prev.ptr, C.c_int32(z_prev.ld), _ptr(coef_prev, "coef_prev"),
                                                     C.c_int32(n_prev), C.c_float(post_prev), C.c_int64(n_points), C.c_int32(groups),
                                                     _ptr(dw_part, "dw_part"), _ptr(db_part, "db_part"), _stream()),
           "vfn_weight_grad_partials_bf16_fold")


def ray_density_weights_bwd(dp: DensityParams, normals, ray_dirs, z_vals, scalars, colors, d_rgb, d_depth, d_weights,
                            d_normals, d_colors, d_scalars):
    n, s = z_vals.shape
    dp.n_rays, dp.n_samples = n, s
    _check(load().vfn_ray_density_weights_bwd(C.byref(dp), _ptr(normals, "normals"), _ptr(ray_dirs, "ray_dirs"),
                                              _ptr(z_vals, "z_vals"), _ptr(scalars, "scalars"), _ptr(colors, "colors"),
                                              _ptr(d_rgb, "d_rgb"), _ptr(d_depth, "d_depth"),
                                              _ptr(d_weights, "d_weights"), _ptr(d_normals, "d_normals"),
                                              _ptr(d_colors, "d_colors"), _ptr(d_scalars, "d_scalars"), _stream()),
           "vfn_ray_density_weights_bwd")


# ------------------------------------------------------------------------------------------------
# f16x3 inference kernels
# ------------------------------------------------------------------------------------------------
PACK16_STATS_WORDS = 64          # tail of an f16x3 pack: max |folded weight| per pack entry (include/vfn.h, "Range guard")
STATUS_ACT_SATURATED, STATUS_INPUT_SATURATED = 1, 2


def f16x3_set_status(word: Optional[torch.Tensor]) -> None:
    """Route the range reports of this thread's f16x3 launches into ``word`` (int32 device tensor, >= 1 element); None: off."""
    _check(load().vfn_f16x3_set_status(_ptr(word, "status_word", torch.int32)), "vfn_f16x3_set_status")


def train_step_workspace_bytes(params: "TrainStepParams", vf_geom: NetGeom, rn_geom: NetGeom) -> int:
    n = int(load().vfn_train_step_workspace_bytes(C.byref(params), C.byref(vf_geom), C.byref(rn_geom)))
    if n < 0:
        raise VfnError(f"vfn_train_step_workspace_bytes failed (status {n}): {load().vfn_last_error().decode()}")
    return n


def train_step(params: "TrainStepParams", io: "TrainStepIO") -> None:
    """One training step (or some of its phases) from C: vf_nerf_amd/stepengine.py fills the structs."""
    _check(load().vfn_train_step(C.byref(params), C.byref(io), _stream()), "vfn_train_step")


def train_step_workspace_layout(params: "TrainStepParams", vf_geom: NetGeom, rn_geom: NetGeom) -> List[int]:
    """[TWS_*]: byte offsets of the session form's regions inside the step workspace, and its counts."""
    out = (C.c_int64 * TWS_COUNT)()
    _check(load().vfn_train_step_workspace_layout(C.byref(params), C.byref(vf_geom), C.byref(rn_geom), out, TWS_COUNT), "vfn_train_step_workspace_layout")
    return [int(v) for v in out]


def train_step_supervision_points(params: "TrainStepParams", io: "TrainStepIO", inward: bool, r_min: float, r_max: float, cx: float, cy: float,
                                  cz: float, centroid_dev: Optional[torch.Tensor], row0: int, count: int, u: Optional[torch.Tensor], seed: int,
                                  offset: int) -> bool:
    """Session form: one supervision batch sampled into rows [row0, row0 + count) of the open step.  -> True when it ran on the step's side
    stream (pass it on to ``train_step_supervision_forward``)."""
    rc = load().vfn_train_step_supervision_points(C.byref(params), C.byref(io), 1 if inward else 0, float(r_min), float(r_max), float(cx), float(cy),
                                                  float(cz), _ptr(centroid_dev, "centroid"), row0, count, _ptr(u, "u"), C.c_uint64(seed & (2 ** 64 - 1)),
                                                  C.c_uint64(offset & (2 ** 64 - 1)), _stream())
    if rc < 0:
        _check(rc, "vfn_train_step_supervision_points")
    return rc == 1


def train_step_supervision_forward(params: "TrainStepParams", io: "TrainStepIO", row0: int, count: int, on_side: bool) -> None:
    """Session form: the vector-only saving forward over supervision rows [row0, pad32(row0 + count)) of the open step."""
    _check(load().vfn_train_step_supervision_forward(C.byref(params), C.byref(io), row0, count, 1 if on_side else 0, _stream()),
           "vfn_train_step_supervision_forward")


def train_step_supervision_backward(params: "TrainStepParams", io: "TrainStepIO", row0: int, count: int) -> None:
    """Session form: chain + weight gradients of supervision rows [row0, pad32(row0 + count)) alone (a backward pass that never reaches the render)."""
    _check(load().vfn_train_step_supervision_backward(C.byref(params), C.byref(io), row0, count, _stream()), "vfn_train_step_supervision_backward")


def f16x3_set_clock_probe(stamps: Optional[torch.Tensor]) -> None:
    """Route the per-workgroup clock stamps of this thread's gradient-free fused launches into ``stamps`` (int64 device tensor
    [slots, 2]: shader-clock cycles, 100 MHz ticks of workgroup b; csrc/vfn_mlp16.hip); None: off."""
    slots = 0 if stamps is None else stamps.numel() // 2
    _check(load().vfn_f16x3_set_clock_probe(_ptr(stamps, "stamps", torch.int64), slots), "vfn_f16x3_set_clock_probe")


def clock_ghz_from_stamps(stamps: torch.Tensor) -> Optional[dict]:
    """{median, min, max} shader clock in GHz over the workgroups that left a stamp (cycles / 100-MHz-ticks x 0.1)."""
    st = stamps.reshape(-1, 2).cpu().double()
    st = st[(st[:, 1] > 0) & (st[:, 0] > 0)]
    if st.shape[0] == 0:
        return None
    ghz = st[:, 0] / st[:, 1] * 0.1
    return {"median": round(float(ghz.median()), 4), "min": round(float(ghz.min()), 4), "max": round(float(ghz.max()), 4),
            "workgroups": int(st.shape[0]), "median_workgroup_us": round(float(st[:, 1].median()) / 100.0, 2)}


def pack16_size(kind: int, geom: NetGeom) -> int:
    n = load().vfn_pack16_size(kind, C.byref(geom))
    if n < 0:
        raise VfnError(f"unsupported network geometry: {load().vfn_last_error().decode()}")
    return int(n)


def pack16_weights(kind: int, geom: NetGeom, layers: Sequence[dict], packed16: torch.Tensor) -> None:
    _check(load().vfn_pack16_weights(kind, C.byref(geom), _layer_array(geom, layers), _ptr(packed16, "packed16", torch.uint8),
                                     _stream()), "vfn_pack16_weights")


def vf_mlp16_fwd(geom: NetGeom, packed16, points):
    m = points.shape[0]
    out = torch.empty(m, 3, device=points.device)
    _check(load().vfn_vf_mlp16_fwd(C.byref(geom), _ptr(packed16, "packed16", torch.uint8), _ptr(points, "points"),
                                   C.c_int64(m), _ptr(out, "out"), _stream()), "vfn_vf_mlp16_fwd")
    return out


BLOCK_BYTES = 1024   # one point's 256 features as split-f16 operand blocks


def vf_feat16_fwd(geom: NetGeom, packed16, points, out_vec, out_blocks) -> None:
    """VF net on points[M,3]; writes out_vec[M,3] and the feature operand blocks: out_blocks[block_rows(M), 1024] (uint8),
    32-row groups in the rendering kernel's register order — slices of larger buffers are fine when they start on a
    multiple of 32 rows."""
    m = points.shape[0]
    if out_blocks.shape[0] < block_rows(m):
        raise VfnError(f"out_blocks holds {out_blocks.shape[0]} rows, {block_rows(m)} (whole groups of 32) are needed")
    _check(load().vfn_vf_feat16_fwd(C.byref(geom), _ptr(packed16, "packed16", torch.uint8), _ptr(points, "points"),
                                    C.c_int64(m), _ptr(out_vec, "out_vec"), _ptr(out_blocks, "out_blocks", torch.uint8),
                                    _stream()), "vfn_vf_feat16_fwd")


def _fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray, out_index, colour_products,
                      normals, colors) -> None:
    _check(load().vfn_vf_render_fused16_products(C.byref(vf_geom), _ptr(vf_packed16, "vf_packed16", torch.uint8),
                                                 C.byref(rn_geom), _ptr(rn_packed16, "rn_packed16", torch.uint8),
                                                 _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"), C.c_int64(points.shape[0]),
                                                 C.c_int32(samples_per_ray),
                                                 _ptr(out_index, "out_index", torch.int32) if out_index is not None else None,
                                                 C.c_int32(colour_products), _ptr(normals, "normals"), _ptr(colors, "colors"),
                                                 _stream()), "vfn_vf_render_fused16_products")


def vf_render_fused16_scatter(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray: int, out_index,
                              normals, colors, colour_products: int = 3) -> None:
    """Fused VF + rendering launch over points[M,3] (view direction of point m: ray_dirs[m // samples_per_ray]); the outputs of
    point m are written to row out_index[m] of the caller's normals / colors (int32; negative: dropped).  colour_products: see
    vfn_vf_render_fused16_products."""
    m = points.shape[0]
    if out_index.shape[0] < m:
        raise VfnError(f"out_index holds {out_index.shape[0]} entries for {m} points")
    if colour_products != 3:
        return _fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray, out_index,
                                 colour_products, normals, colors)
    _check(load().vfn_vf_render_fused16_scatter(C.byref(vf_geom), _ptr(vf_packed16, "vf_packed16", torch.uint8),
                                                C.byref(rn_geom), _ptr(rn_packed16, "rn_packed16", torch.uint8),
                                                _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"), C.c_int64(m),
                                                C.c_int32(samples_per_ray), _ptr(out_index, "out_index", torch.int32),
                                                _ptr(normals, "normals"), _ptr(colors, "colors"), _stream()),
           "vfn_vf_render_fused16_scatter")


def select_samples(weights, points, ray_dirs, sigma=None, z_vals=None):
    """The sparse colour branch's sample selection on its own (vfn_select_samples) -> the selected indices ray * S + j as a Python list
    (synchronises: tests and diagnostics).  ``sigma`` / ``z_vals`` given: the training selection."""
    n, s = weights.shape
    dev = weights.device
    scratch = torch.empty(2 * n, dtype=torch.int32, device=dev)
    count = torch.zeros(4, dtype=torch.int32, device=dev)
    index = torch.empty(n * s, dtype=torch.int32, device=dev)
    pts_sel, dirs_sel = torch.empty(n * s, 3, device=dev), torch.empty(n * s, 3, device=dev)
    _check(load().vfn_select_samples(_ptr(weights, "weights"), _ptr(sigma, "sigma"), _ptr(z_vals, "z_vals"), C.c_int32(n), C.c_int32(s),
                                     _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"), _ptr(scratch, "scratch", torch.int32),
                                     _ptr(count, "count", torch.int32), _ptr(index, "index", torch.int32), _ptr(pts_sel, "points_sel"),
                                     _ptr(dirs_sel, "dirs_sel"), _stream()), "vfn_select_samples")
    k = int(count[0])
    got = index[:k].tolist()
    flat = points.reshape(-1, 3)
    assert torch.equal(pts_sel[:k], flat[index[:k].long()]) and torch.equal(dirs_sel[:k], ray_dirs[(index[:k] // s).long()])
    return got


def scatter_rows3(a, b, index, out_a, out_b) -> None:
    """out_a[index[r]] = a[r], out_b[index[r]] = b[r] for [n,3] fp32 rows (int32 index, negative entries skipped)."""
    n = a.shape[0]
    if index.shape[0] < n:
        raise VfnError(f"index holds {index.shape[0]} entries for {n} rows")
    _check(load().vfn_scatter_rows3(_ptr(a, "a"), _ptr(b, "b"), _ptr(index, "index", torch.int32), C.c_int64(n), _ptr(out_a, "out_a"),
                                    _ptr(out_b, "out_b"), _stream()), "vfn_scatter_rows3")


def render16_from_blocks(rn_geom: NetGeom, rn_packed16, blocks, vecs, dst, points, ray_dirs, samples_per_ray: int,
                         normals=None, colors=None):
    """Rendering net over the stored rows: row r -> sorted position dst[r] (points[dst[r]], ray_dirs[dst[r] / S]); returns
    normals[M,3], colors[M,3] in sorted order (M = points.shape[0]; every sorted sample must be some row's dst)."""
    m = points.shape[0]
    n_rows = dst.shape[0]
    dev = points.device
    if blocks.shape[0] < block_rows(n_rows) or vecs.shape[0] < n_rows:
        raise VfnError(f"{n_rows} rows need blocks[{block_rows(n_rows)}] and vecs[{n_rows}]")
    if normals is None:
        normals = torch.empty(m, 3, device=dev)
        colors = torch.empty(m, 3, device=dev)
    _check(load().vfn_render16_from_blocks(C.byref(rn_geom), _ptr(rn_packed16, "rn_packed16", torch.uint8),
                                           _ptr(blocks, "blocks", torch.uint8), _ptr(vecs, "vecs"),
                                           _ptr(dst, "dst", torch.int32), _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"),
                                           C.c_int64(n_rows), C.c_int32(samples_per_ray), _ptr(normals, "normals"),
                                           _ptr(colors, "colors"), _stream()), "vfn_render16_from_blocks")
    return normals, colors


def vf_mlp16_fwd_train(geom: NetGeom, packed16, points, with_features: bool, saved, aux_vf, masks, save_f16: int = 0,
                       ws_first: int = 0, ws_points: Optional[int] = None):
    """f16x3 VF forward that fills the backward workspace; returns the vector columns [M,3] (the features, when
    evaluated, are in their ``saved`` slot).  ``ws_first`` / ``ws_points``: as in vf_render_fused16_fwd_train."""
    m = points.shape[0]
    out = torch.empty(m, 3, device=points.device)
    _check(load().vfn_vf_mlp16_fwd_train_at(C.byref(geom), _ptr(packed16, "packed16", torch.uint8), _ptr(points, "points"),
                                            C.c_int64(m), C.c_int32(1 if with_features else 0), _ptr(out, "out"),
                                            _ptr(saved, "saved"), _ptr(aux_vf, "aux_vf"), _ptr(masks, "masks", torch.int32),
                                            C.c_int32(int(save_f16)), C.c_int64(ws_first), C.c_int64(m if ws_points is None else ws_points),
                                            _stream()), "vfn_vf_mlp16_fwd_train")
    return out


def vf_render_fused16_fwd_train(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray, saved,
                                aux_vf, aux_rn, masks, save_f16: int = 0, ws_first: int = 0, ws_points: Optional[int] = None,
                                normals=None, colors=None, colour_products: int = 3):
    """``ws_first`` / ``ws_points``: the launch fills points ws_first .. of a workspace sized for ws_points points (default: the
    whole workspace = this launch's points).  ``normals`` / ``colors``: optional [m,3] outputs to write into."""
    m = points.shape[0]
    dev = points.device
    normals = torch.empty(m, 3, device=dev) if normals is None else normals
    colors = torch.empty(m, 3, device=dev) if colors is None else colors
    _check(load().vfn_vf_render_fused16_fwd_train_at(C.byref(vf_geom), _ptr(vf_packed16, "vf_packed16", torch.uint8),
                                                     C.byref(rn_geom), _ptr(rn_packed16, "rn_packed16", torch.uint8),
                                                     _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"), C.c_int64(m),
                                                     C.c_int32(samples_per_ray), _ptr(normals, "normals"),
                                                     _ptr(colors, "colors"), _ptr(saved, "saved"), _ptr(aux_vf, "aux_vf"),
                                                     _ptr(aux_rn, "aux_rn"), _ptr(masks, "masks", torch.int32),
                                                     C.c_int32(int(save_f16)), C.c_int64(ws_first), C.c_int64(m if ws_points is None else ws_points),
                                                     C.c_int32(colour_products), _stream()), "vfn_vf_render_fused16_fwd_train")
    return normals, colors


def vf_render_fused16_fwd(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray: int, colour_products: int = 3):
    m = points.shape[0]
    dev = points.device
    normals = torch.empty(m, 3, device=dev)
    colors = torch.empty(m, 3, device=dev)
    if colour_products != 3:
        _fused16_products(vf_geom, vf_packed16, rn_geom, rn_packed16, points, ray_dirs, samples_per_ray, None, colour_products,
                          normals, colors)
        return normals, colors
    _check(load().vfn_vf_render_fused16_fwd(C.byref(vf_geom), _ptr(vf_packed16, "vf_packed16", torch.uint8),
                                            C.byref(rn_geom), _ptr(rn_packed16, "rn_packed16", torch.uint8),
                                            _ptr(points, "points"), _ptr(ray_dirs, "ray_dirs"), C.c_int64(m),
                                            C.c_int32(samples_per_ray), _ptr(normals, "normals"),
                                            _ptr(colors, "colors"), _stream()), "vfn_vf_render_fused16_fwd")
    return normals, colors


# ------------------------------------------------------------------------------------------------
# dense-grid stages (mesh extraction)
# ------------------------------------------------------------------------------------------------
def grid_lattice_points(axes, n: int, row0: int, count: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Rows [row0, row0 + count) of the separable n^3 lattice with axis tables ``axes`` = (a0[n], a1[n], a2[n]) -> points[count,3]."""
    a0, a1, a2 = axes
    if out is None:
        out = torch.empty(count, 3, device=a0.device)
    _check(load().vfn_grid_lattice_points(_ptr(a0, "axis0"), _ptr(a1, "axis1"), _ptr(a2, "axis2"), C.c_int32(n), C.c_int64(row0), C.c_int64(count),
                                          _ptr(out, "points"), _stream()), "vfn_grid_lattice_points")
    return out


def grid_divergence(vt: torch.Tensor, n: int, threshold: float = -0.5) -> torch.Tensor:
    out = torch.empty(n, n, n, device=vt.device)
    _check(load().vfn_grid_divergence(_ptr(vt, "vt"), C.c_int32(n), C.c_float(threshold), _ptr(out, "out"), _stream()),
           "vfn_grid_divergence")
    return out


def grid_smooth_axis(src: torch.Tensor, dst: torch.Tensor, n: int, axis: int, weights: Sequence[float]) -> None:
    w = (C.c_float * len(weights))(*[float(x) for x in weights])
    _check(load().vfn_grid_smooth_axis(_ptr(src, "in"), _ptr(dst, "out"), C.c_int32(n), C.c_int32(axis), w,
                                       C.c_int32(len(weights)), _stream()), "vfn_grid_smooth_axis")


def grid_unify_direction(divergence: torch.Tensor, vt: torch.Tensor, n: int) -> torch.Tensor:
    choice = torch.empty(n * n * n, 8, dtype=torch.int64, device=vt.device)
    _check(load().vfn_grid_unify_direction(_ptr(divergence, "divergence"), _ptr(vt, "vt"), C.c_int32(n),
                                           _ptr(choice, "choice", torch.int64), _stream()), "vfn_grid_unify_direction")
    return choice


def grid_unify_direction_sides(divergence: torch.Tensor, vt: torch.Tensor, n: int, want_table: bool = True):
    """-> (sides[n^3] uint8: bit q = corner q's side, choice[n^3,8] int64 | None)."""
    sides = torch.empty(n * n * n, dtype=torch.uint8, device=vt.device)
    choice = torch.empty(n * n * n, 8, dtype=torch.int64, device=vt.device) if want_table else None
    _check(load().vfn_grid_unify_direction_sides(_ptr(divergence, "divergence"), _ptr(vt, "vt"), C.c_int32(n), _ptr(sides, "sides", torch.uint8),
                                                 _ptr(choice, "choice", torch.int64), _stream()), "vfn_grid_unify_direction_sides")
    return sides, choice


def grid_comb_format_sides(sides: torch.Tensor, norms: torch.Tensor, n: int):
    dev = norms.device
    different = torch.empty(n * n * n, 28, device=dev)
    pair_norms = torch.empty(n * n * n, 28, 2, device=dev)
    _check(load().vfn_grid_comb_format_sides(_ptr(sides, "sides", torch.uint8), _ptr(norms, "norms"), C.c_int32(n),
                                             _ptr(different, "different_side"), _ptr(pair_norms, "pair_norms"), _stream()),
           "vfn_grid_comb_format_sides")
    return different, pair_norms


def grid_comb_format(choice: torch.Tensor, norms: torch.Tensor, n: int):
    dev = norms.device
    different = torch.empty(n * n * n, 28, device=dev)
    pair_norms = torch.empty(n * n * n, 28, 2, device=dev)
    _check(load().vfn_grid_comb_format(_ptr(choice, "choice", torch.int64), _ptr(norms, "norms"), C.c_int32(n),
                                       _ptr(different, "different_side"), _ptr(pair_norms, "pair_norms"), _stream()),
           "vfn_grid_comb_format")
    return different, pair_norms


# ------------------------------------------------------------------------------------------------
# networks in training mode: batch-statistics BatchNorm, one launch per layer (csrc/vfn_bstat.hip)
# ------------------------------------------------------------------------------------------------
class Cols:
    """Columns [col0, col0 + ...) of a contiguous row-major fp32 matrix: what the row-wise entry points take as
    (pointer, leading dimension)."""

    def __init__(self, base: torch.Tensor, col0: int = 0) -> None:
        if not (base.is_cuda and base.dtype == torch.float32 and base.is_contiguous() and base.dim() == 2):
            raise VfnError("Cols: expected a contiguous 2-D fp32 CUDA/HIP tensor (the HIP path has no CPU fallback)")
        self.base, self.col0, self.ld = base, col0, base.shape[1]

    @property
    def ptr(self) -> C.c_void_p:
        return C.c_void_p(self.base.data_ptr() + 4 * self.col0)

    # what _ptr() asks of a tensor, so that the (pointer, ld) entry points written for whole tensors take column views too
    is_cuda, dtype = True, torch.float32

    def is_contiguous(self) -> bool:
        return True

    def data_ptr(self) -> int:
        return self.base.data_ptr() + 4 * self.col0


def _cols(x) -> "Cols":
    return x if isinstance(x, Cols) else Cols(x)


ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2


def linear_rows_stat_parts(m: int) -> int:
    return int(load().vfn_linear_rows_stat_parts(m))


def bstat_row_parts(m: int) -> int:
    return int(load().vfn_bstat_row_parts(m))


GEMM_EXACT, GEMM_SPLIT_F16, GEMM_SPLIT_BF16, GEMM_BF16X6 = 0, 2, 4, 6      # arithmetic of vfn_linear_rows (bits 1-2 of its first argument)


_wplanes_cache: dict = {}


def wplanes(n_out: int, k_in: int, device) -> torch.Tensor:
    """The scratch the split layer products put W's 16-bit planes into (vfn_linear_rows_ws): one buffer per device AND stream, grown on
    demand — calls on one stream use it one after the other; calls on different streams must not share it."""
    need = int(load().vfn_linear_rows_wplanes_bytes(C.c_int32(n_out), C.c_int32(k_in)))
    key = (str(device), int(_stream().value or 0))
    buf = _wplanes_cache.get(key)
    if buf is None or buf.numel() < need:
        buf = _wplanes_cache[key] = torch.empty(max(need, 1 << 20), dtype=torch.uint8, device=device)
    return buf


def linear_rows(a, w: torch.Tensor, bias, m: int, n_out: int, k_in: int, c, act: int = ACT_NONE, transpose_w: bool = False,
                stats_part=None, arith: int = GEMM_EXACT, planes=None) -> None:
    """``arith``: GEMM_EXACT (fp32 matrix instruction), GEMM_SPLIT_F16 (three f16 products per product, 22 bits: forward GEMMs on
    normalised activations), GEMM_SPLIT_BF16 (three bf16 products, 16 bits with fp32's exponent range) or GEMM_BF16X6 (operands in three
    bf16 parts, six products: 24 bits at fp32's exponent range, fp32-equivalent — backward GEMMs on gradients of any magnitude)."""
    a, c = _cols(a), _cols(c)
    _check(load().vfn_linear_rows_ws(C.c_int32(int(transpose_w) | int(arith)), a.ptr, C.c_int32(a.ld), _ptr(w, "w"), C.c_int32(w.shape[1]),
                                     _ptr(bias, "bias"), C.c_int64(m), C.c_int32(n_out), C.c_int32(k_in), C.c_int32(act), c.ptr,
                                     C.c_int32(c.ld), _ptr(stats_part, "stats_part"), _ptr(planes, "planes", torch.uint8), _stream()),
           "vfn_linear_rows")


def linear_rows_fold(z_prev, coef_prev: torch.Tensor, n_prev: int, post_prev: float, w: torch.Tensor, bias, m: int, n_out: int, k_in: int, c,
                     stats_part=None, arith: int = GEMM_SPLIT_F16, planes=None) -> None:
    """The forward product of a layer whose input is act(z_prev) — the previous layer's BatchNorm + ReLU (+ the skip layer's encoding
    columns behind the first ``n_prev``) — formed inside the product's operand read (include/vfn.h, vfn_linear_rows_fold)."""
    z_prev, c = _cols(z_prev), _cols(c)
    if coef_prev.numel() < 2 * n_prev or n_prev > k_in:
        raise VfnError(f"linear_rows_fold: coefficients of {coef_prev.numel()} floats for n_prev = {n_prev} (needs scale | shift rows), k_in = {k_in}")
    _check(load().vfn_linear_rows_fold(C.c_int32(int(arith)), z_prev.ptr, C.c_int32(z_prev.ld), _ptr(coef_prev, "coef_prev"), C.c_int32(n_prev),
                                       C.c_float(post_prev), _ptr(w, "w"), C.c_int32(w.shape[1]), _ptr(bias, "bias"), C.c_int64(m), C.c_int32(n_out),
                                       C.c_int32(k_in), c.ptr, C.c_int32(c.ld), _ptr(stats_part, "stats_part"), _ptr(planes, "planes", torch.uint8),
                                       _stream()), "vfn_linear_rows_fold")


def linear_rows_dx_sums(dz, w: torch.Tensor, m: int, n_out: int, k_in: int, c, z_prev, coef_prev: torch.Tensor, n_prev: int, post_prev: float,
                        part: torch.Tensor, arith: int = 4, planes=None) -> None:
    """C = dZ W (``arith``: GEMM_SPLIT_BF16 = three bf16 products, GEMM_BF16X6 = bf16 in three parts) + the per-workgroup partials of the previous layer's BatchNorm-backward column sums, taken from C
    in registers (include/vfn.h, vfn_linear_rows_dx_sums); ``part`` [linear_rows_stat_parts(m), 2, n_prev]."""
    dz, c, z_prev = _cols(dz), _cols(c), _cols(z_prev)
    _check(load().vfn_linear_rows_dx_sums(dz.ptr, C.c_int32(dz.ld), _ptr(w, "w"), C.c_int32(w.shape[1]), C.c_int64(m), C.c_int32(n_out),
                                          C.c_int32(k_in), c.ptr, C.c_int32(c.ld), z_prev.ptr, C.c_int32(z_prev.ld), _ptr(coef_prev, "coef_prev"),
                                          C.c_int32(n_prev), C.c_float(post_prev), _ptr(part, "part"), C.c_int32(int(arith)),
                                          _ptr(planes, "planes", torch.uint8), _stream()), "vfn_linear_rows_dx_sums")


def colsum_finish(part: torch.Tensor, n_parts: int, width: int, sums: torch.Tensor) -> None:
    _check(load().vfn_colsum_finish(_ptr(part, "part"), C.c_int64(n_parts), C.c_int32(width), _ptr(sums, "sums", torch.float64),
                                    _stream()), "vfn_colsum_finish")


def bstat_finalize(sums, m: int, n: int, gamma, beta, eps: float, momentum: float, running_mean, running_var, coef) -> None:
    _check(load().vfn_bstat_finalize(_ptr(sums, "sums", torch.float64), C.c_int64(m), C.c_int32(n), _ptr(gamma, "gamma"),
                                     _ptr(beta, "beta"), C.c_float(eps), C.c_float(momentum), _ptr(running_mean, "running_mean"),
                                     _ptr(running_var, "running_var"), _ptr(coef, "coef"), _stream()), "vfn_bstat_finalize")


def bstat_relu_rows(z, coef, m: int, n: int, post_scale: float, h) -> None:
    z, h = _cols(z), _cols(h)
    _check(load().vfn_bstat_relu_rows(z.ptr, C.c_int32(z.ld), _ptr(coef, "coef"), C.c_int64(m), C.c_int32(n),
                                      C.c_float(post_scale), h.ptr, C.c_int32(h.ld), _stream()), "vfn_bstat_relu_rows")


def bstat_relu_bwd_sums(g, z, coef, m: int, n: int, post_scale: float, part) -> None:
    g, z = _cols(g), _cols(z)
    _check(load().vfn_bstat_relu_bwd_sums(g.ptr, C.c_int32(g.ld), z.ptr, C.c_int32(z.ld),
                                          _ptr(coef, "coef"), C.c_int64(m), C.c_int32(n), C.c_float(post_scale),
                                          _ptr(part, "part"), _stream()), "vfn_bstat_relu_bwd_sums")


def bstat_relu_bwd_rows(g, z, coef, sums, m: int, n: int, post_scale: float, dz) -> None:
    g, z, dz = _cols(g), _cols(z), _cols(dz)
    _check(load().vfn_bstat_relu_bwd_rows(g.ptr, C.c_int32(g.ld), z.ptr, C.c_int32(z.ld),
                                          _ptr(coef, "coef"), _ptr(sums, "sums", torch.float64), C.c_int64(m), C.c_int32(n),
                                          C.c_float(post_scale), dz.ptr, C.c_int32(dz.ld), _stream()), "vfn_bstat_relu_bwd_rows")


def act_bwd_rows(act: int, dy, y, m: int, n: int, dz, onehot_col: int = -1) -> None:
    y, dz = _cols(y), _cols(dz)
    dy = None if dy is None else _cols(dy)
    _check(load().vfn_act_bwd_rows(C.c_int32(act), dy.ptr if dy is not None else C.c_void_p(0),
                                   C.c_int32(dy.ld if dy is not None else 0), y.ptr, C.c_int32(y.ld), C.c_int64(m), C.c_int32(n),
                                   C.c_int32(onehot_col), dz.ptr, C.c_int32(dz.ld), _stream()), "vfn_act_bwd_rows")


def embed_rows(src, m: int, multires: int, dst, scale: float = 1.0, rows_per_src: int = 1) -> None:
    src, dst = _cols(src), _cols(dst)
    _check(load().vfn_embed_rows(src.ptr, C.c_int32(src.ld), C.c_int32(rows_per_src), C.c_int64(m), C.c_int32(multires),
                                 C.c_float(scale), C.c_void_p(dst.base.data_ptr()), C.c_int32(dst.ld), C.c_int32(dst.col0),
                                 _stream()), "vfn_embed_rows")


def embed_rows_bwd(src3: torch.Tensor, m: int, multires: int, d_a, scale_a: float, d_b, scale_b: float, d_src3: torch.Tensor,
                   accumulate: bool = False) -> None:
    d_a = _cols(d_a)
    d_b = None if d_b is None else _cols(d_b)
    _check(load().vfn_embed_rows_bwd(_ptr(src3, "src3"), C.c_int64(m), C.c_int32(multires), d_a.ptr, C.c_int32(d_a.ld),
                                     C.c_int32(0), C.c_float(scale_a), d_b.ptr if d_b is not None else C.c_void_p(0),
                                     C.c_int32(d_b.ld if d_b is not None else 0), C.c_int32(0), C.c_float(scale_b),
                                     _ptr(d_src3, "d_src3"), C.c_int32(int(accumulate)), _stream()), "vfn_embed_rows_bwd")
