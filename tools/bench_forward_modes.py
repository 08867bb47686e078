"""The f16x3 forward launches side by side on the same points: vector-only / with features / fused with the rendering net, each
gradient-free and as the activation-saving training forward (fragment-ordered f16 workspace).
    python tools/bench_forward_modes.py [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vf_nerf_amd import lib  # noqa: E402
from vf_nerf_amd.backward import _Workspace  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda", 0)
built = bench.build_trained_scene(dev, 4096, 64, 64, seed=0)
model = built[0] if built is not None else bench.build_scene(dev, 4096, 64, 64, seed=0)[0]
vf, rn = model.vector_field_network, model.rendering_network
g_vf, g_rn, p_vf, p_rn = vf.geometry(), rn.geometry(), vf.packed16_weights(), rn.packed16_weights()
torch.manual_seed(0)
pts = (torch.rand(m, 3, device=dev) - 0.5) * 1.2
dirs = torch.nn.functional.normalize(torch.randn(m // 64, 3, device=dev), dim=-1)
ws = _Workspace(m, 13, dev, f16=True, frag=True, dy16="f16")
flags = ws.fwd_flags()


def timed(name, fn, flop_per_point):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:58s} {ms:7.3f} ms   {m * flop_per_point * 3 / ms / 1e9:7.1f} TFLOP/s executed (3 f16 products)")


VF_VEC = 2.0 * (39 * 256 + 6 * 256 * 256 + 295 * 256 + 256 * 3)        # trunk + vector head (approximately: skip layer 256 + 39 inputs)
VF_ALL = VF_VEC + 2.0 * 256 * 256
RN = 2.0 * (289 * 256 + 3 * 256 * 256 + 256 * 3)
timed("vector-only, gradient-free (vfn_vf_mlp16_fwd)", lambda: lib.vf_mlp16_fwd(g_vf, p_vf, pts), VF_VEC)
timed("vector-only, saving (vfn_vf_mlp16_fwd_train_at, no features)", lambda: lib.vf_mlp16_fwd_train(g_vf, p_vf, pts, False, ws.saved, ws.aux_vf, ws.masks, flags), VF_VEC)
timed("with features, saving", lambda: lib.vf_mlp16_fwd_train(g_vf, p_vf, pts, True, ws.saved, ws.aux_vf, ws.masks, flags), VF_ALL)
timed("fused with the rendering net, gradient-free", lambda: lib.vf_render_fused16_fwd(g_vf, p_vf, g_rn, p_rn, pts, dirs, 64), VF_ALL + RN)
timed("fused with the rendering net, saving", lambda: lib.vf_render_fused16_fwd_train(g_vf, p_vf, g_rn, p_rn, pts, dirs, 64, ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, flags), VF_ALL + RN)
