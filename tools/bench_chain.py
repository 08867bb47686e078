"""The dX chain launches alone (csrc/vfn_bwd16.hip) on a workspace a saving forward filled: vector-only over the vector-field net (what region 1
of a training step runs) and fused over both nets (region 2 / the dense step), default 16-bit storages.
    python tools/bench_chain.py [points]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from vf_nerf_amd import lib  # noqa: E402
from vf_nerf_amd.backward import _Workspace, _head_rows, _packed_bwd16  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda", 0)
built = bench.build_trained_scene(dev, 4096, 64, 64, seed=0)
model = built[0] if built is not None else bench.build_scene(dev, 4096, 64, 64, seed=0)[0]
vf, rn = model.vector_field_network, model.rendering_network
g_vf, g_rn, p_vf, p_rn = vf.geometry(), rn.geometry(), vf.packed16_weights(), rn.packed16_weights()
b_vf, b_rn, h_vf, h_rn = _packed_bwd16(vf, False), _packed_bwd16(rn, False), _head_rows(vf), _head_rows(rn)
torch.manual_seed(0)
pts = (torch.rand(m, 3, device=dev) - 0.5) * 1.2
dirs = torch.nn.functional.normalize(torch.randn(m // 64, 3, device=dev), dim=-1)
ws = _Workspace(m, 13, dev, f16=True, frag=True, dy16="f16")
dy = ws.new_dy()
normals, colors = lib.vf_render_fused16_fwd_train(g_vf, p_vf, g_rn, p_rn, pts, dirs, 64, ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, ws.fwd_flags())[:2]
d_vec = torch.randn(m, 3, device=dev) * 1e-4
d_col = torch.randn(m, 3, device=dev) * 1e-4
dz_vec, dz_rgb = torch.empty(m, 4, device=dev), torch.empty(m, 4, device=dev)
VF_VEC = 2.0 * (6 * 256 * 256 + 295 * 256 + 256 * 3)        # the trunk's dX products (the first layer's inputs need no gradient) + the vector head
VF_ALL = VF_VEC + 2.0 * 256 * 256
RN = 2.0 * (256 * 256 + 3 * 256 * 256 + 256 * 3)


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
    print(f"{name:58s} {ms:7.3f} ms   {m * flop_per_point * 3 / ms / 1e9:7.1f} TFLOP/s executed (3 bf16 products)   ({m} points)")


timed("vector-only chain over the vector-field net", lambda: lib.mlp_bwd_chain_bf16_ws(g_vf, b_vf, h_vf, None, None, None, ws.feats(8), ws.masks, dy, ws.dy_flags(),
                                                                                       None, None, d_vec, normals, None, 3, m, None, dz_vec), VF_VEC)
timed("fused chain over both nets", lambda: lib.mlp_bwd_chain_bf16_ws(g_vf, b_vf, h_vf, g_rn, b_rn, h_rn, ws.feats(8), ws.masks, dy, ws.dy_flags(),
                                                                     d_col, colors, d_vec, normals, None, 3, m, dz_rgb, dz_vec), VF_ALL + RN)
