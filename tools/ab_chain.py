"""Interleaved timing of the fused dX chain (vfn_mlp_bwd_chain_bf16_ws, scaled f16 gradients) across builds of libvfn.so:
    python tools/ab_chain.py vf_nerf_amd/csrc/libvfn.so vf_nerf_amd/csrc/libvfn_<variant>.so ..."""
import sys, ctypes as C, torch, statistics
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
from vf_nerf_amd.backward import _Workspace, _entries, _packed_bwd16, _head_rows
names = sys.argv[1:]
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
with torch.no_grad():
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
ws = _Workspace(m, len(_entries(vf)) + len(_entries(rn)), dev, f16=True, frag=True, dy16="f16")
normals, colors = lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, 128,
                                                   ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags())
g = torch.Generator().manual_seed(1)
dc = (torch.randn(m, 3, generator=g) * 1e-5).to(dev); dn = (torch.randn(m, 3, generator=g) * 1e-6).to(dev)
dy = ws.new_dy(); zr = torch.empty(m, 4, device=dev); zv = torch.empty(m, 4, device=dev)
feats = ws.feats(8)
a = dict(vfw=_packed_bwd16(vf), rnw=_packed_bwd16(rn), vfh=_head_rows(vf).contiguous(), rnh=_head_rows(rn).contiguous())
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
libs = {}
for n in names:
    l = C.CDLL(n); l.vfn_last_error.restype = C.c_char_p; libs[n] = l
def call(l):
    rc = l.vfn_mlp_bwd_chain_bf16_ws(C.byref(vf.geometry()), P(a["vfw"]), P(a["vfh"]), C.byref(rn.geometry()), P(a["rnw"]), P(a["rnh"]), P(feats), P(ws.masks),
                                     P(dy), C.c_int32(ws.dy_flags()), P(dc), P(colors), P(dn), P(normals), None, C.c_int32(3), C.c_int64(m), P(zr), P(zv), stream)
    assert rc == 0, l.vfn_last_error()
times = {n: [] for n in names}
for n in names:
    call(libs[n]); torch.cuda.synchronize()
for rnd in range(6):
    for n in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): call(libs[n])
        e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 3)
for n in names:
    t = times[n]
    print(f"{n.split('/')[-1]:28s} median {statistics.median(t):.4f} ms  min {min(t):.4f}")
