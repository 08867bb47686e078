"""Interleaved timing of vfn_vf_render_fused16_fwd_train (activation-saving forward) across builds of libvfn.so."""
import sys, ctypes as C, torch, statistics
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
from vf_nerf_amd.backward import _Workspace, _entries
names = sys.argv[1:]
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
with torch.no_grad():
    model.reuse_proposal = False
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
ws = _Workspace(m, len(_entries(vf)) + len(_entries(rn)), dev, f16=True, frag=True)       # the default storage: f16 activations, fragment order
normals = torch.empty(m, 3, device=dev); colors = torch.empty(m, 3, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = {}
for n in names:
    l = C.CDLL(n); l.vfn_last_error.restype = C.c_char_p; libs[n] = l
def call(l):
    rc = l.vfn_vf_render_fused16_fwd_train(C.byref(vf.geometry()), C.c_void_p(vf.packed16_weights().data_ptr()), C.byref(rn.geometry()),
        C.c_void_p(rn.packed16_weights().data_ptr()), C.c_void_p(pts.data_ptr()), C.c_void_p(dirs.data_ptr()), C.c_int64(m), C.c_int32(128),
        C.c_void_p(normals.data_ptr()), C.c_void_p(colors.data_ptr()), C.c_void_p(ws.saved.data_ptr()), C.c_void_p(ws.aux_vf.data_ptr()),
        C.c_void_p(ws.aux_rn.data_ptr()), C.c_void_p(ws.masks.data_ptr()), C.c_int32(ws.fwd_flags()), stream)
    assert rc == 0, l.vfn_last_error()
times = {n: [] for n in names}
ref = None
for n in names:
    ws.saved.zero_(); ws.masks.zero_()
    call(libs[n]); torch.cuda.synchronize()
    got = (ws.saved.clone(), ws.masks.clone(), normals.clone(), colors.clone())
    if ref is None:
        ref = got
    else:
        print(n.split('/')[-1], "workspace / masks / outputs identical to the first build:", all(torch.equal(a, b) for a, b in zip(ref, got)))
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
