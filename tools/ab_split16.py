"""Interleaved timing of the two split inference launches (vfn_vf_feat16_fwd, vfn_render16_from_blocks) across builds of
libvfn.so (tools/build_variants.sh):  python tools/ab_split16.py vf_nerf_amd/csrc/libvfn.so vf_nerf_amd/csrc/libvfn_<name>.so"""
import sys, ctypes as C, torch, statistics
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
m = 262144
pts = (torch.rand(m, 3, device=dev) * 2 - 1)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vecs = torch.empty(2 * m, 3, device=dev); blocks = torch.empty(2 * m, 1024, dtype=torch.uint8, device=dev)
vfw, rnw = vf.packed16_weights(), rn.packed16_weights()
n_rows = 2 * m
dst = torch.randperm(n_rows, device=dev).to(torch.int32)
spts = torch.rand(n_rows, 3, device=dev); dirs = torch.nn.functional.normalize(torch.randn(n_rows // 128, 3, device=dev), dim=1)
normals = torch.empty(n_rows, 3, device=dev); colors = torch.empty(n_rows, 3, device=dev)
libs = {}
for n in sys.argv[1:]:
    l = C.CDLL(n); l.vfn_last_error.restype = C.c_char_p; libs[n] = l
def feat(l):
    for half in (0, 1):
        rc = l.vfn_vf_feat16_fwd(C.byref(vf.geometry()), C.c_void_p(vfw.data_ptr()), C.c_void_p(pts.data_ptr()), C.c_int64(m),
                                 C.c_void_p(vecs.data_ptr() + half * m * 12), C.c_void_p(blocks.data_ptr() + half * m * 1024), stream)
        assert rc == 0, l.vfn_last_error()
def rend(l):
    rc = l.vfn_render16_from_blocks(C.byref(rn.geometry()), C.c_void_p(rnw.data_ptr()), C.c_void_p(blocks.data_ptr()), C.c_void_p(vecs.data_ptr()),
                                    C.c_void_p(dst.data_ptr()), C.c_void_p(spts.data_ptr()), C.c_void_p(dirs.data_ptr()), C.c_int64(n_rows),
                                    C.c_int32(128), C.c_void_p(normals.data_ptr()), C.c_void_p(colors.data_ptr()), stream)
    assert rc == 0, l.vfn_last_error()
times = {(n, k): [] for n in libs for k in ("feat x2", "render")}
for n, l in libs.items():
    feat(l); rend(l)
torch.cuda.synchronize()
for rnd in range(8):
    for n, l in libs.items():
        for k, fn in (("feat x2", feat), ("render", rend)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): fn(l)
            e1.record(); torch.cuda.synchronize()
            times[(n, k)].append(e0.elapsed_time(e1) / 4)
for (n, k), t in times.items():
    print(f"{n.split('/')[-1]:24s} {k:8s} median {statistics.median(t):.4f} ms  min {min(t):.4f}")
