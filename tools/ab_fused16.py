"""Interleaved A/B timing of vfn_vf_render_fused16_fwd across several builds of libvfn.so in ONE process on ONE GPU
(devices of this pool differ by several percent, and the kernel is power-limited, so variants are only ever compared
inside one run).

    bash vf_nerf_amd/csrc/build.sh && tools/build_variants.sh "nodma:-DABL_NODMA" "fd3:-DVFN16_FDEPTH=3"
    python tools/ab_fused16.py vf_nerf_amd/csrc/libvfn.so vf_nerf_amd/csrc/libvfn_nodma.so vf_nerf_amd/csrc/libvfn_fd3.so

Build switches: see tools/build_variants.sh.  Timing-only ablations change the DATA the matrix cores see (zeros toggle
fewer wires), which raises the shader clock: compare their cycle counts (tools/stamp_fused16.py), not only their times."""
import sys, ctypes as C, torch, statistics
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
names = sys.argv[1:]
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
with torch.no_grad():
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
normals = torch.empty(m, 3, device=dev); colors = torch.empty(m, 3, device=dev)
ref_n, ref_c = lib.vf_render_fused16_fwd(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, 128,
                                         colour_products=int(__import__('os').environ.get('COLOUR_PRODUCTS', '2')))
libs = {}
for n in names:
    l = C.CDLL(n); l.vfn_last_error.restype = C.c_char_p; libs[n] = l
import os
PRODUCTS = int(os.environ.get("COLOUR_PRODUCTS", "2"))       # the shipped default of gradient-free renders; 3: three products everywhere
def call(l):
    rc = l.vfn_vf_render_fused16_products(C.byref(vf.geometry()), C.c_void_p(vf.packed16_weights().data_ptr()), C.byref(rn.geometry()),
        C.c_void_p(rn.packed16_weights().data_ptr()), C.c_void_p(pts.data_ptr()), C.c_void_p(dirs.data_ptr()), C.c_int64(m), C.c_int32(128),
        None, C.c_int32(PRODUCTS), C.c_void_p(normals.data_ptr()), C.c_void_p(colors.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, l.vfn_last_error()
times = {n: [] for n in names}
for n in names:
    call(libs[n]); torch.cuda.synchronize()
    err = (colors - ref_c).abs().max().item(), (normals - ref_n).abs().max().item()
    print(n.split('/')[-1], "max diff vs default build: colors %.2e normals %.2e" % err)
for rnd in range(8):
    for n in names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call(libs[n])
        e1.record(); torch.cuda.synchronize()
        times[n].append(e0.elapsed_time(e1) / 5)
for n in names:
    t = times[n]
    print(f"{n.split('/')[-1]:28s} median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}")
