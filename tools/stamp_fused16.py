"""Shader cycles and shader clock of the fused f16x3 kernel: per-workgroup s_memtime / s_memrealtime deltas written by a
-DVFN16_STAMPS build (tools/build_variants.sh "stamps:-DVFN16_STAMPS"; timing-only, the colours are overwritten).

    python tools/stamp_fused16.py vf_nerf_amd/csrc/libvfn_stamps.so"""
import sys, ctypes as C, torch, statistics
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
with torch.no_grad():
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
vfw, rnw = vf.packed16_weights(), rn.packed16_weights()
nw = torch.empty(m, 3, device=dev); cw = torch.empty(m, 3, device=dev)
for name in sys.argv[1:]:
    l = C.CDLL(name); l.vfn_last_error.restype = C.c_char_p
    for _ in range(20):
        rc = l.vfn_vf_render_fused16_fwd(C.byref(vf.geometry()), C.c_void_p(vfw.data_ptr()), C.byref(rn.geometry()), C.c_void_p(rnw.data_ptr()),
            C.c_void_p(pts.data_ptr()), C.c_void_p(dirs.data_ptr()), C.c_int64(m), C.c_int32(128), C.c_void_p(nw.data_ptr()), C.c_void_p(cw.data_ptr()), stream)
        assert rc == 0
    torch.cuda.synchronize()
    st = cw.view(-1, 128, 3)[:, 0, :2].cpu()            # first point of each workgroup
    cyc, real = st[:, 0].double(), st[:, 1].double()
    mhz = (cyc / real * 100.0)
    print(f"{name.split('/')[-1]}: per-WG cycles median {cyc.median():.0f} (min {cyc.min():.0f} max {cyc.max():.0f}); "
          f"duration median {real.median() / 100:.1f} us; shader clock median {mhz.median():.0f} MHz; cycles/chunk {cyc.median() / 105:.0f}")
