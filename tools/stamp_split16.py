"""Prologue / total shader cycles per workgroup of the split inference launches (VF + blocks out, rendering from blocks),
from a -DVFN16_STAMPS build (tools/build_variants.sh "stamps:-DVFN16_STAMPS"; timing-only: outputs are overwritten).

    python tools/stamp_split16.py vf_nerf_amd/csrc/libvfn_stamps.so"""
import sys, ctypes as C, torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
m = 262144
pts = (torch.rand(m, 3, device=dev) * 2 - 1)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
l = C.CDLL(sys.argv[1]); l.vfn_last_error.restype = C.c_char_p
vecs = torch.empty(2 * m, 3, device=dev); blocks = torch.empty(2 * m, 1024, dtype=torch.uint8, device=dev)
vfw, rnw = vf.packed16_weights(), rn.packed16_weights()
for _ in range(10):
    rc = l.vfn_vf_feat16_fwd(C.byref(vf.geometry()), C.c_void_p(vfw.data_ptr()), C.c_void_p(pts.data_ptr()), C.c_int64(m),
                             C.c_void_p(vecs.data_ptr()), C.c_void_p(blocks.data_ptr()), stream)
    assert rc == 0, l.vfn_last_error()
torch.cuda.synchronize()
st = vecs[:m].view(-1, 128, 3)[:, 0, :].cpu().double()
print(f"VF + blocks out: prologue {st[:, 0].median():.0f} cycles, workgroup {st[:, 1].median():.0f} cycles = {st[:, 2].median() / 100:.1f} us "
      f"(clock {(st[:, 1] / st[:, 2] * 100).median():.0f} MHz); prologue share {(st[:, 0] / st[:, 1]).median() * 100:.1f} %")
n_rows = 2 * m
dst = torch.arange(n_rows, dtype=torch.int32, device=dev)
spts = torch.rand(n_rows, 3, device=dev); dirs = torch.nn.functional.normalize(torch.randn(n_rows // 128, 3, device=dev), dim=1)
normals = torch.empty(n_rows, 3, device=dev); colors = torch.empty(n_rows, 3, device=dev)
for _ in range(10):
    rc = l.vfn_render16_from_blocks(C.byref(rn.geometry()), C.c_void_p(rnw.data_ptr()), C.c_void_p(blocks.data_ptr()), C.c_void_p(vecs.data_ptr()),
                                    C.c_void_p(dst.data_ptr()), C.c_void_p(spts.data_ptr()), C.c_void_p(dirs.data_ptr()), C.c_int64(n_rows),
                                    C.c_int32(128), C.c_void_p(normals.data_ptr()), C.c_void_p(colors.data_ptr()), stream)
    assert rc == 0, l.vfn_last_error()
torch.cuda.synchronize()
st = colors.view(-1, 128, 3)[:, 0, :].cpu().double()
print(f"rendering from blocks: prologue {st[:, 0].median():.0f} cycles, workgroup {st[:, 1].median():.0f} cycles = {st[:, 2].median() / 100:.1f} us "
      f"(clock {(st[:, 1] / st[:, 2] * 100).median():.0f} MHz); prologue share {(st[:, 0] / st[:, 1]).median() * 100:.1f} %")
