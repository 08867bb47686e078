"""Interleaved A/B of the fused f16x3 launch with the colour branch on three and on two products (vfn_vf_render_fused16_products),
in ONE process on ONE GPU: times at the headline launch size (4096 rays x 128 samples), differences of the outputs, and the
errors of both against the exact-fp32 kernels and against the reference's golden outputs.

    python tools/ab_products.py > profiles/r02/ab_colour_products.txt"""
import os, sys, statistics, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import bench
from vf_nerf_amd import lib
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
with torch.no_grad():
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
args = (vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, 128)
res = {p: lib.vf_render_fused16_fwd(*args, colour_products=p) for p in (3, 2)}
n32, c32, _ = lib.vf_render_fused_fwd(vf.geometry(), vf.packed_weights(), rn.geometry(), rn.packed_weights(), pts, dirs, 128)
torch.cuda.synchronize()
print(f"{m} points (synthetic scene of bench.py)")
print("normals 2 vs 3 products: bit-identical" if torch.equal(res[2][0], res[3][0]) else "normals DIFFER: %.3e" % (res[2][0] - res[3][0]).abs().max().item())
for p in (3, 2):
    print(f"colour branch on {p} products: colours vs exact-fp32 kernels max {(res[p][1] - c32).abs().max().item():.2e}  rms {(res[p][1] - c32).pow(2).mean().sqrt().item():.2e}"
          f"   normals vs exact-fp32 max {(res[p][0] - n32).abs().max().item():.2e}")
times = {3: [], 2: []}
for rnd in range(10):
    for p in (3, 2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.vf_render_fused16_fwd(*args, colour_products=p)
        e1.record(); torch.cuda.synchronize()
        times[p].append(e0.elapsed_time(e1) / 5)
for p in (3, 2):
    t = times[p]
    print(f"colour branch on {p} products: median {statistics.median(t):.4f} ms  min {min(t):.4f}  max {max(t):.4f}   ({m / statistics.median(t) / 1e3:.1f} M samples/s)")
print(f"speed-up of the launch: {statistics.median(times[3]) / statistics.median(times[2]):.3f}x")
# golden fixtures (reference outputs)
from helpers import FIXTURE_NAMES, build_model, load_fixture
for name in FIXTURE_NAMES + ("attached_normals",):
    fx, d = load_fixture(name)
    mod = build_model(fx, d, "cuda:0")
    v, r = mod.vector_field_network, mod.rendering_network
    p = d["points"].reshape(-1, 3).to(dev).contiguous(); s_t = d["z_vals"].shape[1]
    dd = d["ray_dirs"].to(dev).contiguous()
    for k in (3, 2):
        nn, cc = lib.vf_render_fused16_fwd(v.geometry(), v.packed16_weights(), r.geometry(), r.packed16_weights(), p, dd, s_t, colour_products=k)
        rgb = (d["weights"].to(dev).unsqueeze(-1) * cc.reshape(-1, s_t, 3)).sum(1)
        print(f"{name:16s} {k} products: normals {(nn.cpu() - d['normals'].reshape(-1, 3)).abs().max().item():.2e}  colours {(cc.cpu() - d['colors']).abs().max().item():.2e}"
              f"  composited rgb {(rgb.cpu() - d['rgb']).abs().max().item():.2e}   (vs the reference's outputs)")
