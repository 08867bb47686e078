"""Feasibility of overlapping the HBM-bound weight-gradient launches of one part of the batch with the matrix-bound dX chain of the
next part on a second stream (DESIGN.md section 6f lead).  Same kernels, same workspace, regions via ws_first / first_point:

    sequential   chain(all) ; dW_rn(all) ; dW_vf(all)                                   (what the step does today)
    parts = P    chain(part 0) ; [side stream: dW(part 0)] || chain(part 1) ; ... ; dW(last part) on the main stream

    python tools/ab_overlap.py [rays]
"""
import sys, statistics, torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import lib
from vf_nerf_amd.backward import _Workspace, _entries, _packed_bwd16, _head_rows, _layer_table, _ensure_grads

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device('cuda:0')
model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, 0)
vf, rn = model.vector_field_network, model.rendering_network
model.optimizer.zero_grad()
_ensure_grads(vf); _ensure_grads(rn)
with torch.no_grad():
    out = model.render(pose, uv, K, 0)
pts = out.points_coarse.reshape(-1, 3).contiguous(); dirs = out.ray_dirs[::128].contiguous()
m = pts.shape[0]
vf_h, rn_h = len(_entries(vf)), len(_entries(rn))
ws = _Workspace(m, vf_h + rn_h, dev, f16=True, frag=True, dy16="f16")
normals, colors = lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, 128,
                                                   ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags())
g = torch.Generator().manual_seed(1)
dc = (torch.randn(m, 3, generator=g) * 1e-5).to(dev); dn = (torch.randn(m, 3, generator=g) * 1e-6).to(dev)
dy = ws.new_dy(); zr = torch.empty(m, 4, device=dev); zv = torch.empty(m, 4, device=dev)
feats = ws.feats(vf_h - 1)
forms = ws.frag_forms()
vfw, rnw, vfhd, rnhd = _packed_bwd16(vf), _packed_bwd16(rn), _head_rows(vf).contiguous(), _head_rows(rn).contiguous()
tab_vf, tab_rn = _layer_table(vf), _layer_table(rn)


def scratch(net, count):
    return torch.empty(lib.net_weight_grads_scratch_bytes(net._kind, net.geometry(), count), dtype=torch.uint8, device=dev)


def chain(first, count):
    lib.mlp_bwd_chain_bf16_ws(vf.geometry(), vfw, vfhd, rn.geometry(), rnw, rnhd, feats, ws.masks, dy, ws.dy_flags(), dc[first:first + count],
                              colors[first:first + count], dn[first:first + count], normals[first:first + count], None, 3, count, zr, zv,
                              ws_first=first, ws_points=m)


def dw(first, count, s_rn, s_vf):
    lib.net_weight_grads_frag(rn._kind, rn.geometry(), tab_rn, ws.saved, vf_h, dy, ws.slot_floats, forms[0], forms[1], feats, ws.aux_rn, zr, count,
                              True, True, s_rn, first_point=first)
    lib.net_weight_grads_frag(vf._kind, vf.geometry(), tab_vf, ws.saved, 0, dy, ws.slot_floats, forms[0], forms[1], None, ws.aux_vf, zv, count,
                              True, True, s_vf, first_point=first)


side = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)
full = (scratch(rn, m), scratch(vf, m))


def sequential():
    chain(0, m)
    dw(0, m, *full)


def overlapped(parts):
    bounds = [(m * i // parts) // 128 * 128 for i in range(parts)] + [m]
    scr = [(scratch(rn, bounds[i + 1] - bounds[i]), scratch(vf, bounds[i + 1] - bounds[i])) for i in range(parts)]

    def run():
        for i in range(parts):
            first, count = bounds[i], bounds[i + 1] - bounds[i]
            chain(first, count)
            if i + 1 < parts:
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    dw(first, count, *scr[i])      # accumulates into .grad: the parts' un-folds are serialised on the side stream
            else:
                main.wait_stream(side)
                dw(first, count, *scr[i])
    return run


variants = {"sequential": sequential, "2 parts": overlapped(2), "3 parts": overlapped(3), "4 parts": overlapped(4)}
times = {k: [] for k in variants}
for f in variants.values():
    f(); torch.cuda.synchronize()
for rnd in range(6):
    for k, f in variants.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            f()
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 3)
print(f"{rays} rays x 128: chain + all weight gradients of the fine pass, {m} points")
for k, t in times.items():
    print(f"  {k:12s} median {statistics.median(t):.4f} ms  min {min(t):.4f}")
