#!/usr/bin/env python3
"""SURVEY.md section 8(e) caveat, measured: the supervision term is a mean over a DATA-DEPENDENT number of rows per shard (2 x n_sup sampled points + the
ray samples inside the centre ball, models/helpers/functions.py:137-157), so with data-parallel ranks the average of the shard means is not the mean over the
global batch (what the reference's nn.DataParallel computes on GPU 0).  On the trained scene of tests/golden/trained_far.npz, 512 rays x (64 + 64), CPU oracle:
    python tools/shard_mean_supervision.py > profiles/r06/shard_mean_supervision.txt"""
import sys, torch, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
from helpers import load_fixture, build_model, oracle_settings
from oracle import vfnerf_oracle as O
from vf_nerf_amd import synthetic
torch.set_num_threads(8)
fx, d = load_fixture("trained_far")
model = build_model(fx, d)
vf_sd = {k: v.detach() for k, v in model.vector_field_network.state_dict().items()}
rn_sd = {k: v.detach() for k, v in model.rendering_network.state_dict().items()}
n = 512
uv, pose, K = synthetic.pinhole_batch(n, 64, 64, 60.0, seed=5, pose=synthetic.orbit_pose(-35.0, 5.0, 0.9))
g = torch.Generator().manual_seed(1)
st = oracle_settings(fx)
uni = dict(u_coarse=torch.rand(n, st.n_samples, generator=g), u_fine=torch.rand(n, st.n_fine, generator=g), u_add=torch.rand(n, st.n_fine, generator=g))
with torch.no_grad():
    out = O.render(uv, pose, K, vf_sd, rn_sd, st, **uni)
cen = torch.tensor([0.0, 0.0, 0.55]); radius = 0.15
pts, nrm = out["points"], out["normals"].reshape(out["points"].shape)
for world in (2, 8):
    per = n // world
    means, rows = [], []
    allp, allg = [], []
    for r in range(world):
        sl = slice(r * per, (r + 1) * per)
        s_t = pts.shape[1]
        n_sup = (per * s_t) // 10
        bu = torch.rand(n_sup, 3, generator=g, dtype=torch.float64); cu = torch.rand(n_sup, 3, generator=g, dtype=torch.float64)
        bp, bgt = O.sphere_shell_points(bu, 1.0 - 5 * radius, 1.0, cen, True)
        cp, cgt = O.sphere_shell_points(cu, 0.0, radius, cen, False)
        with torch.no_grad():
            bpred = O.vf_mlp(bp.float(), vf_sd)[:, :3]; cpred = O.vf_mlp(cp.float(), vf_sd)[:, :3]
        rp, rgt = O.center_indices_and_gt(pts[sl], nrm[sl], cen, radius)
        pred = torch.cat([bpred, rp, cpred]); gt = torch.cat([bgt.float(), rgt, cgt.float()])
        means.append(float(torch.nn.functional.mse_loss(pred, gt))); rows.append(pred.shape[0])
        allp.append(pred); allg.append(gt)
        print(world, r, "centre-ball rows", rp.shape[0], "of", per * s_t, "samples; supervised rows", pred.shape[0], "mse", means[-1])
    glob = float(torch.nn.functional.mse_loss(torch.cat(allp), torch.cat(allg)))
    print(world, "mean of shard means", np.mean(means), "global", glob, "rel diff", abs(np.mean(means) - glob) / glob, "row weights", [round(r * world / sum(rows), 4) for r in rows])
