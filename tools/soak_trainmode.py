"""Soak check of the training-mode VF forward (batch-statistics BatchNorm + Jacobian rows) against the CPU oracle over odd row
counts (partial 128-row GEMM tiles, partial 64-row row-kernel blocks, M = 2: the smallest batch BatchNorm accepts)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import build_model, load_fixture
from oracle import vfnerf_oracle as O
fx, d = load_fixture("train_mode")
worst = 0.0
import os
for m in [int(v) for v in os.environ.get('SOAK_M', '2,5,33,129,257,1000,4099').split(',')]:
    gpu = build_model(fx, d, device="cuda:0"); gpu.train()
    cpu = build_model(fx, d)
    sd = {k: v.clone() for k, v in cpu.vector_field_network.state_dict().items()}
    g = torch.Generator().manual_seed(m)
    pts = torch.rand(m, 3, generator=g) * 2 - 1
    want = O.vf_mlp_train(pts.clone(), sd).detach()
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in cpu.vector_field_network.state_dict().items()}
    want64 = O.vf_mlp_train(pts.double().clone(), sd64).detach()
    o_net = float((want[:, :259].double() - want64[:, :259]).abs().max())
    o_jac = float((want[:, 259:].double() - want64[:, 259:]).abs().max() / want64[:, 259:].abs().max().clamp_min(1e-30))
    got = gpu.vector_field_network(pts.to("cuda:0")).detach().cpu()
    e_net = float((got[:, :259] - want[:, :259]).abs().max())
    e_jac = float((got[:, 259:] - want[:, 259:]).abs().max() / want[:, 259:].abs().max().clamp_min(1e-30))
    rm = gpu.vector_field_network.layers[3][1].running_var.cpu()
    e_rv = float((rm - sd["layers.3.1.running_var"]).abs().max())
    g_net = float((got[:, :259].double() - want64[:, :259]).abs().max())
    g_jac = float((got[:, 259:].double() - want64[:, 259:]).abs().max() / want64[:, 259:].abs().max().clamp_min(1e-30))
    print(f"M={m}: vs fp32 oracle: network {e_net:.2e} jacobian {e_jac:.2e} | vs float64: HIP {g_net:.2e} / {g_jac:.2e}, fp32 oracle {o_net:.2e} / {o_jac:.2e} | running_var {e_rv:.2e}")
    worst = max(worst, e_net)
print("worst network error", worst)
