import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from helpers import load_fixture, build_model, loss_coefficients
DEV = "cuda:0"
for name in ("c1_det", "bench_sizes"):
    for act, grad in (("f16", "f16"), ("fp32", "f16"), ("f16", "fp32")):
        fx, d = load_fixture(name)
        g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
        model = build_model(fx, d, device=DEV)
        model.activation_storage, model.gradient_storage = act, grad
        out = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g})
        a, b, c = [t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape)]
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
        model.optimizer.zero_grad()
        loss.backward()
        bad = []
        for tag, net in (("vf", model.vector_field_network), ("rn", model.rendering_network)):
            for k, p in net.named_parameters():
                nf = int((~torch.isfinite(p.grad)).sum())
                if nf:
                    rows = (~torch.isfinite(p.grad)).reshape(p.grad.shape[0], -1).any(dim=1).nonzero().reshape(-1).tolist()
                    bad.append((tag, k, nf, rows[:8], len(rows)))
        print(name, act, grad, "non-finite:", bad if bad else "none")
