"""Diagnostic: how much does the parameter gradient of a fixture move when the WEIGHTS move by what f16 rounding moves them (default
three-product kernels throughout)?  Puts the single-product mode's gradient difference on bench_sizes into proportion."""
import sys, torch
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from helpers import build_model, load_fixture
from test_hip_single_product import _step
for name in ("bench_sizes", "c1_perturb"):
    fx, d = load_fixture(name)
    model = build_model(fx, d, device="cuda:0")
    model.reuse_proposal_training = False
    l3, o3, g3 = _step(model, d, 3)
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.unique_parameters():
            if p.dim() == 2:
                p.mul_(1.0 + (torch.rand(p.shape, generator=gen).to(p.device) - 0.5) * 2.0 ** -10)     # uniform in +-2^-11: an f16 rounding
    for net in (model.vector_field_network, model.rendering_network):
        net.invalidate_packs() if hasattr(net, "invalidate_packs") else None
    l3p, o3p, g3p = _step(model, d, 3)
    same = (o3p.z_vals == o3.z_vals).all(dim=1)
    print(name, "same samples", int(same.sum()), "/", same.numel(), "max normal diff", float((o3p.coarse_normals - o3.coarse_normals).abs().max()))
    for tag in ("vf", "rn"):
        cos = float(torch.dot(g3p[tag], g3[tag]) / (g3p[tag].norm() * g3[tag].norm()))
        print("   ", tag, "three products, weights perturbed by 2^-11: cos", round(cos, 5), "ratio", round(float(g3p[tag].norm() / g3[tag].norm()), 4))
