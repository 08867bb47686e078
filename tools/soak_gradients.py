"""Soak check of the training backward over odd batch sizes: parameter gradients of a fixed functional with the 16-bit
matrix-core path (default and f16 activation storage) against the exact-fp32 kernels, same weights / rays / draws.  Sizes hit
ragged 128-point tiles, odd slab counts of the weight-gradient kernels and single-ray batches."""
import sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
worst = {}
for n, (s_c, n_f) in [(1, (64, 64)), (3, (17, 9)), (33, (64, 64)), (100, (100, 35)), (257, (64, 64)), (1000, (33, 31))]:
    grads = {}
    for tag, prec, act in (("fp32", "fp32", "fp32"), ("f16x3", "f16x3", "fp32"), ("f16act", "f16x3", "f16")):
        model, uv, pose, K = bench.build_scene(dev, n, s_c, n_f, seed=n)
        model.precision, model.activation_storage = prec, act
        model._rng_offset = 0
        g = torch.Generator().manual_seed(n)
        a, b = torch.randn(n, 3, generator=g).to(dev), torch.randn(n, 1, generator=g).to(dev)
        c = (0.05 * torch.randn(n, s_c + n_f, 3, generator=g)).to(dev)
        out = model.render(pose, uv, K, epoch=0)
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
        loss.backward()
        grads[tag] = {k: p.grad.detach().clone() for k, p in list(model.vector_field_network.named_parameters()) + list(model.rendering_network.named_parameters())}
        assert all(bool(torch.isfinite(v).all()) for v in grads[tag].values()), (n, tag)
    for tag in ("f16x3", "f16act"):
        e = max(float((grads[tag][k] - grads["fp32"][k]).abs().max() / grads["fp32"][k].abs().max().clamp_min(1e-30)) for k in grads["fp32"])
        worst[(n, tag)] = e
    print(f"n={n} ({s_c}+{n_f}): max rel gradient difference to the fp32 kernels: f16x3 {worst[(n, 'f16x3')]:.2e}, f16 activations {worst[(n, 'f16act')]:.2e}")
