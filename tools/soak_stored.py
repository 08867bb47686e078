"""Soak check of the training render with one VF evaluation per distinct sample (backward.StoredFinePass) against the sorted fine
pass with its separate proposal launch, over odd sizes: forward outputs bit-equal, parameter gradients equal up to the order of
the sums over points, for both gradient storages; sizes whose proposal block is not whole groups of 32 points must fall back."""
import sys, itertools, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
bad = 0
cases = list(itertools.product((1, 2, 32, 33, 100, 257, 1024), ((64, 64), (100, 35), (32, 17), (17, 9), (96, 2))))
for n, (s_c, n_f) in cases:
    g = torch.Generator().manual_seed(n)
    rgb_gt = torch.rand(n, 3, generator=g).to(dev)
    res = {}
    for grads in ("fp32", "f16"):
        for stored in (True, False):
            model, uv, pose, K = bench.build_scene(dev, n, s_c, n_f, seed=n)
            model.gradient_storage, model.reuse_proposal_training = grads, stored
            model._keep_saved, model._debug_dst = True, None
            model.optimizer.zero_grad()
            out = model.render(pose, uv, K, epoch=0)
            loss = (out.coarse_rgb_values - rgb_gt).abs().mean() + 0.1 * ((out.coarse_normals.norm(dim=-1) - 1) ** 2).mean() + out.coarse_depth_map.mean()
            loss.backward()
            res[grads, stored] = (out, [p.grad.detach().clone() for p in model.unique_parameters()], model._debug_dst is not None)
    for grads in ("fp32", "f16"):
        (o1, g1, used), (o0, g0, _) = res[grads, True], res[grads, False]
        expect = (n * s_c) % 32 == 0
        same = all(torch.equal(getattr(o1, f), getattr(o0, f)) for f in ("z_vals", "coarse_normals", "coarse_colors", "coarse_rgb_values", "coarse_depth_map"))
        worst = max(float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for a, b in zip(g1[:-3], g0[:-3]))      # (the 3 density scalars: float atomics)
        tol = (1e-4 if grads == "f16" else 2e-5) if used else 1e-12       # f16 storage: the per-lane scales see different neighbours in a different order
        if not (same and used == expect and worst < tol):
            bad += 1
            print("MISMATCH", n, s_c, n_f, grads, "forward equal", same, "stored used", used, "expected", expect, f"worst gradient difference {worst:.2e}")
torch.cuda.synchronize()
print(f"{len(cases)} cases x 2 storages, {bad} mismatches")
