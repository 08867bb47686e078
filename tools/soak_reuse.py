"""Soak check of the inference paths over odd sizes: the one-call render (vfn_render_fwd, five merged launches), the same pipeline
launch by launch from Python, the eight-launch plan from C, and the single fused launch over all sorted samples (no reuse of the
proposal evaluation) — bit-equality of every output for many (rays, S_c, N_f), with the colour branch on two and on three
products, including row counts that are not multiples of the 32-row groups."""
import sys, itertools, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
bad = 0
cases = list(itertools.product((1, 3, 31, 33, 100, 257, 1000, 4097), ((64, 64), (100, 35), (100, 100), (17, 9), (33, 2))))
for n, (s_c, n_f) in cases:
    model, uv, pose, K = bench.build_scene(dev, n, s_c, n_f, seed=n)
    ok, finite = True, True
    for products in (2, 3):
        model.colour_products = products
        outs = []
        for reuse, one_call, separate, streams in ((True, True, False, 2), (True, True, False, 1), (True, False, False, 1), (True, True, True, 1),
                                                   (False, False, False, 1)):
            model.reuse_proposal, model.one_call_render, model.render_separate_launches, model.render_streams = reuse, one_call, separate, streams
            model._rng_offset = 0
            with torch.no_grad():
                outs.append(model.render(pose, uv, K, epoch=0))
        a = outs[0]
        ok = ok and all(torch.equal(getattr(a, f), getattr(b, f)) for b in outs[1:]
                        for f in ("z_vals", "points_coarse", "coarse_normals", "coarse_colors", "coarse_rgb_values", "coarse_depth_map"))
        finite = finite and bool(torch.isfinite(a.coarse_rgb_values).all())
    if not (ok and finite):
        bad += 1
        print("MISMATCH", n, s_c, n_f, ok, finite)
torch.cuda.synchronize()
print(f"{len(cases)} cases, {bad} mismatches")
