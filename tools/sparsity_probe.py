import sys, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for tag in ('random', 'trained'):
    if tag == 'random':
        model, uv, pose, K = bench.build_scene(dev, 4096, 64, 64, 0)
    else:
        model, uv, pose, K, what = bench.build_trained_scene(dev, 4096, 64, 64, 0)
    with torch.no_grad():
        model.one_call_render = False
        from vf_nerf_amd import lib
        out = model.render(pose, uv, K, 0)
        n, s_t = out.z_vals.shape
        rd = out.ray_dirs.reshape(n, s_t, 3)[:, 0].contiguous()
        _, w, _, _, _ = lib.ray_density_weights(model._density_params(), out.coarse_normals.reshape(-1, 3).contiguous(), rd, out.z_vals, model.density.raw_scalars(), colors=out.coarse_colors, want_sigma=False)
    pos = (w > 0)
    per_ray = pos.float().sum(1)
    print(tag, 'samples with w>0:', float(pos.float().mean()), ' w>1e-6:', float((w > 1e-6).float().mean()), ' w>1e-4:', float((w>1e-4).float().mean()),
          ' rays with any:', float((per_ray > 0).float().mean()), ' mean positive per hit ray:', float(per_ray[per_ray > 0].mean()))
    # groups of 32 consecutive sorted samples with any w>0
    g = pos.reshape(n, s_t // 32, 32).any(-1).float().mean()
    print('   groups of 32 (sorted order) with any w>0:', float(g))
