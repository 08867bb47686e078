"""BASELINE.json's full sizes (4096-ray chunks x 128 samples, the shipped 100 + {35..100} schedule) checked through
size-independent properties, plus the edge cases the path has: empty batches, ragged tails, the sample-count limit."""
import pytest
import torch

import vf_nerf_amd
from vf_nerf_amd import lib, synthetic

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(n_samples=64, n_importance=64, perturb=True, max_samples=100):
    torch.manual_seed(0)
    cfg = vf_nerf_amd.shipped_config(torch.device(DEV), n_samples=n_samples, n_importance=n_importance, perturb=perturb,
                                     dir_to_normal_th=-0.2, max_samples=max_samples)
    m = vf_nerf_amd.VectorFieldNerf(cfg)
    m.eval()
    synthetic.scale_hidden_weights(m.vector_field_network, m.rendering_network, 2.0)
    with torch.no_grad():
        pts = synthetic.frustum_points(20000, seed=1234).to(DEV)
        mean, std = synthetic.vector_head_stats_from_tanh(m.vector_field_network(pts, vector_only=True))
        synthetic.recentre_vector_head(m.vector_field_network, mean, std)
    return m


def test_full_chunk_properties_and_precision_agreement():
    m = _model()
    uv, pose, K = synthetic.pinhole_batch(4096, 1200, 680, 600.0, seed=9, device=DEV)
    g = torch.Generator().manual_seed(3)
    uni = {"u_coarse": torch.rand(4096, 64, generator=g), "u_fine": torch.rand(4096, 64, generator=g),
           "u_add": torch.rand(4096, 64, generator=g)}
    outs = {}
    with torch.no_grad():
        for prec in ("f16x3", "fp32"):
            m.precision = prec
            outs[prec] = m.render(pose, uv, K, 0, uniforms=uni)
        again = m.render(pose, uv, K, 0, uniforms=uni)
    a, b = outs["f16x3"], outs["fp32"]
    assert torch.equal(again.coarse_rgb_values, b.coarse_rgb_values), "same inputs, same draws -> bitwise same image"
    z = a.z_vals
    assert z.shape == (4096, 128) and bool((z[:, 1:] >= z[:, :-1]).all()), "fine samples are sorted per ray"
    assert float(a.coarse_normals.abs().max()) <= 1.0 and float(a.coarse_colors.min()) >= 0.0 and float(a.coarse_colors.max()) <= 1.0
    assert float(a.coarse_rgb_values.min()) >= -1e-6 and float(a.coarse_rgb_values.max()) <= 1.0 + 1e-5
    same = (a.z_vals == b.z_vals).all(dim=1)
    assert float(same.float().mean()) > 0.99, "proposal argmax agrees between the two arithmetic paths"
    err = (a.coarse_rgb_values - b.coarse_rgb_values).abs().max(dim=1)[0][same]
    derr = (a.coarse_depth_map - b.coarse_depth_map).abs().reshape(-1)[same]
    print(f"4096x128: identical sampling {float(same.float().mean()):.4f}; f16x3 vs fp32 max rgb {float(err.max()):.2e}, "
          f"depth {float(derr.max()):.2e}; rays hitting a surface {float((b.coarse_depth_map > 0).float().mean()):.2f}")
    assert float((err < 1e-4).float().mean()) > 0.999 and float((derr < 1e-4).float().mean()) > 0.999


def test_composite_is_linear_in_colours_and_weights_sum_below_one():
    m = _model()
    n, s = 4096, 128
    g = torch.Generator().manual_seed(5)
    normals = torch.nn.functional.normalize(torch.randn(n, s, 3, generator=g), dim=-1).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=-1).to(DEV)
    z = torch.sort(torch.rand(n, s, generator=g), dim=1)[0].to(DEV)
    c1 = torch.rand(n, s, 3, generator=g).to(DEV)
    c2 = torch.rand(n, s, 3, generator=g).to(DEV)
    scal = m.density.raw_scalars()
    dp = m._density_params
    _, w, _, r1, d1 = lib.ray_density_weights(dp(), normals, dirs, z, scal, colors=c1)
    _, _, _, r2, _ = lib.ray_density_weights(dp(), normals, dirs, z, scal, colors=c2)
    _, _, _, r12, _ = lib.ray_density_weights(dp(), normals, dirs, z, scal, colors=(0.25 * c1 + 0.5 * c2).contiguous())
    assert float((r12 - (0.25 * r1 + 0.5 * r2)).abs().max()) < 2e-6
    tot = w.sum(dim=1)
    assert float(tot.max()) <= 1.0 + 1e-5 and float(w.min()) >= 0.0
    assert float((d1.reshape(-1) - (w * z).sum(1)).abs().max()) < 2e-6


def test_shipped_schedule_sizes_and_ragged_tails():
    """100 coarse + min(N_samples, max_samples) fine (Q16): S_t = 135 and 200; ray counts that are not multiples of
    any tile size."""
    m = _model(n_samples=100, n_importance=35)
    for n_rays, n_fine in ((1000, 35), (777, 100), (1, 40)):
        m.fine_sampler.N_samples = n_fine
        uv, pose, K = synthetic.pinhole_batch(n_rays, 1200, 680, 600.0, seed=n_rays, device=DEV)
        with torch.no_grad():
            out = m.render(pose, uv, K, 0)
        s_t = 100 + n_fine
        assert out.z_vals.shape == (n_rays, s_t) and out.coarse_colors.shape == (n_rays * s_t, 3)
        assert bool(torch.isfinite(out.coarse_rgb_values).all()) and bool((out.z_vals[:, 1:] >= out.z_vals[:, :-1]).all())
    m.fine_sampler.N_samples = 500                       # capped by max_samples = 100
    with torch.no_grad():
        out = m.render(pose, uv, K, 0)
    assert out.z_vals.shape[1] == 200


def test_empty_batch_and_sample_limit():
    m = _model(n_samples=16, n_importance=16)
    uv, pose, K = synthetic.pinhole_batch(4, 64, 64, 60.0, seed=1, device=DEV)
    with torch.no_grad():
        out = m.render(pose[:0], uv[:0], K[:0], 0)
    assert out.coarse_rgb_values.shape == (0, 3) and out.z_vals.shape == (0, 32)
    assert m.vector_field_network(torch.zeros(0, 3, device=DEV)).shape == (0, 259)
    big = _model(n_samples=400, n_importance=200, max_samples=200)
    with torch.no_grad(), pytest.raises(lib.VfnError, match="bad sizes|outside"):
        big.render(pose, uv, K, 0)


def test_render_chunked_equals_chunk_by_chunk():
    """A view rendered in 1024-ray chunks on two alternating streams equals the same chunks rendered one after the other
    (deterministic sampling; the random extras of rays without a surface hit are replayed by resetting the Philox offset)."""
    import torch
    import bench
    dev = torch.device("cuda:0")
    model, _, _, _ = bench.build_scene(dev, 16, 32, 32, seed=0, perturb=False)
    from vf_nerf_amd import synthetic
    uv, pose, K = synthetic.pinhole_image(96, 64, 48.0, device=dev)
    n = uv.shape[0]
    with torch.no_grad():
        model._rng_offset = 0
        rgb2, depth2 = model.render_chunked(pose, uv, K, epoch=0, chunk=1024, n_streams=2)
        model._rng_offset = 0
        rgb1 = torch.empty(n, 3, device=dev)
        depth1 = torch.empty(n, 1, device=dev)
        for lo in range(0, n, 1024):
            o = model.render(pose[lo:lo + 1024], uv[lo:lo + 1024], K[lo:lo + 1024], epoch=0)
            rgb1[lo:lo + 1024], depth1[lo:lo + 1024] = o.coarse_rgb_values, o.coarse_depth_map
    torch.cuda.synchronize()
    assert torch.equal(rgb1, rgb2) and torch.equal(depth1, depth2)
    assert float(rgb1.abs().sum()) > 0
