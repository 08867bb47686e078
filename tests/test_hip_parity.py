"""HIP kernels (through the C ABI / ctypes) against the CPU oracle and the reference's golden vectors.

Stage tests inject the reference's own intermediates so that one stage's rounding cannot flip a
discontinuity (argmax, mask, sort) of a later one; the end-to-end test then accounts for such flips per ray.
Tolerances: bit-exact for sample indices and z values; 1e-4 relative (max|a-b| / max(1, max|b|)) for fp32
values as BASELINE.json's north_star states — the observed errors are ~1e-6 and the tighter bounds below
assert that."""
import os

import pytest
import torch

from helpers import FIXTURE_NAMES, build_model, load_fixture, oracle_settings, rel_err
from oracle import vfnerf_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4        # the contract
TIGHT = 2e-5      # what exact-fp32 MFMA + fp32 transcendentals actually deliver


def dev():
    return torch.device("cuda:0")


def to_dev(d):
    return {k: v.to(dev()) for k, v in d.items()}


@pytest.fixture(scope="module", params=FIXTURE_NAMES)
def case(request):
    from vf_nerf_amd import lib
    lib.load()
    fx, d = load_fixture(request.param)
    model = build_model(fx, d, device="cuda:0")
    return fx, d, to_dev(d), model


def test_library_is_the_hip_build():
    from vf_nerf_amd import lib
    l = lib.load()
    assert l.vfn_abi_version() == lib.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libvfn.so" in f.read()


def test_cpu_tensors_are_rejected_loudly():
    from vf_nerf_amd import lib
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    vf = model.vector_field_network
    with pytest.raises(lib.VfnError):
        lib.vf_mlp_fwd(vf.geometry(), vf.packed_weights(), torch.zeros(8, 3), 3)


def test_raygen_and_coarse_sampler(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    far_t = g["far_per_ray"].reshape(-1).contiguous() if "far_per_ray" in g else None
    t_vals = torch.linspace(0., 1., steps=fx["n_samples"]).to(dev())
    directions, ray_dirs, cam_loc, z, pts = lib.raygen_uniform(g["uv"], g["pose"], g["intrinsics"], t_vals,
                                                               fx["n_samples"], fx["near"], fx["far"], far_t,
                                                               g.get("u_coarse"))
    assert torch.equal(cam_loc.cpu(), d["cam_loc"])
    assert torch.equal(z.cpu(), d["z_coarse"]), "coarse z values must be bit-exact"
    assert rel_err(directions, d["directions"]) < 1e-6
    assert rel_err(ray_dirs, d["ray_dirs"]) < 1e-6
    ref_pts = d["cam_loc"][:, None, :] + d["z_coarse"][:, :, None] * d["directions"][:, None, :]
    assert rel_err(pts, ref_pts) < 1e-6


def test_vf_mlp_vector_only(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    pts = (g["cam_loc"][:, None, :] + g["z_coarse"][:, :, None] * g["directions"][:, None, :]).reshape(-1, 3).contiguous()
    vf = model.vector_field_network
    out = lib.vf_mlp_fwd(vf.geometry(), vf.packed_weights(), pts, 3)
    err = rel_err(out, d["normals_coarse"].reshape(-1, 3))
    print(f"vf vector-only rel err {err:.3e}")
    assert err < TIGHT


def test_vf_mlp_full_row(case):
    from oracle import vfnerf_oracle as O
    fx, d, g, model = case
    pts = g["points"].reshape(-1, 3).contiguous()
    with torch.no_grad():
        out = model.vector_field_network(pts)
    cpu_sd = {k: v.cpu() for k, v in model.vector_field_network.state_dict().items()}
    ref = O.vf_mlp(d["points"].reshape(-1, 3), cpu_sd)
    assert out.shape == ref.shape == (pts.shape[0], 259)
    err_n, err_f = rel_err(out[:, :3], ref[:, :3]), rel_err(out[:, 3:], ref[:, 3:])
    print(f"vf full rel err normals {err_n:.3e} feats {err_f:.3e}")
    assert err_n < TIGHT and err_f < TIGHT
    assert rel_err(out[::8, 3:], d["feats_sub"]) < TIGHT     # the reference's own features
    assert rel_err(out[:, :3], d["normals"].reshape(-1, 3)) < TIGHT


def test_density_weights_argmax_coarse(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    sigma, w, imax, _, _ = lib.ray_density_weights(model._density_params(), g["normals_coarse"].contiguous(),
                                                   g["ray_dirs"].contiguous(), g["z_coarse"].contiguous(),
                                                   model.density.raw_scalars(), want_argmax=True)
    es, ew = rel_err(sigma, d["sigma_coarse"]), rel_err(w, d["weights_coarse"])
    print(f"coarse sigma rel err {es:.3e} weights {ew:.3e}")
    assert es < TIGHT and ew < TIGHT
    assert torch.equal(imax.cpu(), d["max_indices"]), "argmax of the proposal weights must be bit-exact"


def test_range_fine_sampler_bit_exact(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    far_t = g["far_per_ray"].reshape(-1).contiguous() if "far_per_ray" in g else None
    z, pts = lib.range_fine_sample(g["z_coarse"].contiguous(), g["max_indices"].contiguous(),
                                   g["directions"].contiguous(), g["cam_loc"].contiguous(), fx["n_importance"],
                                   fx["near"], fx["far"], fx["fine_range"], g["u_add"].contiguous(),
                                   g.get("u_fine"), far_t)
    assert torch.equal(z.cpu(), d["z_vals"]), "fine z values (sorted) must be bit-exact"
    assert torch.equal(pts.cpu(), d["points"]), "fine points must be bit-exact given identical directions"


def test_fused_fine_pass(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    vf, rn = model.vector_field_network, model.rendering_network
    s_t = fx["n_samples"] + fx["n_importance"]
    normals, colors, feats = lib.vf_render_fused_fwd(vf.geometry(), vf.packed_weights(), rn.geometry(),
                                                     rn.packed_weights(), g["points"].reshape(-1, 3).contiguous(),
                                                     g["ray_dirs"].contiguous(), s_t, want_feats=True)
    en, ec = rel_err(normals, d["normals"].reshape(-1, 3)), rel_err(colors, d["colors"])
    ef = rel_err(feats[::8], d["feats_sub"])
    print(f"fused normals {en:.3e} colors {ec:.3e} feats {ef:.3e}")
    assert en < TIGHT and ec < TIGHT and ef < TIGHT


def test_render_mlp_alone(case):
    from oracle import vfnerf_oracle as O
    fx, d, g, model = case
    s_t = fx["n_samples"] + fx["n_importance"]
    cpu_vf = {k: v.cpu() for k, v in model.vector_field_network.state_dict().items()}
    ref_vf = O.vf_mlp(d["points"].reshape(-1, 3), cpu_vf)
    rep = d["ray_dirs"].unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
    with torch.no_grad():
        colors = model.rendering_network(g["points"].reshape(-1, 3), ref_vf[:, :3].to(dev()), rep.to(dev()),
                                         ref_vf[:, 3:].contiguous().to(dev()))
    err = rel_err(colors, d["colors"])
    print(f"render-only colors rel err {err:.3e}")
    assert err < TIGHT


def test_composite_with_reference_inputs(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    sigma, w, _, rgb, depth = lib.ray_density_weights(model._density_params(), g["normals"].contiguous(),
                                                      g["ray_dirs"].contiguous(), g["z_vals"].contiguous(),
                                                      model.density.raw_scalars(), colors=g["colors"].contiguous())
    errs = dict(sigma=rel_err(sigma, d["sigma"]), weights=rel_err(w, d["weights"]), rgb=rel_err(rgb, d["rgb"]),
                depth=rel_err(depth, d["depth"]))
    print("composite rel errs", {k: f"{v:.3e}" for k, v in errs.items()})
    assert max(errs.values()) < TIGHT
    assert depth.shape == (d["depth"].shape[0], 1)


def _account(out, d, label):
    """Per-ray accounting of a render against a reference fixture -> dict of measured figures (printed by the callers):
    fraction of rays whose sorted sample depths are bit-identical, fraction of those inside 1e-4 (rgb, depth), the worst
    absolute errors, and the worst RELATIVE error over entries with |ref| > 1e-2 (an absolute bound alone says little about
    values in [0, 1])."""
    same_z = (out.z_vals.cpu() == d["z_vals"]).all(dim=1)
    rgb, depth = out.coarse_rgb_values.cpu(), out.coarse_depth_map.cpu()
    rgb_err = (rgb - d["rgb"]).abs().max(dim=1)[0]
    dep_err = (depth - d["depth"]).abs().reshape(-1)
    ok = (rgb_err < TOL) & (dep_err < TOL * max(1.0, float(d["depth"].abs().max())))
    g = same_z
    nrm_err = (out.coarse_normals.cpu() - d["normals"]).abs().amax(dim=(1, 2))
    col_err = (out.coarse_colors.cpu().reshape(d["z_vals"].shape[0], -1, 3) - d["colors"].reshape(d["z_vals"].shape[0], -1, 3)).abs().amax(dim=(1, 2))

    def worst_rel(a, b):
        big = b.abs() > 1e-2
        return float(((a - b).abs()[big] / b.abs()[big]).max()) if bool(big.any()) else 0.0

    res = dict(frac_same_z=float(same_z.float().mean()), n_rays=int(same_z.numel()), n_diff_z=int((~same_z).sum()),
               frac_within_tol=float(ok[g].float().mean()) if bool(g.any()) else 0.0,
               max_rgb=float(rgb_err[g].max()), max_depth=float(dep_err[g].max()), max_normals=float(nrm_err[g].max()),
               max_colors=float(col_err[g].max()), rel_rgb=worst_rel(rgb[g], d["rgb"][g]), rel_depth=worst_rel(depth[g], d["depth"][g]))
    print(label, {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in res.items()})
    return res


def test_render_end_to_end(case):
    """Whole render() (the DEFAULT path: f16x3 with three products everywhere, one C call) with the reference's random draws replayed.
    A ray only counts as an outlier when a discontinuity (argmax / mask threshold) flipped.  Bounds are what is observed on
    these fixtures (every ray samples identically, every ray inside 1e-4), with one ray of slack on the larger fixtures."""
    from oracle import vfnerf_oracle as O
    fx, d, g, model = case
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    with torch.no_grad():
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    n = d["uv"].shape[0]
    s_t = fx["n_samples"] + fx["n_importance"]
    assert out.points_coarse.shape == (n, s_t, 3) and out.coarse_normals.shape == (n, s_t, 3)
    assert out.coarse_rgb_values.shape == (n, 3) and out.coarse_depth_map.shape == (n, 1)
    assert out.z_vals.shape == (n, s_t) and out.ray_dirs.shape == (n * s_t, 3) and out.coarse_colors.shape == (n * s_t, 3)
    assert out.fine_normals is None and out.directional_derivtives is None
    r = _account(out, d, f"end-to-end {fx.get('n_rays')}x{s_t}")
    assert r["frac_same_z"] >= 0.99 or r["n_diff_z"] <= 1, "fp32 noise may move the argmax of one ray, not more"
    assert r["frac_within_tol"] >= 0.995 or (r["n_rays"] - r["n_diff_z"]) * (1 - r["frac_within_tol"]) <= 1.01
    assert r["rel_rgb"] < 2e-4 and r["rel_depth"] < 2e-4, "element-wise relative error of entries with |ref| > 1e-2 (observed: <= 1e-4)"
    good = (out.z_vals.cpu() == d["z_vals"]).all(dim=1)
    psnr = O.psnr(out.coarse_rgb_values.cpu()[good], d["rgb"][good])
    print(f"PSNR vs reference (matching rays): {psnr:.1f} dB")
    assert psnr > 80.0


@pytest.mark.parametrize("mode", ["default", "launch_by_launch", "colour2", "fp32"])
@pytest.mark.parametrize("name", ["trained_256", "trained_256_shipped", "trained_far"])
def test_render_on_trained_weights(name, mode):
    """Weights the REFERENCE'S OWN TRAINER arrived at (train_epoch on a teacher-rendered target at the shipped 8 x 256 / 4 x 256
    geometry, tests/golden/make_trained_golden.py), i.e. outside the synthetic init family of every other fixture, through the
    fused f16x3 kernels: the default path (three products everywhere, one C call), the same launch by launch, the opt-in
    two-product colour branch, and the exact-fp32 kernels.  ``trained_256``: 1 200 optimizer steps x 64 rays; ``trained_far``:
    6 000 steps x 256 rays, FAR from the init family — far enough that the two-product colour branch is outside the contract
    on it (the fixture's ``curve.colour_gap``).  Sample depths bit-exact; rgb / depth / normals / colours inside the 1e-4
    contract on EVERY ray; the range guard, in strict mode, has nothing to report about the default path.  The opt-in mode
    either stays inside the contract and on two products (the 1 200-step state) or is caught by the strict guard's measured
    self-check BEFORE the call returns, which then hands back the three-product colours (the far state)."""
    import warnings
    if not os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{name}.npz")):
        pytest.skip(f"tests/golden/{name}.npz has not been generated (make_trained_golden.py --far)")
    fx, d = load_fixture(name)
    g = to_dev(d)
    model = build_model(fx, d, device="cuda:0")
    assert model.colour_products == 3
    if mode == "launch_by_launch":
        model.one_call_render = False
    elif mode == "colour2":
        model.colour_products = 2
    elif mode == "fp32":
        model.precision = "fp32"
    model.f16x3_guard = "strict"
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    switched = [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)]
    r = _account(out, d, f"trained {name} {mode}")
    assert torch.equal(out.z_vals.cpu(), d["z_vals"]), "sample depths must be bit-exact on trained weights"
    assert r["frac_within_tol"] == 1.0
    assert r["max_rgb"] < TOL and r["max_depth"] < TOL and r["max_normals"] < TOL and r["max_colors"] < TOL
    if mode != "fp32":
        assert model.uses_f16x3() and model.f16x3_disabled is None, model.f16x3_disabled
        assert model.range_guard.check_now(dev()) is None
    if mode != "colour2":
        assert not switched, switched                      # a guard switch warns: it must not happen on the default path
        assert model.colour_products == 3 and model.range_guard.colour_products_reason is None
    else:
        print(f"trained {name}: opt-in two-product colours -> ran on {model.colour_products} products "
              f"({model.range_guard.colour_products_reason or 'the guard kept two'})")
        if name == "trained_far":                          # far from init: the self-check must have refused the two-product colours
            assert model.colour_products == 3 and model.range_guard.colour_products_reason is not None
            assert any("colour_products" in m for m in switched)
        if model.colour_products == 3:
            assert r["max_colors"] < TIGHT and r["max_rgb"] < TIGHT       # what came back is the three-product render
    if mode in ("default", "launch_by_launch", "fp32"):
        assert r["max_colors"] < TIGHT and r["max_rgb"] < TIGHT


def test_render_at_an_annealing_epoch_matches_the_reference():
    """Past ``anneal_start`` the reference rewrites ``config.cos_sim_weights`` on every render (vector_field_nerf.py:232-234) and
    then does not use them (get_density takes uniform weights).  Fixture captured from the reference at epoch 1000
    with the shipped "hard" annealing: the HIP render at that epoch — both precisions, one-call and launch-by-launch — samples
    the same depths and agrees within the contract; the facade's config holds the annealed window afterwards, like the
    reference's."""
    fx, d = load_fixture("anneal_epoch")
    g = {k: v.to(dev()) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    for precision in ("fp32", "f16x3"):
        for one_call in (True, False):
            model = build_model(fx, d, device=dev())
            model.precision, model.one_call_render = precision, one_call
            before = model.config.cos_sim_weights.clone()
            with torch.no_grad():
                out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=fx["epoch"], uniforms=uni)
            assert torch.equal(out.z_vals.cpu(), d["z_vals"]), (precision, one_call)
            assert rel_err(out.coarse_normals.reshape(-1, 3), d["normals"].reshape(-1, 3)) < 2e-5
            assert rel_err(out.coarse_rgb_values, d["rgb"]) < TOL and rel_err(out.coarse_depth_map.reshape(-1), d["depth"].reshape(-1)) < TOL
            after = model.config.cos_sim_weights
            assert not torch.equal(after.cpu(), before.cpu()) and abs(float(after.sum()) - 1.0) < 1e-5, "the annealed window replaces the uniform one"


def test_white_background_adds_missing_opacity():
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    g = to_dev(d)
    with torch.no_grad():
        a = model.render(g["pose"], g["uv"], g["intrinsics"], 0, False, uniforms={"u_add": g["u_add"]})
        b = model.render(g["pose"], g["uv"], g["intrinsics"], 0, True, uniforms={"u_add": g["u_add"]})
    assert float((b.coarse_rgb_values - a.coarse_rgb_values).min()) >= -1e-6


def test_philox_uniforms():
    from vf_nerf_amd import lib
    out = torch.empty(1 << 20, device=dev())
    lib.fill_uniform(out, seed=7, offset=0)
    assert float(out.min()) >= 0.0 and float(out.max()) < 1.0
    assert abs(float(out.mean()) - 0.5) < 2e-3 and abs(float(out.var()) - 1 / 12) < 2e-3
    again = torch.empty_like(out)
    lib.fill_uniform(again, seed=7, offset=0)
    assert torch.equal(out, again)
    lib.fill_uniform(again, seed=7, offset=1)
    assert torch.equal(out[4:], again[:-4])     # offset counts Philox 4-tuples


def test_get_density_api(case):
    fx, d, g, model = case
    n, s_t = d["z_vals"].shape
    rep = g["ray_dirs"].unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
    sigma = model.get_density(g["normals"], rep, True)
    assert rel_err(sigma, d["sigma"]) < TIGHT


def test_grid_query_matches_full_forward():
    """vf_nerf_amd.grid.get_set_predictions (vector-only kernel, pinned overlapped copies, chunk dealing) against
    decoder(x)[:, :3] as evaluation/utils/mc_utils.py:88-104 computes it; two 'ranks' together cover every row."""
    from vf_nerf_amd import grid
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    dec = model.fine_vector_field_network
    gen = torch.Generator().manual_seed(0)
    samples = torch.rand(10000, 3, generator=gen) * 2 - 1
    # expected values from the ORACLE (decoder(x)[:, :3] on the CPU, oracle.vf_mlp is pinned by the reference's goldens)
    cpu_sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
    want = O.vf_mlp(samples, cpu_sd, 6, (4,))[:, :3]
    with torch.no_grad():
        full = dec(samples.to(dev()))[:, :3].cpu()          # the HIP full-row forward agrees with it as well
    assert rel_err(full, want) < 2e-5
    for precision, tol in (("fp32", 2e-5), ("f16x3", 2e-5)):     # both within the contract of the CPU path (its own fp32 rounding ~5e-6)
        model.precision = precision
        got = grid.get_set_predictions(dec, samples, 3000, dev())
        assert got.shape == (10000, 3) and rel_err(got, want) < tol, precision
    parts = [grid.get_set_predictions(dec, samples, 3000, dev(), rank=r, world_size=2) for r in range(2)]
    assert rel_err(parts[0] + parts[1], want) < 2e-5
    # device-resident grid (no staging) and a 4-column sample tensor (xyz + value, as mc_utils builds it)
    four = torch.cat([samples, torch.zeros(10000, 1)], dim=1).to(dev())
    assert rel_err(grid.get_set_predictions(dec, four, 3000, dev()).cpu(), want) < 2e-5
    assert float(parts[0][3000:6000].abs().max()) == 0.0 and float(parts[1][:3000].abs().max()) == 0.0


def test_grid_query_of_the_reference_lattice_needs_no_upload():
    """evaluation/methods.py:190-208 builds the query grid on the host as a separable lattice; ``get_set_predictions`` regenerates it on the
    device from its three axis tables (``vfn_grid_lattice_points``) while the host verifies the rows: the points are the host tensor's bit
    for bit, so the predictions are the upload path's bit for bit — with a translation and a centroid in the grid, for one rank and for two.
    A grid that only LOOKS like the lattice (one coordinate moved) is noticed and evaluated from the caller's tensor."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vf_nerf_amd import grid, lib
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    dec = model.fine_vector_field_network
    res = 40
    samples = bench.reference_lattice(res, scale=0.9, translation=(0.05, -0.1, 0.02), centroid=(0.0, 0.1, 0.45))
    lat = grid.lattice_axes(samples)
    pts = lib.grid_lattice_points(tuple(a.to(dev()) for a in lat[1:]), res, 12345, 30001)
    assert torch.equal(pts.cpu(), samples[12345:12345 + 30001])               # the kernel: rows of the caller's grid, bit for bit
    keep = grid.LATTICE_FAST_PATH
    try:
        grid.LATTICE_FAST_PATH = False
        want = grid.get_set_predictions(dec, samples, 3000, dev())
        assert grid.last_path == "upload"
        grid.LATTICE_FAST_PATH = True
        got = grid.get_set_predictions(dec, samples, 3000, dev())
        assert grid.last_path == "lattice" and torch.equal(got, want)
        parts = [grid.get_set_predictions(dec, samples, 3000, dev(), rank=r, world_size=2) for r in range(2)]
        assert grid.last_path == "lattice" and torch.equal(parts[0] + parts[1], want)
        cpu_sd = {k: v.detach().cpu() for k, v in dec.state_dict().items()}
        assert rel_err(got[::17], O.vf_mlp(samples[::17], cpu_sd, 6, (4,))[:, :3]) < 2e-5
        almost = samples.clone()
        almost[res ** 3 // 2 + 11, 0] += 1e-3
        got2 = grid.get_set_predictions(dec, almost, 3000, dev())
        assert grid.last_path == "upload"
        grid.LATTICE_FAST_PATH = False
        assert torch.equal(got2, grid.get_set_predictions(dec, almost, 3000, dev())) and not torch.equal(got2, want)
    finally:
        grid.LATTICE_FAST_PATH = keep


def test_quaternion_pose_rays():
    """pose[N,7] = (qr,qi,qj,qk,tx,ty,tz): utils/rendering.py:27-33 + pinhole_model.quat_to_rot (CUDA-only in the
    reference, Q13), against the oracle's device-free restatement."""
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import lib
    gen = torch.Generator().manual_seed(2)
    n = 77
    q = torch.randn(n, 4, generator=gen)
    t = torch.randn(n, 3, generator=gen)
    pose7 = torch.cat([q, t], dim=1)
    uv = torch.rand(n, 2, generator=gen) * 64
    K = torch.eye(4).repeat(n, 1, 1)
    K[:, 0, 0] = 60.0; K[:, 1, 1] = -55.0; K[:, 0, 2] = 31.5; K[:, 1, 2] = 31.5; K[:, 0, 1] = 0.3   # negative fy: z = -1
    d_ref, rd_ref, c_ref = O.ray_directions(uv, pose7, K)
    t_vals = torch.linspace(0., 1., steps=8).to(dev())
    directions, ray_dirs, cam_loc, z, pts = lib.raygen_uniform(uv.to(dev()), pose7.to(dev()), K.to(dev()), t_vals, 8, 0.0, 1.0)
    assert torch.equal(cam_loc.cpu(), c_ref)
    assert rel_err(directions, d_ref) < 2e-6 and rel_err(ray_dirs, rd_ref) < 2e-6
    assert float(directions.cpu()[:, :].abs().max()) > 0 and float((ray_dirs.norm(dim=1) - 1).abs().max()) < 1e-5


def test_secondary_entry_points(case):
    """get_vector_field / get_colors / get_weights_and_color (vector_field_nerf.py:341-440) against the oracle."""
    from oracle import vfnerf_oracle as O
    fx, d, g, model = case
    n, s_t = d["z_vals"].shape
    cpu_vf = {k: v.cpu() for k, v in model.vector_field_network.state_dict().items()}
    cpu_rn = {k: v.cpu() for k, v in model.rendering_network.state_dict().items()}
    rep = d["ray_dirs"].unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
    with torch.no_grad():
        w, c = model.get_weights_and_color(g["points"], rep.to(dev()), g["z_vals"], epoch=0)
    # (weights: the density's scale-100 Laplace CDF amplifies the ~6e-6 by which two fp32 evaluations of the normals differ; on the
    # 64 + 64-sample fixture that is 8.1e-5 with every kernel, the exact-fp32 ones included — inside the 1e-4 contract, not TIGHT)
    assert rel_err(w, d["weights"]) < (1e-4 if fx["n_samples"] + fx["n_importance"] >= 128 else TIGHT) and rel_err(c, d["colors"]) < TIGHT
    if not fx["perturb"]:
        with torch.no_grad():
            vec = model.get_vector_field(g["pose"], g["uv"], g["intrinsics"])
            colors, flat, rep_c = model.get_colors(g["pose"], g["uv"], g["intrinsics"], epoch=0)
        assert rel_err(vec, d["normals_coarse"].reshape(-1, 3)) < TIGHT
        ref_vf = O.vf_mlp(flat.cpu(), cpu_vf)
        ref_c = O.render_mlp(flat.cpu(), ref_vf[:, :3], rep_c.cpu(), ref_vf[:, 3:], cpu_rn)
        assert rel_err(colors, ref_c) < TIGHT


def test_supervision_sphere_shell_sampler():
    """vfn_sample_sphere_shell / vf_nerf_amd.supervision against the oracle's restatement of SphereSampler +
    sample_border_points / sample_center_points on the same uniforms, and the device Philox stream's statistics."""
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import lib, supervision
    gen = torch.Generator().manual_seed(8)
    n = 5000
    u = torch.rand(n, 3, generator=gen)
    c = torch.tensor([0.1, -0.2, 0.6])
    for r_min, r_max, inward in ((0.75, 1.0, True), (0.0, 0.05, False), (0.3, 0.3, True)):
        want_p, want_g = O.sphere_shell_points(u, r_min, r_max, c, inward)
        got_p, got_g = lib.sample_sphere_shell(n, r_min, r_max, c.to(dev()), inward, u=u.to(dev()))
        assert float((got_p.cpu() - want_p).abs().max()) < 2e-6
        big = (want_p - c).norm(dim=1) > 1e-3            # the direction of a point ~1e-7 from the centroid is all rounding
        assert float((got_g.cpu() - want_g)[big].abs().max()) < 2e-4 if r_max < 0.1 else float((got_g.cpu() - want_g).abs().max()) < 2e-5
    supervision.manual_seed(11)
    bp, bg = supervision.sample_border_points(0.75, 1.0, 20000, c, dev())
    cp, cg = supervision.sample_center_points(c, 0.05, 20000, dev())
    rb, rc = (bp.cpu() - c).norm(dim=1), (cp.cpu() - c).norm(dim=1)
    assert float(rb.min()) >= 0.75 - 1e-6 and float(rb.max()) <= 1.0 + 1e-6 and float(rc.max()) <= 0.05 + 1e-6
    assert float(((bp.cpu() - c) / rb[:, None]).mean(0).abs().max()) < 0.02        # directions uniform on the sphere
    assert abs(float(((rc / 0.05) ** 3).mean()) - 0.5) < 0.02                        # cbrt(u) radius law
    assert float((bg.cpu() + (bp.cpu() - c) / rb[:, None]).abs().max()) < 1e-5      # border gt points at the centroid
    assert float((cg.cpu() - (cp.cpu() - c) / rc[:, None]).abs().max()) < 1e-3      # centre gt points away from it
    again = supervision.sample_border_points(0.75, 1.0, 20000, c, dev())[0]
    assert not torch.equal(again, bp), "the stream advances"


def test_numerical_directional_derivatives_on_hip():
    """numerical_jacobian=True end to end on the HIP path against the reference's golden output.  The quantity is a central
    difference with epsilon = 1e-5 of fp32 network outputs, so a forward difference of 1e-7 between two implementations
    shows up as 5e-3 in a Jacobian entry — the tolerance is that noise floor, not the 1e-4 of the smooth outputs."""
    fx, d = load_fixture("numjac_det")
    model = build_model(fx, d, device="cuda:0")
    g = {k: v.to(dev()) for k, v in d.items() if isinstance(v, torch.Tensor)}
    with torch.no_grad():
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={"u_add": g["u_add"]})
    assert torch.equal(out.z_vals.cpu(), d["z_vals"])
    assert rel_err(out.coarse_rgb_values, d["rgb"]) < 2e-5
    dd, ref = out.directional_derivtives.cpu(), d["directional_derivatives"]
    assert dd.shape == ref.shape
    err = (dd - ref).abs()
    # yardstick: the same quantity from the oracle evaluated in float64 (no cancellation noise).  The reference's own fp32
    # output deviates from it by the noise floor; the HIP path must sit on the same floor.
    from oracle import vfnerf_oracle as O
    from helpers import oracle_settings
    f64 = lambda sd: {k: v.double() for k, v in sd.items()}
    cpu_model = build_model(fx, d)
    o64 = O.render(d["uv"].double(), d["pose"].double(), d["intrinsics"].double(), f64(cpu_model.vector_field_network.state_dict()),
                   f64(cpu_model.rendering_network.state_dict()), oracle_settings(fx), u_add=d["u_add"].double())
    exact = o64["directional_derivatives"].float()
    e_ref, e_hip = (ref - exact).abs(), (dd - exact).abs()
    print(f"directional derivatives (max |value| {float(ref.abs().max()):.1f}): vs reference max {float(err.max()):.3e} median "
          f"{float(err.median()):.3e}; vs float64: reference median {float(e_ref.median()):.3e} max {float(e_ref.max()):.3e}, "
          f"HIP median {float(e_hip.median()):.3e} max {float(e_hip.max()):.3e}")
    assert float(e_hip.median()) <= 3.0 * float(e_ref.median()) + 1e-3 and float(e_hip.max()) <= 3.0 * float(e_ref.max()) + 1e-2
    assert float(err.max()) < 1e-2 * float(ref.abs().max()) + 0.5
    # with gradients: the fine-pass half of the derivatives carries autograd history through six more VF forwards
    for p in model.unique_parameters():
        p.grad = None
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={"u_add": g["u_add"]})
    assert out.directional_derivtives.requires_grad
    out.directional_derivtives.sum().backward()
    assert all(torch.isfinite(p.grad).all() for p in model.vector_field_network.parameters() if p.grad is not None)
    assert float(model.vector_field_network.layers[0][0].weight.grad.abs().max()) > 0


def test_grid_stages_on_hip():
    """vf_nerf_amd.grid.extract_divergence / smooth_vf / unify_direction / make_comb_format (one HIP kernel each) against
    outputs of the reference's own functions on the same fields.  The masks and choices are thresholds / argmins of fp32
    expressions: a cell may differ only if its deciding quantity is within rounding of the decision boundary."""
    import os
    import numpy as np
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import grid
    raw = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_stages.npz"))
    g = {k: torch.from_numpy(raw[k]) for k in raw.files}
    for n in (10, 13):
        pred = g[f"n{n}.pred"].to(dev())
        div = grid.extract_divergence(pred, n)
        assert div.shape == (n, n, n) and float((div.cpu() != g[f"n{n}.div"]).float().sum()) == 0
        for k, sigma, key in ((3, 1.0, "smooth3"), (9, 2.0, "smooth9")):
            sm = grid.smooth_vf(pred.reshape(n, n, n, 3), k=k, sigma=sigma)
            assert sm.shape == (n, n, n, 3) and float((sm.cpu() - g[f"n{n}.{key}"]).abs().max()) < 2e-6
        vt = torch.nn.functional.normalize(pred, dim=1).reshape(n, n, n, 3)
        choice = grid.unify_direction(div, vt.permute(3, 0, 1, 2), N=n)
        assert choice.dtype == torch.int64 and choice.shape == (n ** 3, 8)
        mism = (choice.cpu() != g[f"n{n}.choice"]).sum()
        assert int(mism) == 0, int(mism)
        norms = torch.norm(g[f"n{n}.pred"], dim=1).to(dev())          # as the reference computes them (CPU), then a pure gather
        comb, pair_norms = grid.make_comb_format(choice, norms, n)
        assert torch.equal(comb.cpu(), g[f"n{n}.comb"]) and torch.equal(pair_norms.cpu(), g[f"n{n}.pair_norms"])
    # a larger random grid against the oracle (edge planes, empty and full masks)
    n = 33
    gen = torch.Generator().manual_seed(5)
    pred = torch.randn(n ** 3, 3, generator=gen)
    div = grid.extract_divergence(pred.to(dev()), n).cpu()
    want = O.grid_divergence(pred, n)
    assert float((div != want).float().mean()) < 1e-3
    with pytest.raises(Exception):
        grid.extract_divergence(pred, n)      # host tensors are refused: no CPU fallback


@pytest.mark.parametrize("n", [64, 70, 131])
def test_grid_stages_tiled_kernels_at_larger_sizes(n):
    """The LDS-tiled / register-window kernels of round 3 on grids that span several footprints, march segments and partial
    tiles (64: whole tiles and vector accesses; 70: partial tiles, two march segments; 131: odd row lengths -> the scalar
    paths, a last wave that is not full), against the oracle's restatement of the reference's functions.  Thresholded /
    argmin'ed outputs may differ where the deciding fp32 quantity sits within rounding of the boundary (conv3d sums in another
    order): bounded as a fraction; everything that is a pure gather or comparison of integers is exact."""
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import grid, lib
    gen = torch.Generator().manual_seed(n)
    ax = torch.linspace(-1, 1, n)
    p = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
    # a field with surfaces: points towards the nearest of two spheres' shells, plus noise; a few exactly-zero vectors
    d1, d2 = p - torch.tensor([0.3, 0.0, 0.0]), p + torch.tensor([0.4, 0.2, 0.1])
    r1, r2 = d1.norm(dim=1, keepdim=True), d2.norm(dim=1, keepdim=True)
    f1, f2 = -d1 / r1.clamp_min(1e-6) * torch.sign(r1 - 0.45), -d2 / r2.clamp_min(1e-6) * torch.sign(r2 - 0.3)
    pred = torch.where((r1 - 0.45).abs() < (r2 - 0.3).abs(), f1, f2) * (0.2 + torch.rand(n ** 3, 1, generator=gen)) + \
        0.05 * torch.randn(n ** 3, 3, generator=gen)
    pred[::997] = 0.0
    pred = pred.contiguous()
    dp = pred.to(dev())
    div = grid.extract_divergence(dp, n)
    want_div = O.grid_divergence(pred, n)
    frac = float((div.cpu() != want_div).float().mean())
    print(f"n={n}: surface cells {float(want_div.mean()):.4f}, divergence mask mismatches {frac:.2e}")
    assert float(want_div.mean()) > 0.005 and frac < 2e-4
    assert float(div[-1].abs().max()) == 0 and float(div[:, -1].abs().max()) == 0 and float(div[:, :, -1].abs().max()) == 0
    for k, sigma in ((3, 1.0), (9, 2.0), (5, 1.5)):                      # 5: the generic per-voxel kernel
        sm = grid.smooth_vf(dp.reshape(n, n, n, 3), k=k, sigma=sigma)
        err = float((sm.cpu() - O.smooth_field(pred.reshape(n, n, n, 3), k, sigma)).abs().max())
        assert err < 2e-6, (k, err)
    assert torch.equal(dp.cpu(), pred), "smooth_vf must not write into its input"
    vt = torch.nn.functional.normalize(pred, dim=1)
    dvt = vt.to(dev()).reshape(n, n, n, 3)
    choice = grid.unify_direction(div, dvt.permute(3, 0, 1, 2), N=n)
    want_choice = O.grid_unify_direction(div.cpu(), vt.reshape(n, n, n, 3).permute(3, 0, 1, 2), n)
    cell_diff = float((choice.cpu() != want_choice).any(dim=1).float().sum() / max(1.0, float(div.sum())))
    print(f"n={n}: surface cells whose corner sides differ from the oracle's: {cell_diff:.2e}")
    assert choice.shape == (n ** 3, 8) and choice.dtype == torch.int64 and cell_diff < 2e-3
    assert int(choice.min()) == 0 and int(choice.max()) == 1 and int(choice[div.reshape(-1) != 1].abs().sum()) == 0
    sides, table = lib.grid_unify_direction_sides(div.reshape(-1).contiguous(), dvt.reshape(-1, 3).contiguous(), n)
    assert torch.equal(table, choice)
    bits = torch.stack([(sides.long() >> q) & 1 for q in range(8)], dim=1)
    assert torch.equal(bits, choice), "the side byte holds the same eight decisions"
    norms = torch.norm(pred, dim=1)
    dn = norms.to(dev())
    want_comb, want_pairs = O.grid_comb_format(choice.cpu(), norms, n)
    comb, pairs = grid.make_comb_format(choice, dn, n)                     # the tensor unify_direction returned: side bytes
    assert torch.equal(comb.cpu(), want_comb) and torch.equal(pairs.cpu(), want_pairs)
    comb2, pairs2 = grid.make_comb_format(choice.clone(), dn, n)           # any other int64 table: read as it is
    assert torch.equal(comb2, comb) and torch.equal(pairs2, pairs)
    odd = choice.clone()
    odd[::3] = odd[::3] * 5 + torch.arange(8, device=odd.device) % 3       # entries outside {0, 1}: "!=" on the integers themselves
    comb3, _ = grid.make_comb_format(odd, dn, n)
    assert torch.equal(comb3.cpu(), O.grid_comb_format(odd.cpu(), norms, n)[0])


def test_shared_pose_and_intrinsics():
    """One pose / intrinsics matrix per image instead of the per-ray replicas the reference's datasets upload."""
    fx, d = load_fixture("c1_det")
    model = build_model(fx, d, device="cuda:0")
    g = {k: v.to(dev()) for k, v in d.items() if isinstance(v, torch.Tensor)}
    with torch.no_grad():
        a = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={"u_add": g["u_add"]})
        b = model.render(g["pose"][0], g["uv"], g["intrinsics"][:1], epoch=0, uniforms={"u_add": g["u_add"]})
    assert torch.equal(a.coarse_rgb_values, b.coarse_rgb_values) and torch.equal(a.z_vals, b.z_vals)


def test_c_abi_from_plain_c(tmp_path):
    """The boundary is a C ABI: tests/c_abi/abi_smoke.c (C11, gcc, no Python, no torch) builds a geometry, packs random
    weights given in the reference's layout, launches rays / both VF kernels / one batch-statistics layer on its own stream
    and checks them against closed forms, against each other and against a host loop; also the error statuses."""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(repo, "vf_nerf_amd", "csrc")
    exe = str(tmp_path / "abi_smoke")
    build = subprocess.run(["gcc", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(repo, "tests", "c_abi", "abi_smoke.c"),
                            "-I" + os.path.join(repo, "include"), "-I/opt/rocm/include", "-L" + csrc, "-lvfn", "-L/opt/rocm/lib",
                            "-lamdhip64", "-lm", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib", "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(run.stdout.strip())
    assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
    assert "abi_smoke: ok" in run.stdout
