"""The CPU oracle (oracle/vfnerf_oracle.py) against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle; the GPU tests then compare HIP vs oracle."""
import pytest
import torch

from helpers import FIXTURE_NAMES, GRAD_KEYS, build_model, grad_rel_err, load_fixture, oracle_gradients, oracle_settings
from oracle import vfnerf_oracle as O


def _run(name):
    fx, d = load_fixture(name)
    model = build_model(fx, d)
    far = d.get("far_per_ray")
    out = O.render(d["uv"], d["pose"], d["intrinsics"], model.vector_field_network.state_dict(),
                   model.rendering_network.state_dict(), oracle_settings(fx), u_coarse=d.get("u_coarse"),
                   u_fine=d.get("u_fine"), u_add=d["u_add"], far=far)
    return fx, d, out


@pytest.mark.parametrize("name", FIXTURE_NAMES)
def test_render_matches_reference_bitwise_stages(name):
    """Sampling stages and indices are bit-exact; everything downstream of the MLPs within 1e-6."""
    fx, d, out = _run(name)
    for k in ("directions", "ray_dirs", "cam_loc", "z_coarse"):
        assert torch.equal(out[k], d[k]), k
    assert torch.equal(out["max_indices"], d["max_indices"])
    assert torch.equal(out["z_vals"], d["z_vals"])
    assert torch.equal(out["points"], d["points"])
    tol = 1e-6
    for k in ("normals_coarse", "window_cos_coarse", "sigma_coarse", "weights_coarse", "normals", "window_cos",
              "sigma", "weights", "colors", "rgb", "depth"):
        scale = max(1.0, float(d[k].abs().max()))
        assert float((out[k] - d[k]).abs().max()) <= tol * scale, (k, float((out[k] - d[k]).abs().max()))
    assert float((out["vf_out"][::8, 3:] - d["feats_sub"]).abs().max()) <= tol


def test_fixture_covers_edge_cases():
    """At least one fixture has rays whose proposal weights are all zero (argmax 0 -> uniform extras, Q9),
    an odd S_t, a per-ray far and a non-default window."""
    _, d = load_fixture("odd_orbit")
    assert int((d["max_indices"] == 0).sum()) > 0 and int((d["max_indices"] > 0).sum()) > 0
    assert d["z_vals"].shape[1] % 2 == 1 and "far_per_ray" in d


def test_window_cosine_short_rays():
    """Rays shorter than two window halves leave the adjacent cosine untouched (functions.py:59-70)."""
    torch.manual_seed(0)
    n = torch.randn(3, 9, 3)
    w = torch.ones(11) / 11
    c = O.window_cosine(n, w)
    assert torch.allclose(c, torch.nn.functional.cosine_similarity(n[:, :-1], n[:, 1:], dim=2))


def test_density_is_zero_above_half_cosine():
    """sigma > 0 only where the (negated) cosine is below the dropped cutoff -0.5, i.e. cos < 0.5 (Q5)."""
    p = O.DensityParams()
    x = torch.linspace(-1, 1, 201).reshape(-1, 1)
    s = O.laplace_density(-x, p)
    assert float(s[x > 0.5].abs().max()) == 0.0 and float(s[x < 0.49].min()) > 0.0


def test_psnr_definition():
    a = torch.zeros(10, 3)
    b = torch.full((10, 3), 0.1)
    assert abs(O.psnr(a, b) - 20.0) < 1e-4


@pytest.mark.parametrize("name", FIXTURE_NAMES)
def test_oracle_gradients_match_reference(name):
    """Autograd of the oracle against gradients captured from the reference's own backward pass."""
    from helpers import GRAD_KEYS, grad_rel_err, oracle_gradients
    fx, d = load_fixture(name)
    model = build_model(fx, d)
    loss, grads = oracle_gradients(fx, d, model)
    assert abs(loss - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    for net, key in GRAD_KEYS:
        err = grad_rel_err(grads[f"{net}.{key}"], d[f"grad.{net}.{key}"])
        assert err < 1e-4, (net, key, err)
    for k in ("beta", "mean", "scale"):
        assert grad_rel_err(grads[f"density.{k}"].reshape(1), d[f"grad.density.{k}"]) < 1e-4, k


def test_numerical_directional_derivatives_golden():
    """numerical_jacobian=True (vector_field_nerf.py:258-262,299-301,476-526): the oracle's restatement against the
    reference's own output, including the transposed Jacobian of the fine pass."""
    from helpers import build_model, load_fixture, oracle_settings
    from oracle import vfnerf_oracle as O
    fx, d = load_fixture("numjac_det")
    model = build_model(fx, d)
    out = O.render(d["uv"], d["pose"], d["intrinsics"], model.vector_field_network.state_dict(),
                   model.rendering_network.state_dict(), oracle_settings(fx), u_add=d["u_add"])
    n, s_c, n_f = fx["n_rays"], fx["n_samples"], fx["n_importance"]
    assert out["directional_derivatives"].shape == (2 * n * (s_c + s_c + n_f),)
    assert float((out["directional_derivatives"] - d["directional_derivatives"]).abs().max()) <= 1e-4 * float(d["directional_derivatives"].abs().max())
    assert float((out["rgb"] - d["rgb"]).abs().max()) < 1e-6


def _grid_golden():
    import os
    import numpy as np
    raw = np.load(os.path.join(os.path.dirname(__file__), "golden", "grid_stages.npz"))
    return {k: torch.from_numpy(raw[k]) for k in raw.files}


def test_grid_stages_oracle_vs_reference_golden():
    """The oracle's restatement of extract_divergence / smooth_vf / unify_direction / make_comb_format
    (evaluation/utils/mc_utils.py, guassian_smoothing.py) against outputs of the reference's own functions."""
    from oracle import vfnerf_oracle as O
    g = _grid_golden()
    for n in (10, 13):
        pred = g[f"n{n}.pred"]
        div = O.grid_divergence(pred, n)
        assert torch.equal(div, g[f"n{n}.div"])
        assert float((O.smooth_field(pred.reshape(n, n, n, 3), 3, 1.0) - g[f"n{n}.smooth3"]).abs().max()) < 1e-6
        assert float((O.smooth_field(pred.reshape(n, n, n, 3), 9, 2.0) - g[f"n{n}.smooth9"]).abs().max()) < 1e-6
        vt = torch.nn.functional.normalize(pred, dim=1).reshape(n, n, n, 3).permute(3, 0, 1, 2)
        choice = O.grid_unify_direction(div, vt, n)
        assert torch.equal(choice, g[f"n{n}.choice"])
        comb, pair_norms = O.grid_comb_format(choice, torch.norm(pred, dim=1), n)
        assert torch.equal(comb, g[f"n{n}.comb"]) and torch.equal(pair_norms, g[f"n{n}.pair_norms"])


def test_train_mode_matches_reference():
    """Networks in train mode (model.train(), vector_field_nerf.py:139-150): batch-statistics BatchNorm in both nets, the
    VF forward's three autograd.grad rows, analytic directional derivatives (coarse values twice, Q10), gradients through the
    batch statistics, running statistics after two VF batches / one rendering-net batch."""
    fx, d = load_fixture("train_mode")
    model = build_model(fx, d)
    loss, g = oracle_gradients(fx, d, model)
    out = g["_out"]
    assert torch.equal(out["z_vals"], d["z_vals"]) and torch.equal(out["points"], d["points"])
    for k in ("normals", "colors", "rgb", "depth"):
        assert float((out[k].reshape(d[k].shape) - d[k]).abs().max()) <= 2e-6, k
    assert float((out["vf_out"][:, :259] - d["vf_out"][:, :259]).abs().max()) <= 2e-6
    jac, want = out["vf_out"][:, 259:].detach(), d["vf_out"][:, 259:]
    assert jac.shape == want.shape == (d["z_vals"].numel(), 9)
    assert float((jac - want).abs().max()) <= 2e-5 * float(want.abs().max())
    dd, dd_want = out["directional_derivatives"], d["directional_derivatives"]
    assert dd.shape == dd_want.shape and float((dd - dd_want).abs().max()) <= 2e-5 * float(dd_want.abs().max())
    assert abs(loss - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    for net, key in GRAD_KEYS:
        want = d[f"grad.{net}.{key}"]
        if key.endswith(".0.bias"):        # Linear bias in front of batch statistics: the gradient is exactly 0 up to rounding
            assert float(g[f"{net}.{key}"].abs().max()) <= 1e-4 and float(want.abs().max()) <= 1e-4
            continue
        assert grad_rel_err(g[f"{net}.{key}"], want) <= 2e-4, (net, key, grad_rel_err(g[f"{net}.{key}"], want))
    for name in ("beta", "mean", "scale"):
        assert grad_rel_err(g[f"density.{name}"].reshape(1), d[f"grad.density.{name}"]) <= 2e-4, name
    for net, i in (("vf", 0), ("vf", 3), ("vf", 7), ("rn", 0), ("rn", 3)):
        sd = g["_state"][net]
        for stat in ("running_mean", "running_var"):
            got, want = sd[f"layers.{i}.1.{stat}"], d[f"bn.{net}.{i}.{stat}"]
            assert float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), (net, i, stat)
        assert int(sd[f"layers.{i}.1.num_batches_tracked"]) == int(d[f"bn.{net}.{i}.num_batches_tracked"])
