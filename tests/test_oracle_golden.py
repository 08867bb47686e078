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


@pytest.mark.parametrize("name", FIXTURE_NAMES + ("anneal_epoch",))      # (anneal_epoch: captured at epoch 1000, past anneal_start — Q6)
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


# ------------------------------------------------------------------------------------------------
# trainer-side pieces against the reference's own VFLoss / SphereSampler / helpers / train_epoch
# ------------------------------------------------------------------------------------------------
def test_vf_loss_matches_reference_vfloss():
    """models/losses/vf_loss.py:34-87 on every branch: early epoch, past norm_smaller_than_one_start with directional
    derivatives, derivatives before their start epoch, a batch without depth and without supervised normals."""
    from helpers import load_trainer_fixture
    _, d = load_trainer_fixture()
    w = O.LossWeights(directional_derivatives=0.3)
    base = {k[len("loss.in."):]: v for k, v in d.items() if k.startswith("loss.in.")}
    for name in ("early", "late", "dd_before_start", "no_depth_no_sup"):
        epoch, dd, depth, sup = [int(v) for v in d[f"loss.{name}.case"]]
        total, terms = O.vf_loss_terms(base["rgb"], base["depth"], base["normals"], base["sup"] if sup else torch.empty(0, 3),
                                       base["rgb_gt"], base["depth_gt"] if depth else torch.empty(0),
                                       base["sup_gt"] if sup else torch.empty(0), w, epoch, base["dd"] if dd else None)
        assert abs(float(total) - float(d[f"loss.{name}.total"])) <= 1e-6 * max(1.0, abs(float(total))), name
        got = torch.stack([t.double() for t in terms])
        assert float((got - d[f"loss.{name}.terms"]).abs().max()) <= 1e-6, (name, got, d[f"loss.{name}.terms"])
    assert float(d["loss.late.terms"][4]) > 0 and float(d["loss.late.terms"][5]) > 0 and float(d["loss.dd_before_start.terms"][5]) == 0


def test_supervision_samplers_match_reference():
    """SphereSampler.sample + sample_border_points / sample_center_points (sampler.py:160-193, functions.py:100-135) replayed
    on the sampler's own numpy draws: points and ground truth bit-identical; get_border_indices_and_gt / get_center_indices_and_gt
    (functions.py:75-98,137-157) on fixed inputs."""
    from helpers import load_trainer_fixture
    fx, d = load_trainer_fixture()
    c = torch.tensor(fx["centroid"])
    for t in range(fx["steps"]):
        r_min, r_max, num = d[f"s{t}.border_args"].tolist()
        assert int(num) == (fx["n_rays"] * (fx["n_samples"] + fx["n_importance"])) // 10
        assert abs(r_min - (fx["far"] - 5 * fx["border_radius"])) < 1e-12 and r_max == fx["far"]
        p, g = O.sphere_shell_points_from_draws(d[f"s{t}.border_draws"], r_min, r_max, c, inward=True)
        assert torch.equal(p, d[f"s{t}.border_points"]) and torch.equal(g, d[f"s{t}.border_gt"])
        radius, num_c = d[f"s{t}.center_args"].tolist()
        p, g = O.sphere_shell_points_from_draws(d[f"s{t}.center_draws"], 0.0, radius, c, inward=False)
        assert torch.equal(p, d[f"s{t}.center_points"]) and torch.equal(g, d[f"s{t}.center_gt"])
        # the unit-uniform form (what the device sampler is replayed against) agrees to float32 rounding
        p2, g2 = O.sphere_shell_points(d[f"s{t}.center_u"], 0.0, radius, c, inward=False)
        assert float((p2 - p).abs().max()) <= 1e-7 and float((g2 - g).abs().max()) <= 1e-5
        # ray samples inside the centre ball: selection on the reference's own render outputs
        n_, g_ = O.center_indices_and_gt(d[f"s{t}.out.points"], d[f"s{t}.out.normals"], c, radius)
        assert torch.equal(n_, d[f"s{t}.ray_center_normals"]) and torch.equal(g_, d[f"s{t}.ray_center_gt"])
    far, radius = d["border_idx.args"].tolist()
    a, b = O.border_indices_and_gt(d["border_idx.points"], d["border_idx.normals"], far, radius, d["border_idx.centroid"])
    assert torch.equal(a, d["border_idx.out_normals"]) and torch.equal(b, d["border_idx.out_gt"]) and 0 < a.shape[0] < 60


def test_trainer_epoch_matches_reference_train_epoch():
    """Three optimizer steps of the reference's own VectorFieldNerfRunner.train_epoch (train/vector_field_nerf_train.py:161-260;
    captured by tests/golden/make_train_golden.py) against the oracle's restatement from the same weights, batches and draws:
    sampled depths bit-identical at every step, the six loss terms, the total, the value clip_grad_norm_ returned, the learning
    rate, and the watched parameters after every optimizer.step (VF parameters: two Adam updates per step, Q4)."""
    from helpers import TRAINER_WATCH, load_trainer_fixture, lr_gamma, trainer_batches, trainer_loss_weights, watched_slice
    fx, d = load_trainer_fixture()
    model = build_model(fx, d)
    vf_sd = {k: v.detach().clone() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().clone() for k, v in model.rendering_network.state_dict().items()}
    names = {"vf": [k for k, _ in model.vector_field_network.named_parameters()],
             "rn": [k for k, _ in model.rendering_network.named_parameters()]}
    for sd, keys in ((vf_sd, names["vf"]), (rn_sd, names["rn"])):
        for k in keys:
            sd[k].requires_grad_(True)
    density = {k: torch.tensor(v, requires_grad=True) for k, v in (("beta", 0.5), ("scale", 100.0), ("mean", 0.7))}
    w0 = {f"{net}.{key}": watched_slice(dict(vf=vf_sd, rn=rn_sd)[net][key].detach(), how).clone() for net, key, how in TRAINER_WATCH}
    seen = []

    def on_step(t, rec):
        sds = {"vf": vf_sd, "rn": rn_sd}
        seen.append({f"{net}.{key}": watched_slice(sds[net][key].detach(), how).clone() for net, key, how in TRAINER_WATCH} |
                    {f"density.{k}": v.detach().clone().reshape(1) for k, v in density.items()})

    recs, opt = O.trainer_epoch(vf_sd, rn_sd, density, names, trainer_batches(fx, d), oracle_settings(fx), trainer_loss_weights(fx),
                                fx["epoch"], torch.tensor(fx["centroid"]), fx["border_radius"], fx["far"], fx["lr"], lr_gamma(fx),
                                fx["clip_norm"], on_step=on_step)
    lr = fx["lr"]
    for t, rec in enumerate(recs):
        assert torch.equal(rec["out"]["z_vals"], d[f"s{t}.out.z_vals"]), f"step {t}: sampled depths differ"
        for k in ("rgb", "depth", "normals"):
            want = d[f"s{t}.out.{k}"]
            assert float((rec["out"][k].reshape(want.shape) - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), (t, k)
        got = torch.stack([x.double() for x in rec["terms"]])
        assert float((got - d[f"s{t}.loss_terms"]).abs().max()) <= 2e-6, (t, got, d[f"s{t}.loss_terms"])
        assert abs(float(rec["loss"]) - float(d[f"s{t}.loss"])) <= 2e-6 * max(1.0, float(d[f"s{t}.loss"]))
        assert abs(float(rec["clip_total_norm"]) - float(d[f"s{t}.clip_total_norm"])) <= 1e-4 * float(d[f"s{t}.clip_total_norm"]), t
        assert abs(rec["lr"] - float(d[f"s{t}.lr"])) <= 1e-12
        # parameters after the step: a step moves a weight by <= lr per Adam update; agreement is asked to 2 % of one update
        for k, v in seen[t].items():
            want = d[f"s{t}.after.{k}"]
            assert float((v - want).abs().max()) <= 0.02 * lr + 1e-6 * float(want.abs().max()), (t, k, float((v - want).abs().max()))
    # Q4: the aliased VF parameters took two updates per step, the rendering net's one
    moved_vf = float((seen[0]["vf.layers.8.weight"] - w0["vf.layers.8.weight"]).abs().max())
    moved_rn = float((seen[0]["rn.layers.4.weight"] - w0["rn.layers.4.weight"]).abs().max())
    assert abs(moved_vf - 2 * lr) < 0.05 * lr and abs(moved_rn - lr) < 0.05 * lr, (moved_vf, moved_rn)
    assert abs(opt.param_groups[0]["lr"] - float(d["final_lr"])) <= 1e-15


def test_attached_normals_gradients_match_reference():
    """detach_normals=False (rendering_network.py:76-77, not the shipped value): the colours' gradient also reaches the normals
    and through them the VF net — forward unchanged, VF gradients different from the detached case."""
    fx, d = load_fixture("attached_normals")
    assert fx["detach_normals"] is False
    model = build_model(fx, d)
    loss, grads = oracle_gradients(fx, d, model)
    assert torch.equal(grads["_out"]["z_vals"], d["z_vals"])
    assert abs(loss - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    for net, key in GRAD_KEYS:
        err = grad_rel_err(grads[f"{net}.{key}"], d[f"grad.{net}.{key}"])
        assert err < 1e-4, (net, key, err)
    _, detached = oracle_gradients(dict(fx, detach_normals=True), d, model)
    k = "vf.layers.7.1.weight"
    assert grad_rel_err(detached[k], d[f"grad.{k}"]) > 1e-3, "the fixture must tell the two settings apart"


@pytest.mark.parametrize("task,recorded", [("trained_256_run", "trained_256"), ("trained_far_run", "trained_far")])
def test_recorded_run_fixture_is_the_recorded_run(task, recorded):
    """tests/golden/trained_256_run.npz (make_run_golden.py): the task of the run recorded in trained_256.npz.  Row 0 of its reference curves
    IS the recorded curve (the reference trainer re-run on the regenerated batches reproduced it bit for bit), the initial weights rebuild
    from the recipe + the stored vector head to the reference model's checksum, and the targets are what the teacher rendered (hit fraction
    of the recorded run)."""
    import ast
    import os
    import sys

    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    run = np.load(os.path.join(here, "golden", f"{task}.npz"))
    rec = np.load(os.path.join(here, "golden", f"{recorded}.npz"))
    assert float(run["runs.reproduces_recorded"][0]) == 0.0 and np.array_equal(run["runs.loss"][0], rec["curve.loss"])
    assert np.array_equal(run["runs.terms"][0], rec["curve.terms"]) and np.array_equal(run["runs.clip"][0], rec["curve.clip"])
    assert np.array_equal(run["runs.psnr_before_after"][0], rec["curve.psnr_before_after"])
    recipe = ast.literal_eval(str(run["train_recipe"]))
    assert recipe == ast.literal_eval(str(rec["train_recipe"]))
    r, steps = run["runs.loss"].shape
    assert r >= 4 and steps == recipe["epochs"] * recipe["steps_per_epoch"] and len({tuple(s) for s in run["runs.seeds"].tolist()}) == r
    assert run["batch.uv"].shape == (recipe["steps_per_epoch"], recipe["n_rays"], 2) and run["batch.rgb"].shape == (recipe["steps_per_epoch"], recipe["n_rays"], 3)
    hit = float((run["batch.depth"] > 0.02).mean())
    assert abs(hit - float(rec["curve.target_hit_fraction"][0])) < 1e-6
    sys.path.insert(0, os.path.join(os.path.dirname(here), "tools"))
    import replay_reference_run as rr
    import torch
    model = rr.build_student(run, recipe, torch.device("cpu"))      # (asserts the checksum of the reference model the run started from)
    assert model.ray_sampler.N_samples == recipe["n_samples"]
