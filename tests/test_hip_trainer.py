"""The trainer-facing surface on the GPU, against outputs of the REFERENCE's own trainer / loss / samplers / checkpoint
(tests/golden/trainer_steps.npz, ref_checkpoint_latest.pth — made by tests/golden/make_train_golden.py, which runs
train/vector_field_nerf_train.py:161-292 itself) and against the oracle those fixtures pin."""
import os
import tempfile
from types import SimpleNamespace

import pytest
import torch

from helpers import (GRAD_KEYS, REF_CHECKPOINT, TRAINER_WATCH, build_model, grad_rel_err, load_fixture, load_trainer_fixture,
                     loss_coefficients, lr_gamma, narrow_checkpoint_model, oracle_gradients, oracle_settings, rel_err,
                     trainer_batches, trainer_loss_weights, watched_slice)
from oracle import vfnerf_oracle as O
from vf_nerf_amd import lib, loss as vloss, optim, supervision

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_run(fx, d):
    """The oracle's restatement of the three trainer steps on this host (bit-identical to the reference's run in the build
    container; re-run here for the per-step gradients, which the fixture does not store)."""
    model = build_model(fx, d)
    vf_sd = {k: v.detach().clone() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().clone() for k, v in model.rendering_network.state_dict().items()}
    names = {"vf": [k for k, _ in model.vector_field_network.named_parameters()],
             "rn": [k for k, _ in model.rendering_network.named_parameters()]}
    for sd, keys in ((vf_sd, names["vf"]), (rn_sd, names["rn"])):
        for k in keys:
            sd[k].requires_grad_(True)
    density = {k: torch.tensor(v, requires_grad=True) for k, v in (("beta", 0.5), ("scale", 100.0), ("mean", 0.7))}
    grads = []

    def on_step(t, rec):
        sds = {"vf": vf_sd, "rn": rn_sd}
        grads.append({f"{net}.{key}": watched_slice(sds[net][key].grad.detach(), how).clone() for net, key, how in TRAINER_WATCH})

    recs, _ = O.trainer_epoch(vf_sd, rn_sd, density, names, trainer_batches(fx, d), oracle_settings(fx), trainer_loss_weights(fx),
                              fx["epoch"], torch.tensor(fx["centroid"]), fx["border_radius"], fx["far"], fx["lr"], lr_gamma(fx),
                              fx["clip_norm"], on_step=on_step)
    return recs, grads


@pytest.mark.parametrize("precision", ["fp32", "f16x3+fp32grads", "f16x3"])
def test_trainer_steps_replay_on_hip(precision):
    """Three steps of the reference's train_epoch (train/vector_field_nerf_train.py:169-260) replayed on the HIP path through
    the interface the trainer uses — render(pose, pixels, intrinsics, epoch, white), functions.sample_border_points /
    get_center_indices_and_gt / sample_center_points, vector_field_network(points)[:, :3], VFLoss, optimizer.zero_grad,
    backward, clip_grad_norm_(model.parameters(), clip), optimizer.step, scheduler.step — with the reference's own draws.
    Steps 0 and 1: the six loss terms and the total within 1e-4, the clip norm within 1e-3, sampled depths bit-identical, the
    parameters after the step within 2 % of one Adam update where the gradient is significant — with the exact-fp32 kernels and
    with the f16x3 kernels keeping fp32 gradients between the chain and the weight-gradient kernels.  The DEFAULT storage (scaled
    f16 gradients, every parameter gradient within 1e-3 of the fp32 kernels') is held to the same bounds at step 0 and to 1e-3 on
    the loss terms at step 1 (3.4e-4 measured): Adam's first update is +-lr by the SIGN of each gradient, so the 5e-4 of rounding
    in the gradients decides more signs than the 7e-5 of the fp32 storage do.  Step 2 onward is reported
    and bounded loosely: Adam's early updates are +-lr per weight, i.e. sign decisions on gradients that are partly
    rounding noise, so two fp32-accurate implementations drift apart by design."""
    fx, d = load_trainer_fixture()
    dev = torch.device(DEV)
    model = build_model(fx, d, device=DEV)
    model.precision = precision.split("+")[0]
    # (the launch-by-launch autograd path of backward.py, dense colours: the same calls through the step session — the default since round
    #  5 — are test_reference_call_sequence_replays_the_reference_trainer_steps)
    model.step_sessions = False
    if "fp32grads" in precision:
        model.gradient_storage = "fp32"
    default_storage = precision == "f16x3"
    model.scheduler = torch.optim.lr_scheduler.ExponentialLR(model.optimizer, lr_gamma(fx))     # lr_decay_steps of the fixture
    crit = vloss.VFLoss(SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                        SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1,
                                        directional_derivatives=0.0))
    centroid = torch.tensor(fx["centroid"], device=dev)
    far, radius = fx["far"], fx["border_radius"]
    oracle_recs, oracle_grads = _oracle_run(fx, d)
    lr = fx["lr"]
    report = []
    for t, b in enumerate(trainer_batches(fx, d, device=DEV)):
        supervision.replay_uniforms(b["border_u"], b["center_u"])
        # ---- the trainer's step, line for line (:172-260, the non-"center" init branch) ----
        outputs = model.render(b["pose"], b["uv"], b["intrinsics"], fx["epoch"], False,
                               uniforms={k: b[k] for k in ("u_coarse", "u_fine", "u_add")})
        n_sup = (outputs.points_coarse.shape[0] * outputs.points_coarse.shape[1]) // 10
        supervised_normals = torch.empty(0, 3, device=dev)
        gt_normals = torch.empty(0, device=dev)
        border_points, border_gt = supervision.sample_border_points(far - 5 * radius, far, n_sup, centroid, dev)
        supervised_normals = torch.cat([supervised_normals, model.vector_field_network(border_points)[:, :3]], dim=0)
        gt_normals = torch.cat([gt_normals.reshape(-1, 3), border_gt], dim=0)
        rc_normals, rc_gt = supervision.get_center_indices_and_gt(outputs.points_coarse, outputs.coarse_normals, centroid, radius)
        center_points, center_gt = supervision.sample_center_points(centroid, radius, n_sup, dev)
        supervised_normals = torch.cat([supervised_normals, rc_normals, model.vector_field_network(center_points)[:, :3]], dim=0)
        gt_normals = torch.cat([gt_normals, rc_gt, center_gt], dim=0)
        predictions = {"rgb": outputs.coarse_rgb_values, "depth": outputs.coarse_depth_map,
                       "normals": outputs.coarse_normals.reshape(-1, 3), "supervised_normals": supervised_normals,
                       "directional_derivatives": outputs.directional_derivtives}
        ground_truth = {"rgb": b["rgb_gt"].reshape(-1, 3), "depth": b["depth_gt"], "supervised_normals": gt_normals}
        loss, losses_dict = crit(predictions, ground_truth, fx["epoch"])
        model.optimizer.zero_grad()
        loss.backward()
        total_norm = optim.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
        lr_now = model.optimizer.param_groups[0]["lr"]
        model.optimizer.step()
        model.scheduler.step()
        # ---- against the reference's record of the same step ----
        same_z = bool(torch.equal(outputs.z_vals.cpu(), d[f"s{t}.out.z_vals"]))
        terms = torch.tensor(list(losses_dict.values()), dtype=torch.float64)
        e_terms = float((terms - d[f"s{t}.loss_terms"]).abs().max())
        e_loss = abs(float(loss) - float(d[f"s{t}.loss"])) / max(1.0, float(d[f"s{t}.loss"]))
        e_clip = abs(float(total_norm) - float(d[f"s{t}.clip_total_norm"])) / float(d[f"s{t}.clip_total_norm"])
        e_pts = max(float((border_points.cpu() - d[f"s{t}.border_points"]).abs().max()),
                    float((center_points.cpu() - d[f"s{t}.center_points"]).abs().max()))
        nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
        worst_w = 0.0
        for net, key, how in TRAINER_WATCH:
            got = watched_slice(dict(nets[net].named_parameters())[key].detach().cpu(), how)
            want = d[f"s{t}.after.{net}.{key}"]
            g = oracle_grads[t][f"{net}.{key}"]
            sig = g.abs() > 1e-2 * g.abs().max()
            worst_w = max(worst_w, float((got - want).abs()[sig].max()))
        report.append((t, same_z, e_terms, e_loss, e_clip, worst_w / lr, e_pts))
        print(f"[{precision}] step {t}: depths identical {same_z}; loss terms |d| {e_terms:.2e}; total rel {e_loss:.2e}; clip norm rel "
              f"{e_clip:.2e}; watched weights (significant gradients) off by {worst_w / lr:.3f} lr; supervision points |d| {e_pts:.1e}")
        assert abs(lr_now - float(d[f"s{t}.lr"])) < 1e-12
        assert int(rc_normals.shape[0]) == int(d[f"s{t}.ray_center_normals"].shape[0]) or not same_z
        if t <= 1:
            assert same_z and e_pts < 5e-6
            assert e_terms < (1e-3 if (default_storage and t == 1) else 1e-4) and e_loss < 1e-4, report[-1]
        if t == 0:      # identical weights: the gradient norm and the update itself agree
            assert e_clip < 1e-3 and worst_w < 0.02 * lr + 1e-7, report[-1]
        else:
            # From the second step on the two runs no longer hold the same weights: the first Adam update is +-lr per weight by
            # the SIGN of a gradient that is partly rounding noise, and on this 672-point batch one ReLU unit landing on the other
            # side of zero moves a layer's gradient by a percent.  Observed: clip norm 0.15 % (fp32 kernels) / 2 % (f16x3) off at
            # step 1 with the watched weights within one update, 2 % / 26 % at step 2.  Bounded, not pinned (DESIGN.md section 5,
            # loss-curve agreement).
            # (default 16-bit gradient storage: more of the first update's signs are decided by rounding, see the docstring — by
            # step 2 that run has its own weights, so only its loss is bounded there)
            if default_storage:
                assert (e_clip < 0.3 and worst_w < 4.0 * lr) if t == 1 else e_loss < 0.1, report[-1]
            else:
                assert e_clip < (0.05 if t == 1 else 0.5) and worst_w < (2.0 if t == 1 else 6.0) * lr and e_loss < 5e-2, report[-1]
    st = model.optimizer.state[model.vector_field_network.layers[8].weight]
    assert float(st["step"]) == 2 * fx["steps"], "the aliased VF parameters take two Adam updates per step (Q4)"
    assert float(model.optimizer.state[model.rendering_network.layers[4].weight]["step"]) == fx["steps"]
    assert abs(model.optimizer.param_groups[0]["lr"] - float(d["final_lr"])) < 1e-15


@pytest.mark.parametrize("colours", ["sparse", "dense"])
def test_one_call_step_replays_the_reference_trainer_steps(colours):
    """The steps of the reference's OWN train_epoch recorded in tests/golden/trainer_steps.npz (draws, six loss terms, clip norm,
    post-step weights) replayed through trainer.TrainStep's ONE C call (vfn_train_step) — by default with the sparse colour branch —
    instead of through the calls the trainer makes one by one (test_trainer_steps_replay_on_hip).  Step 0 starts from the
    reference's own state: sampled depths bit-identical, loss terms within 1e-4, the clip norm within 1e-3 and the watched weights
    after the update within 2 % of one Adam update where the gradient is significant — the bounds the launch-by-launch replay is held
    to with the default 16-bit storages.  Step 1 within 1e-3 on the loss terms; later steps are other trajectories (Adam's sign
    decisions), bounded loosely."""
    from vf_nerf_amd import trainer
    fx, d = load_trainer_fixture()
    dev = torch.device(DEV)
    model = build_model(fx, d, device=DEV)
    model.sparse_colour_training = colours == "sparse"
    model.scheduler = torch.optim.lr_scheduler.ExponentialLR(model.optimizer, lr_gamma(fx))
    crit = vloss.VFLoss(SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                        SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0))
    step = trainer.TrainStep(model, fx["centroid"], border_radius=fx["border_radius"], far=fx["far"], criterion=crit)
    _, oracle_grads = _oracle_run(fx, d)
    lr = fx["lr"]
    for t, b in enumerate(trainer_batches(fx, d, device=DEV)):
        supervision.replay_uniforms(b["border_u"], b["center_u"])
        lr_now = model.optimizer.param_groups[0]["lr"]
        loss, terms = step(b["pose"], b["uv"], b["intrinsics"], b["rgb_gt"], b["depth_gt"], epoch=fx["epoch"],
                           uniforms={k: b[k] for k in ("u_coarse", "u_fine", "u_add")})
        assert step.one_call.why_not is None, step.one_call.why_not
        out = step.last_outputs
        same_z = bool(torch.equal(out.z_vals.cpu(), d[f"s{t}.out.z_vals"]))
        e_terms = float((torch.tensor(list(terms.values()), dtype=torch.float64) - d[f"s{t}.loss_terms"]).abs().max())
        e_loss = abs(float(loss) - float(d[f"s{t}.loss"])) / max(1.0, float(d[f"s{t}.loss"]))
        e_clip = abs(float(step.last_total_norm) - float(d[f"s{t}.clip_total_norm"])) / float(d[f"s{t}.clip_total_norm"])
        nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
        worst_w = 0.0
        for net, key, how in TRAINER_WATCH:
            got = watched_slice(dict(nets[net].named_parameters())[key].detach().cpu(), how)
            g = oracle_grads[t][f"{net}.{key}"]
            sig = g.abs() > 1e-2 * g.abs().max()
            worst_w = max(worst_w, float((got - d[f"s{t}.after.{net}.{key}"]).abs()[sig].max()))
        print(f"[one call, {colours} colours] step {t}: depths identical {same_z}; loss terms |d| {e_terms:.2e}; total rel {e_loss:.2e}; clip norm rel "
              f"{e_clip:.2e}; watched weights (significant gradients) off by {worst_w / lr:.3f} lr")
        assert abs(lr_now - float(d[f"s{t}.lr"])) < 1e-12
        if t == 0:
            assert same_z and e_terms < 1e-4 and e_loss < 1e-4 and e_clip < 1e-3 and worst_w < 0.02 * lr + 1e-7
        elif t == 1:
            # (the second step starts from weights that already differ by Adam's sign decisions on rounding-level gradients: the clip
            # norm of this 672-point batch is 21 % / 32 % off with the dense / sparse colour branch, 22 % launch by launch)
            assert same_z and e_terms < 1e-3 and e_clip < 0.5 and worst_w < 4.0 * lr
        else:
            assert e_loss < 0.1
    assert float(model.optimizer.state[model.vector_field_network.layers[8].weight]["step"]) == 2 * fx["steps"]
    assert abs(model.optimizer.param_groups[0]["lr"] - float(d["final_lr"])) < 1e-15


@pytest.mark.parametrize("colours", ["sparse", "dense"])
def test_reference_call_sequence_replays_the_reference_trainer_steps(colours):
    """VERDICT r04 next 1(a).  The reference's OWN train_epoch call sequence — render, functions.sample_border_points,
    vector_field_network(points)[:, :3], functions.get_center_indices_and_gt, functions.sample_center_points, the second network call, the
    torch.cat's, VFLoss, optimizer.zero_grad, backward, torch.nn.utils.clip_grad_norm_(model.parameters(), clip), optimizer.step,
    scheduler.step, loss.item() (train/vector_field_nerf_train.py:177-275, restated call for call in tools/reference_sequence.py on the
    names vf_nerf_amd.dropin installs) — replayed on the steps recorded from the reference's trainer (tests/golden/trainer_steps.npz),
    and it must TAKE THE STEP SESSION: the render and the backward one C call each on the training step's workspace (with the sparse
    colour branch by default), the supervision forwards on that workspace's rows, the centre-ball rows selected inside the loss kernels.
    Held to the bounds of the one-call replay above."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from reference_sequence import ReferenceLoop, StandInDataset
    from vf_nerf_amd import stepengine
    fx, d = load_trainer_fixture()
    model = build_model(fx, d, device=DEV)
    model.sparse_colour_training = colours == "sparse"
    model.scheduler = torch.optim.lr_scheduler.ExponentialLR(model.optimizer, lr_gamma(fx))
    crit = vloss.VFLoss(SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                        SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0))
    loop = ReferenceLoop(model, crit, StandInDataset(fx["centroid"], fx["far"]), fx["border_radius"])
    _, oracle_grads = _oracle_run(fx, d)
    lr = fx["lr"]
    eng = stepengine.StepEngine.of(model)
    seen_losses, seen_terms = [], []
    for t, b in enumerate(trainer_batches(fx, d, device=DEV)):
        supervision.replay_uniforms(b["border_u"], b["center_u"])
        loop.render_uniforms = {k: b[k] for k in ("u_coarse", "u_fine", "u_add")}
        lr_now = model.optimizer.param_groups[0]["lr"]
        data = {"uv": b["uv"].unsqueeze(0), "intrinsics": b["intrinsics"].unsqueeze(0), "pose": b["pose"].unsqueeze(0),
                "rgb": b["rgb_gt"].unsqueeze(0), "depth": b["depth_gt"].unsqueeze(0)}
        loss, terms = loop(data, fx["epoch"])
        session = eng.session
        assert eng.why_not is None and session is not None and session.backward_done, eng.why_not
        assert len(session.regions) == 2 and all(r["forwarded"] and not r["pending"] for r in session.regions.values())
        assert session.ray_centre is not None and session.ray_centre["consumed"]
        out = loop.last_outputs
        same_z = bool(torch.equal(out.z_vals.cpu(), d[f"s{t}.out.z_vals"]))
        e_terms = float((torch.tensor([terms[k] for k in vloss._NAMES], dtype=torch.float64) - d[f"s{t}.loss_terms"]).abs().max())
        e_loss = abs(float(loss) - float(d[f"s{t}.loss"])) / max(1.0, float(d[f"s{t}.loss"]))
        # the loop's own `loss.item()` / `losses_dict[key]` reads (train.py:262-275) are deferred scalars: numbers that equal the tensor's value
        from vf_nerf_amd.deferred import DeferredScalar
        assert isinstance(loss.item(), DeferredScalar) and loss.item() == float(loss) and isinstance(terms["rgb_loss"], DeferredScalar)
        seen_losses.append(float(loss))
        seen_terms.append([float(terms[k]) for k in vloss._NAMES])
        e_clip = abs(float(loop.last_total_norm) - float(d[f"s{t}.clip_total_norm"])) / float(d[f"s{t}.clip_total_norm"])
        nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
        worst_w = 0.0
        for net, key, how in TRAINER_WATCH:
            got = watched_slice(dict(nets[net].named_parameters())[key].detach().cpu(), how)
            g = oracle_grads[t][f"{net}.{key}"]
            sig = g.abs() > 1e-2 * g.abs().max()
            worst_w = max(worst_w, float((got - d[f"s{t}.after.{net}.{key}"]).abs()[sig].max()))
        print(f"[reference call sequence, {colours} colours] step {t}: depths identical {same_z}; loss terms |d| {e_terms:.2e}; total rel {e_loss:.2e}; "
              f"clip norm rel {e_clip:.2e}; watched weights (significant gradients) off by {worst_w / lr:.3f} lr")
        assert abs(lr_now - float(d[f"s{t}.lr"])) < 1e-12
        if t == 0:
            assert same_z and e_terms < 1e-4 and e_loss < 1e-4 and e_clip < 1e-3 and worst_w < 0.02 * lr + 1e-7
        elif t == 1:
            assert same_z and e_terms < 1e-3 and e_clip < 0.5 and worst_w < 4.0 * lr
        else:
            assert e_loss < 0.1
    assert float(model.optimizer.state[model.vector_field_network.layers[8].weight]["step"]) == 2 * fx["steps"]
    assert abs(model.optimizer.param_groups[0]["lr"] - float(d["final_lr"])) < 1e-15
    # the running sums the loop kept the way train_epoch keeps them (added on the device, step after step) are the sums of the steps' values
    avg = loop.average_losses
    assert abs(float(avg["loss"]) - sum(seen_losses)) < 1e-6 * max(1.0, sum(seen_losses))
    for j, k in enumerate(vloss._NAMES):
        assert abs(float(avg[k]) - sum(row[j] for row in seen_terms)) < 1e-6 * max(1.0, sum(row[j] for row in seen_terms)), k


def test_dropin_wraps_clip_grad_norm_for_the_duplicated_list():
    """`import vf_nerf_amd.dropin` leaves the trainer's own `torch.nn.utils.clip_grad_norm_(model.parameters(), c)` line
    (train/vector_field_nerf_train.py:254-255) correct on the GPU: duplicated device parameters go through the sequential
    semantics (clipped once per occurrence, Q4), anything else through PyTorch's function."""
    import vf_nerf_amd.dropin as dropin
    dropin._patch_clip_grad_norm()
    try:
        torch.manual_seed(1)
        a = [torch.nn.Parameter(torch.randn(5, 7, device=DEV)), torch.nn.Parameter(torch.randn(9, device=DEV)),
             torch.nn.Parameter(torch.randn(3, 3, device=DEV))]
        b = [torch.nn.Parameter(p.detach().cpu().clone()) for p in a]
        for p, q in zip(a, b):
            p.grad = torch.randn_like(p) * 3
            q.grad = p.grad.cpu().clone()
        na = torch.nn.utils.clip_grad_norm_(a + a[:2], 0.5)                           # patched: duplicates on the device
        nb = dropin._torch_clip(b + b[:2], 0.5, foreach=False)                        # PyTorch's sequential loop on the CPU
        assert abs(float(na) - float(nb)) < 1e-5 * float(nb)
        for p, q in zip(a, b):
            assert rel_err(p.grad, q.grad) < 1e-6
        n1 = torch.nn.utils.clip_grad_norm_(a, 0.1)                                    # no duplicates: PyTorch's own path
        assert float(n1) > 0
    finally:
        dropin.uninstall_clip_grad_norm()


def test_reference_written_checkpoint_renders_on_hip():
    """load() of the checkpoint the REFERENCE wrote (narrow model: hidden widths 64 / 32, so the layer-at-a-time row kernels
    run, not the fused ones), then the VF forward and one render() with the reference's draws against what the reference
    itself computed from that state."""
    _, d = load_trainer_fixture()
    model, c = narrow_checkpoint_model(d, device=DEV)
    assert not model.vector_field_network.supports_fused() and not model.uses_f16x3()
    assert model.load(REF_CHECKPOINT) == c["saved_epoch"] + 1
    g = {k[len("ckpt."):]: v.to(DEV) for k, v in d.items() if k.startswith("ckpt.")}
    with torch.no_grad():
        vf_out = model.vector_field_network(g["probe_points"])
        assert vf_out.shape == d["ckpt.vf_out"].shape and rel_err(vf_out, d["ckpt.vf_out"]) < 2e-5
        out = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add")})
    assert torch.equal(out.z_vals.cpu(), d["ckpt.z_vals"])
    for got, key in ((out.coarse_rgb_values, "rgb"), (out.coarse_depth_map, "depth"), (out.coarse_normals, "normals"),
                     (out.coarse_colors, "colors")):
        assert rel_err(got.reshape(d[f"ckpt.{key}"].shape), d[f"ckpt.{key}"]) < 5e-5, key
    assert float(d["ckpt.depth"].min()) < 0, "the fixture keeps a ray whose un-clamped fine window reaches negative depths (Q9)"
    with tempfile.TemporaryDirectory() as tmp:           # and back: a checkpoint saved from the device loads in the CPU layout
        model.save(5, tmp)
        again = torch.load(os.path.join(tmp, "5.pth"), map_location="cpu")
    ref = torch.load(REF_CHECKPOINT, map_location="cpu")
    assert all(torch.equal(again["vf_net"][k], ref["vf_net"][k]) for k in ref["vf_net"])


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit", "shipped_sizes", "c1_det"])
def test_sampler_objects_sample_bit_exact(name):
    """ray_sampler.sample / fine_sampler.sample / get_z_vals called on the sampler OBJECTS (ray_sampler.py:49-80,113-142,
    264-302) with the reference's draws: depths and points bit-identical to the golden vectors, per-ray far included."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device=DEV)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    rs, fs = model.ray_sampler, model.fine_sampler
    if fx["perturb"]:
        rs.replay = [g["u_coarse"]]
    pts_c, z_c = rs.sample(g["directions"], g["cam_loc"], device=torch.device(DEV))
    assert torch.equal(z_c.cpu(), d["z_coarse"])
    if fx["perturb"]:
        rs.replay = [g["u_coarse"]]
    assert torch.equal(rs.get_z_vals(g["directions"], g["cam_loc"]).cpu(), d["z_coarse"])
    fs.replay = ([g["u_fine"]] if fx["perturb"] else []) + [g["u_add"]]
    pts, z = fs.sample(g["directions"], g["cam_loc"], device=torch.device(DEV), coarse_z_vals=z_c, coarse_weights=g["weights_coarse"])
    assert torch.equal(z.cpu(), d["z_vals"]) and torch.equal(pts.cpu(), d["points"])
    assert torch.equal(lib.rows_argmax(g["weights_coarse"]).cpu(), d["max_indices"])
    # production draws: the sampler's own Philox stream, advancing between calls
    a = rs.get_z_vals(g["directions"], g["cam_loc"])
    b = rs.get_z_vals(g["directions"], g["cam_loc"])
    assert (not fx["perturb"]) == bool(torch.equal(a, b))
    # additional depths are merged and the points recomputed (ray_sampler.py:69-78)
    extra = torch.full((z_c.shape[0], 2), 0.333, device=DEV)
    if fx["perturb"]:
        rs.replay = [g["u_coarse"]]
    p2, z2 = rs.sample(g["directions"], g["cam_loc"], additional_depths=extra)
    assert z2.shape[1] == z_c.shape[1] + 2 and bool((z2[:, 1:] >= z2[:, :-1]).all())
    assert rel_err(p2, g["cam_loc"].unsqueeze(1) + z2.unsqueeze(2) * g["directions"].unsqueeze(1)) < 1e-6


def test_rows_argmax_ties_and_empty_rows():
    w = torch.zeros(7, 130, device=DEV)
    w[1, 5] = w[1, 77] = 2.0            # tie: the first wins
    w[2, 129] = 1.0
    w[3] = -1.0
    w[3, 64] = -0.5
    got = lib.rows_argmax(w).cpu()
    assert got.tolist() == torch.argmax(w.cpu(), dim=-1).tolist() == [0, 5, 129, 64, 0, 0, 0]


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit"])
def test_get_density_is_part_of_the_graph(name):
    """VectorFieldNerf.get_density under autograd (vector_field_nerf.py:442-474 is differentiable in the reference): gradients
    wrt the normals and the three density scalars against the oracle's autograd of the same function."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device=DEV)
    n, s_t = d["z_vals"].shape
    gen = torch.Generator().manual_seed(3)
    coef = torch.randn(n, s_t, generator=gen)
    normals = d["normals"].clone().requires_grad_(True)
    beta, mean, scale = (torch.tensor(v, requires_grad=True) for v in (0.5, 0.7, 100.0))
    st = oracle_settings(fx)
    sigma_ref = O.ray_density(normals, d["ray_dirs"], st.n_window, st.dir_to_normal_th, st.density, beta, mean, scale)
    (sigma_ref * coef).sum().backward()
    nd = d["normals"].to(DEV).requires_grad_(True)
    rep = d["ray_dirs"].to(DEV).unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
    for p in model.density.parameters():
        p.grad = None
    sigma = model.get_density(nd, rep, True)
    assert sigma.requires_grad and rel_err(sigma, sigma_ref.detach()) < 1e-6
    (sigma * coef.to(DEV)).sum().backward()
    assert grad_rel_err(nd.grad, normals.grad) < 1e-4
    for p_name, ref in (("beta", beta), ("mean", mean), ("scale", scale)):
        got = getattr(model.density, p_name).grad
        assert got is not None and grad_rel_err(got.reshape(1), ref.grad.reshape(1)) < 1e-4, p_name
    with torch.no_grad():
        assert not model.get_density(nd, rep).requires_grad


def test_attached_normals_gradients_on_hip():
    """detach_normals=False (rendering_network.py:76-77; the shipped conf sets True): the colours' gradient reaches the
    normals.  The fused dX chain is written for the detached case, so render() calls the networks one after the other here;
    gradients against the reference's own backward and the oracle's."""
    fx, d = load_fixture("attached_normals")
    model = build_model(fx, d, device=DEV)
    assert model.rendering_network.config.detach_normals is False
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    for p in model.unique_parameters():
        p.grad = None
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add")})
    assert torch.equal(out.z_vals.cpu(), d["z_vals"])
    a, b, c = (t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape))
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) <= 1e-4 * max(1.0, abs(float(d["loss"])))
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    worst = 0.0
    for tag, key in GRAD_KEYS:
        err = grad_rel_err(dict(nets[tag].named_parameters())[key].grad, d[f"grad.{tag}.{key}"])
        worst = max(worst, err)
        assert err < 5e-3, (tag, key, err)     # 312 points: one ReLU unit on the other side of zero is 0.3 % of a layer's gradient
    print(f"attached normals: worst gradient error vs the reference's backward {worst:.2e}")
    # and the setting matters: the detached model's VF gradient differs by far more than the tolerance
    _, detached = oracle_gradients(dict(fx, detach_normals=True), d, build_model(fx, d))
    apart = max(grad_rel_err(p.grad, detached[f"vf.{k}"]) for k, p in nets["vf"].named_parameters())
    print(f"attached vs detached normals: VF gradients up to {apart:.2e} apart")
    assert apart > 3 * worst


def test_numerical_jacobian_gradients_have_the_right_values():
    """numerical_jacobian=True under autograd: the six offset forwards of the fine pass run the exact-fp32 kernels while the
    model's precision is f16x3; their backward must use the kernels that match what THOSE forwards saved (it used to re-derive
    the choice from the live precision and read sign masks that were never written).  Gradients of the reference's functional
    (make_golden.capture_grads, directional-derivative term included) against the all-fp32 HIP run (tight: the offset
    forwards are the same launches) and against the reference's own backward (loose: a central difference with eps = 1e-5
    amplifies rounding by 1e5 in value, and its derivative wrt the weights by as much)."""
    fx, d = load_fixture("numjac_det")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    gen = torch.Generator().manual_seed(4343)
    grads = {}
    for precision in ("f16x3", "fp32"):
        model = build_model(fx, d, device=DEV)
        model.precision = precision
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={"u_add": g["u_add"]})
        assert model.precision == precision and model.vector_field_network.precision == precision
        a, b, c = (t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape))
        dd = out.directional_derivtives
        w_dd = (1e-3 * torch.rand(dd.shape[0], generator=torch.Generator().manual_seed(4343))).to(DEV)
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum() + (dd * w_dd).sum()
        loss.backward()
        grads[precision] = {f"{tag}.{k}": p.grad.detach().cpu().clone() for tag, net in
                            (("vf", model.vector_field_network), ("rn", model.rendering_network)) for k, p in net.named_parameters()}
    del gen
    worst_pair, worst_ref = 0.0, 0.0
    for key, v in grads["f16x3"].items():
        assert torch.isfinite(v).all(), key
        worst_pair = max(worst_pair, grad_rel_err(v, grads["fp32"][key]))
    for tag, key in GRAD_KEYS:
        worst_ref = max(worst_ref, grad_rel_err(grads["f16x3"][f"{tag}.{key}"], d[f"grad.{tag}.{key}"]))
    print(f"numerical Jacobian gradients: f16x3 vs fp32 HIP runs {worst_pair:.2e}; vs the reference's backward {worst_ref:.2e}")
    assert worst_pair < 2e-2 and worst_ref < 0.2


def test_flat_adam_and_fused_clip_match_the_sequential_loops():
    """optim.FlatAdam (one flat buffer, csrc/vfn_adam.hip: one launch per step) and the fused clip_grad_norm_ on the device against
    torch.optim.Adam(foreach=False) / torch.nn.utils.clip_grad_norm_(foreach=False) on the CPU — the sequential per-entry loops
    the reference ran with — over a parameter list that names some tensors twice (Q4): total norm, scaled gradients, parameters,
    moments and step counters over several steps, then a state_dict round trip in both directions."""
    from vf_nerf_amd import optim
    shapes = [(7, 5), (5,), (3, 3), (4,), (130, 70)]

    def make(device):
        torch.manual_seed(4)
        ps = [torch.nn.Parameter(torch.randn(*s).to(device)) for s in shapes]
        return ps, ps + ps[:2]                 # the first two are listed twice

    pa, la = make("cpu")
    pb, lb = make(DEV)
    oa = torch.optim.Adam(la, lr=3e-3, foreach=False)
    ob = optim.FlatAdam(lb, lr=3e-3)
    gen = torch.Generator().manual_seed(9)
    for it in range(6):
        ob.zero_grad()
        g = [torch.randn(*s, generator=gen) * (10.0 if it == 2 else 0.1) for s in shapes]
        for p, gg in zip(pa, g):
            p.grad = gg.clone()
        for p, gg in zip(pb, g):
            p.grad.add_(gg.to(DEV))               # autograd accumulates into the flat views
        na = torch.nn.utils.clip_grad_norm_(la, 0.5, foreach=False)
        nb = optim.clip_grad_norm_(lb, 0.5)
        assert abs(float(na) - float(nb)) <= 2e-6 * float(na), (it, float(na), float(nb))
        for a, b in zip(pa, pb):
            assert rel_err(b.grad, a.grad) < 2e-6, it
        oa.step()
        ob.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert float((a.detach() - b.detach().cpu()).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max())), (it, i)
    f = ob.flat()
    assert f is not None and all(p.grad.data_ptr() == f["grad"].data_ptr() + 4 * off for p, off, _, _ in f["entries"])
    for a, b in zip(pa, pb):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"])
        assert rel_err(sb["exp_avg"], sa["exp_avg"]) < 1e-5 and rel_err(sb["exp_avg_sq"], sa["exp_avg_sq"]) < 1e-5
    assert float(ob.state[pb[0]]["step"]) == 12.0 and float(ob.state[pb[2]]["step"]) == 6.0
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert sda["param_groups"][0]["params"] == sdb["param_groups"][0]["params"] and sda["state"].keys() == sdb["state"].keys()
    # a checkpoint written by torch.optim.Adam loads into FlatAdam (and the next step continues from it) and vice versa
    pc, lc = make(DEV)
    oc = optim.FlatAdam(lc, lr=3e-3)
    oc.load_state_dict(sda)
    with torch.no_grad():
        for c, a in zip(pc, pa):
            c.copy_(a.detach().to(DEV))
    g = [torch.randn(*s, generator=gen) * 0.1 for s in shapes]
    oc.zero_grad()
    for p, gg in zip(pc, g):
        p.grad.add_(gg.to(DEV))
    for p, gg in zip(pa, g):
        p.grad = gg.clone()
    oa.step()
    oc.step()
    for a, c in zip(pa, pc):
        assert float((a.detach() - c.detach().cpu()).abs().max()) <= 2e-6 * max(1.0, float(a.abs().max()))
    od = torch.optim.Adam(make("cpu")[1], lr=3e-3, foreach=False)
    od.load_state_dict(oc.state_dict())
    assert float(od.state[od.param_groups[0]["params"][0]]["step"]) == 14.0


def test_model_optimizer_is_flat_and_survives_load_and_repacking():
    """The facade's optimizer: FlatAdam over the duplicated list (805 780 unique elements, the VF parameters in the region of
    multiplicity 2); a step changes what render() computes (the weight packs are invalidated although the update goes through
    raw pointers); load() of a checkpoint and .to() keep everything bound."""
    from vf_nerf_amd import optim
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device=DEV)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
    assert isinstance(model.optimizer, optim.FlatAdam)
    with torch.no_grad():
        before = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms=uni).coarse_rgb_values.clone()
    out = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms=uni)
    model.optimizer.zero_grad()
    (out.coarse_rgb_values.sum() + out.coarse_normals.pow(2).sum()).backward()
    norm = optim.clip_grad_norm_(model.parameters(), 0.5)
    model.optimizer.step()
    f = model.optimizer.flat()
    assert f["param"].numel() == 805780 and f["regions"][0] == (0, 531342, 2) and f["regions"][1][2] == 1
    assert float(norm) > 0 and float(model.optimizer.state[model.vector_field_network.layers[0][0].weight]["step"]) == 2.0
    with torch.no_grad():
        after = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms=uni).coarse_rgb_values
    assert float((after - before).abs().max()) > 1e-6, "the step must reach the kernels' weight packs"
    with tempfile.TemporaryDirectory() as tmp:
        model.save(3, tmp)
        other = build_model(fx, d, device=DEV)
        assert other.load(os.path.join(tmp, "latest.pth")) == 4
    with torch.no_grad():
        again = other.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms=uni).coarse_rgb_values
    assert torch.equal(again, after)
    st = other.optimizer.state[other.vector_field_network.layers[0][0].weight]
    assert float(st["step"]) == 2.0 and float(st["exp_avg"].abs().max()) > 0
    out = other.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms=uni)          # and training continues from it
    other.optimizer.zero_grad()
    out.coarse_rgb_values.sum().backward()
    other.optimizer.step()
    assert float(other.optimizer.state[other.vector_field_network.layers[0][0].weight]["step"]) == 4.0


def test_argmax_nan_rule_is_the_same_in_every_kernel():
    """torch.argmax's order (ray_sampler.py:277 takes torch.argmax of the proposal weights): the first maximum, and a NaN beats
    every number (the first NaN wins).  The stand-alone row argmax and the argmax inside the density / composite launch share
    one comparison (csrc/vfn_rays.hip: argmax_takes), so they agree with torch and with each other on rows holding NaNs."""
    w = torch.rand(6, 130, device=DEV)
    w[1, 70] = float("nan")
    w[2, 3] = w[2, 99] = float("nan")
    w[3, 129] = float("nan")
    w[4] = float("nan")
    want = torch.argmax(w.cpu(), dim=-1)
    assert want.tolist()[1:5] == [70, 3, 129, 0]
    assert torch.equal(lib.rows_argmax(w).cpu(), want)
    # the same rule inside vfn_ray_density_weights (a NaN depth makes the weights NaN from that sample on; a NaN normal does not:
    # the cosine's clamped norms drop it): the launch's argmax is torch's argmax of the launch's own weights, and the stand-alone
    # kernel agrees
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device=DEV)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    z = g["z_coarse"].clone().contiguous()
    z[5, 9] = float("nan")
    z[7, 0] = float("nan")
    _, wts, imax, _, _ = lib.ray_density_weights(model._density_params(), g["normals_coarse"].contiguous(), g["ray_dirs"].contiguous(), z,
                                                 model.density.raw_scalars(), want_argmax=True)
    print("rays with NaN weights:", int(torch.isnan(wts).any(dim=1).sum()))
    assert torch.equal(imax.cpu(), torch.argmax(wts.cpu(), dim=-1)), "argmax of the launch's own weights, torch order"
    assert torch.equal(lib.rows_argmax(wts).cpu(), imax.cpu())


@pytest.mark.parametrize("storage", ["f16+f16", "fp32+fp32", "f16+bf16"])
def test_non_finite_upstream_gradient_reaches_the_weight_gradients(storage):
    """A diverging step must not be masked: the reference's fp32 autograd turns an Inf in d(loss)/d(rgb) into non-finite
    parameter gradients (and clip_grad_norm_ into a NaN norm).  The scaled-f16 gradient storage used to encode a tile whose
    largest magnitude is Inf as "all zero" and to drop NaNs from the maximum (csrc/vfn_bwd16.hip: tile_scale); now the lane's
    values leave as NaN.  Every storage: each watched weight gradient is non-finite somewhere and the clip norm is not finite."""
    fx, d = load_fixture("bench_sizes")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    model = build_model(fx, d, device=DEV)
    model.activation_storage, model.gradient_storage = storage.split("+")
    out = model.render(g["pose"], g["uv"], g["intrinsics"], 0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add")})
    coeff = torch.ones_like(out.coarse_rgb_values)
    hit = int(torch.argmax(out.coarse_depth_map.detach().reshape(-1)))          # a ray that composites something
    coeff[hit, 1] = float("inf")
    model.optimizer.zero_grad()
    (out.coarse_rgb_values * coeff).sum().backward()
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    for net, key in (("rn", "layers.4.weight"), ("rn", "layers.1.0.weight"), ("vf", "layers.8.weight"), ("vf", "layers.5.0.weight"), ("vf", "layers.0.0.weight")):
        grad = dict(nets[net].named_parameters())[key].grad
        assert not bool(torch.isfinite(grad).all()), f"{storage}: {net}.{key} came out finite from an infinite upstream gradient"
    norm = optim.clip_grad_norm_(model.parameters(), 0.5)
    assert not bool(torch.isfinite(norm)), float(norm)


def test_training_converges_on_teacher_targets():
    """SURVEY.md section 8d C3 in small: the reference trainer's step (trainer.TrainStep = train/vector_field_nerf_train.py:172-260)
    on a LEARNABLE target — rgb / depth of eight orbit views rendered by a teacher of another weight seed — from the same initial
    weights, batches and random streams with the exact-fp32 kernels and with the default 16-bit path (f16x3 forward, f16
    activations, scaled-f16 gradients).  300 steps of 1 024 rays x (64 + 64): both start from the same loss, both bring it below
    0.6 of where it started (observed: 0.4 .. 0.5), and neither ends far from the other.  Single runs are chaotic (a ReLU unit or
    an argmax on the other side flips a discrete event and the trajectories part): the band here is wide on purpose; the long
    comparison with three random streams per arithmetic and its 0.3 dB band is tools/train_curve.py ->
    profiles/r03/train_curve.json."""
    import vf_nerf_amd
    from vf_nerf_amd import synthetic, trainer
    dev = torch.device(DEV)

    def scene(seed):
        torch.manual_seed(seed)
        cfg = vf_nerf_amd.shipped_config(dev, n_samples=64, n_importance=64, perturb=True, dir_to_normal_th=-0.2)
        m = vf_nerf_amd.VectorFieldNerf(cfg)
        m.eval()
        synthetic.scale_hidden_weights(m.vector_field_network, m.rendering_network, 2.0)
        with torch.no_grad():
            pts = synthetic.frustum_points(20000, seed=1234).to(dev)
            keep = m.precision
            m.precision = "fp32"
            mean, std = synthetic.vector_head_stats_from_tanh(m.vector_field_network(pts, vector_only=True))
            m.precision = keep
            synthetic.recentre_vector_head(m.vector_field_network, mean, std)
        return m

    pool = trainer.TeacherTargets(scene(1), views=8, width=64, height=64, focal=60.0, seed=5)
    curves = {}
    for tag in ("fp32", "f16x3"):
        model = scene(0)
        model.precision = tag
        model.rng_seed, model._rng_offset = 11, 0
        supervision.manual_seed(3)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
        losses = []
        for t in range(300):
            pose, uv, K, rgb_gt, depth_gt = pool.batch(t, 1024)
            losses.append(step(pose, uv, K, rgb_gt, depth_gt, epoch=0)[0])
        curves[tag] = [float(x) for x in losses]
        first, last = sum(curves[tag][:10]) / 10, sum(curves[tag][-50:]) / 50
        print(f"[{tag}] loss first 10 steps {first:.4f} -> last 50 steps {last:.4f} (x{last / first:.3f})")
        assert last < 0.6 * first, (tag, first, last)
    a, b = sum(curves["fp32"][-50:]) / 50, sum(curves["f16x3"][-50:]) / 50
    print(f"final loss f16x3 / fp32 = {b / a:.4f}; first-step losses {curves['fp32'][0]:.6f} / {curves['f16x3'][0]:.6f}")
    assert abs(curves["f16x3"][0] - curves["fp32"][0]) < 1e-4 * curves["fp32"][0], "same state, same batch: the same first loss"
    assert 0.6 < b / a < 1.67


def test_render_after_training_stays_inside_the_contract():
    """What a trained model's gradient-free render returns: 1 500 optimizer steps of the reference trainer's
    step (trainer.TrainStep, train/vector_field_nerf_train.py:172-260) on 1 024-ray batches of a teacher-rendered target, then the
    evaluator's kind of call (evaluation/methods.py:528: render under no_grad) with the range guard in strict mode:

    * the DEFAULT path (three f16 products everywhere) returns colours / rgb / depth within 1e-4 of the exact-fp32 kernels on the same
      weights, rays and draws, samples the same depths bit for bit, and the guard has nothing to report;
    * the OPT-IN two-product colour branch is measured on that state: its raw difference to three products is printed, and when it is
      above the guard's tolerance the strict guard must have refused it before the call returned (what came back is the three-product
      render).  Round 3's finding, now asserted: trained states are out of that mode's reach."""
    import warnings
    import vf_nerf_amd
    from vf_nerf_amd import guard as vguard, synthetic, trainer
    dev = torch.device(DEV)

    def scene(seed):
        torch.manual_seed(seed)
        cfg = vf_nerf_amd.shipped_config(dev, n_samples=64, n_importance=64, perturb=True, dir_to_normal_th=-0.2)
        m = vf_nerf_amd.VectorFieldNerf(cfg)
        m.eval()
        synthetic.scale_hidden_weights(m.vector_field_network, m.rendering_network, 2.0)
        with torch.no_grad():
            pts = synthetic.frustum_points(20000, seed=1234).to(dev)
            mean, std = synthetic.vector_head_stats_from_tanh(m.vector_field_network(pts, vector_only=True))
            synthetic.recentre_vector_head(m.vector_field_network, mean, std)
        return m

    pool = trainer.TeacherTargets(scene(1), views=8, width=64, height=64, focal=60.0, seed=5)
    model = scene(0)
    assert model.colour_products == 3 and model.precision == "f16x3"
    model.rng_seed, model._rng_offset = 11, 0
    supervision.manual_seed(3)
    step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
    losses = []
    for t in range(1500):
        pose, uv, K, rgb_gt, depth_gt = pool.batch(t, 1024)
        losses.append(step(pose, uv, K, rgb_gt, depth_gt, epoch=0)[0])
    first, last = float(sum(losses[:10])) / 10, float(sum(losses[-50:])) / 50
    print(f"trained 1500 steps x 1024 rays: loss {first:.4f} -> {last:.4f}; guard: fp32 switch {model.f16x3_disabled}")
    assert last < 0.6 * first and model.f16x3_disabled is None

    pose, uv, K, _, _ = pool.batch(900_000, 1024)
    g = torch.Generator().manual_seed(4)
    uni = {"u_coarse": torch.rand(1024, 64, generator=g).to(dev), "u_fine": torch.rand(1024, 64, generator=g).to(dev),
           "u_add": torch.rand(1024, 64, generator=g).to(dev)}

    def render(products, precision="f16x3", guard="strict"):
        model.colour_products, model.precision, model.f16x3_guard = products, precision, guard
        model.range_guard.colour_products_reason = None
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with torch.no_grad():
                out = model.render(pose, uv, K, epoch=0, uniforms=uni)
        return out, [str(w.message) for w in caught if issubclass(w.category, RuntimeWarning)]

    exact, _ = render(3, "fp32")
    default, warned = render(3)
    assert not warned and model.colour_products == 3 and model.f16x3_disabled is None and model.range_guard.check_now(dev) is None
    assert torch.equal(default.z_vals, exact.z_vals), "the trained model's sample depths: bit-identical to the exact-fp32 kernels'"
    errs = {k: float((getattr(default, k) - getattr(exact, k)).abs().max()) for k in ("coarse_colors", "coarse_rgb_values", "coarse_depth_map", "coarse_normals")}
    print("default path (three products) vs the exact-fp32 kernels after training:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert errs["coarse_colors"] < 2e-5 and errs["coarse_normals"] < 2e-5, errs
    inside = ((default.coarse_rgb_values - exact.coarse_rgb_values).abs().max(dim=1)[0] < 1e-4) & \
        ((default.coarse_depth_map - exact.coarse_depth_map).abs().reshape(-1) < 1e-4)
    print(f"rays inside 1e-4 (rgb and depth): {float(inside.float().mean()):.4f}")
    assert float(inside.float().mean()) >= 0.998          # (the density's conditioning: a few rays per thousand sit on a threshold)

    raw2, _ = render(2, guard="off")
    raw_diff = float((raw2.coarse_colors - default.coarse_colors).abs().max())
    strict2, warned2 = render(2)
    ran = model.colour_products
    print(f"opt-in two-product colours after training: raw difference to three products {raw_diff:.2e} (guard tolerance {vguard.COLOUR_CHECK_TOL:.0e}, "
          f"contract 1e-4); strict guard ran the call on {ran} products ({model.range_guard.colour_products_reason or 'kept two'})")
    assert torch.equal(raw2.z_vals, default.z_vals) and torch.equal(raw2.coarse_depth_map, default.coarse_depth_map)
    if ran == 3:
        assert any("colour_products" in m for m in warned2) and torch.equal(strict2.coarse_colors, default.coarse_colors)
    else:
        assert float((strict2.coarse_colors - default.coarse_colors).abs().max()) < 1e-4
    # the self-check samples 128 rays of the call; it must catch anything that is outside the CONTRACT on the whole call
    assert ran == 3 or raw_diff < 1e-4


def test_evaluator_loop_through_the_dropin_gets_the_chunked_values():
    """The evaluator renders a view chunk by chunk (evaluation/methods.py:513-545): per chunk it uploads pixels / pose / intrinsics,
    calls model.render(pose, pixels, intrinsics, epoch, white) and pulls coarse_rgb_values / coarse_depth_map back with .cpu().
    That loop, restated here shape for shape on host tensors as the dataset hands them over (per-ray pose / intrinsics, a last
    chunk that is not full), against evaluator.render_view — what vf_nerf_amd.dropin puts behind evaluation.methods.render_images:
    chunks on two streams, one download.  Same random stream position -> the same values, bit for bit."""
    import numpy as np
    from vf_nerf_amd import evaluator, synthetic
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device=DEV)
    w, h, split = 40, 27, 256                                  # 1080 rays: four full chunks and one of 56
    uv, pose, K = synthetic.pinhole_image(w, h, 38.0, pose=synthetic.orbit_pose(20.0, 8.0, 0.9))
    assert not uv.is_cuda and pose.shape == (w * h, 4, 4)
    device = torch.device(DEV)

    model.rng_seed, model._rng_offset = 5, 0
    rgb = np.zeros((h, w, 3))
    depth_map = np.zeros((h, w, 1))
    n = uv.shape[0]
    pixels_split, pose_split, k_split = torch.split(uv, split, dim=0), torch.split(pose, split, dim=0), torch.split(K, split, dim=0)
    with torch.no_grad():
        for j in range(int(np.ceil(n / split))):
            pixels = pixels_split[j].to(device)
            output = model.render(pose_split[j].to(device), pixels, k_split[j].to(device), 0, False)
            rows, cols = pixels[:, 1].long().cpu().numpy(), pixels[:, 0].long().cpu().numpy()
            rgb[rows, cols, :] = output.coarse_rgb_values.cpu().numpy()
            depth_map[rows, cols, :] = output.coarse_depth_map.cpu().numpy()

    model.rng_seed, model._rng_offset = 5, 0
    rgb_values, depth_values = evaluator.render_view(model, pose, uv, K, 0, split_size=split, min_chunk=0)      # the reference's chunking
    assert rgb_values.shape == (n, 3) and depth_values.shape == (n, 1)
    rows, cols = uv[:, 1].long().numpy(), uv[:, 0].long().numpy()
    assert np.array_equal(rgb[rows, cols, :].astype(np.float32), rgb_values)
    assert np.array_equal(depth_map[rows, cols, :].astype(np.float32), depth_values)
    assert float(depth_values.max()) > 0.05, "the view composites something"
    # a pose / intrinsics shared by the whole view (one copy) gives the same image
    model.rng_seed, model._rng_offset = 5, 0
    rgb_shared, _ = evaluator.render_view(model, pose[0], uv, K[0], 0, split_size=split, min_chunk=0)
    assert np.array_equal(rgb_shared, rgb_values)
    # The default groups the view into larger chunks than the reference's split (rays are independent; its split bounds ITS
    # activation memory).  With deterministic samplers the only draw left is the uniform fine samples of rays WITHOUT a surface
    # (Q9): rays whose value does not depend on the seed must not depend on the chunking either, bit for bit.
    model.ray_sampler.deterministic = model.fine_sampler.deterministic = True
    views = {}
    for tag, seed, mc in (("split", 5, 0), ("split, other seed", 6, 0), ("grouped", 5, evaluator.MIN_CHUNK)):
        model.rng_seed, model._rng_offset = seed, 0
        views[tag] = evaluator.render_view(model, pose, uv, K, 0, split_size=split, min_chunk=mc)
    draw_free = (views["split"][0] == views["split, other seed"][0]).all(axis=1) & (views["split"][1] == views["split, other seed"][1]).all(axis=1)
    assert draw_free.mean() > 0.3, "the view has rays with a surface"
    assert np.array_equal(views["grouped"][0][draw_free], views["split"][0][draw_free])
    assert np.array_equal(views["grouped"][1][draw_free], views["split"][1][draw_free])


def test_frozen_parameters_are_left_alone_by_the_training_step():
    """requires_grad_(False) on parameters of both nets: the in-place gradient mode and the flat Adam kernel would
    write every element of their flat buffers, so such a model takes autograd's own accumulation and the per-parameter Adam
    path — frozen tensors keep their values bit for bit and get neither a gradient nor optimizer state, the others train; the
    supervision forward's loss still reaches the vector-field net although its FIRST parameter is frozen."""
    from vf_nerf_amd import trainer
    fx, d = load_fixture("c1_perturb")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    model = build_model(fx, d, device=DEV)
    vf, rn = model.vector_field_network, model.rendering_network
    frozen = [vf.layers[0][0].weight, vf.layers[3][0].bias, rn.layers[1][0].weight]
    for p in frozen:
        p.requires_grad_(False)
    before = [p.detach().clone() for p in frozen]
    moving = [vf.layers[5][0].weight, rn.layers[4].weight, vf.layers[8].weight]
    moving_before = [p.detach().clone() for p in moving]
    step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
    gen = torch.Generator().manual_seed(1)
    rgb_gt, depth_gt = torch.rand(fx["n_rays"], 3, generator=gen).to(DEV), torch.rand(fx["n_rays"], 1, generator=gen).to(DEV)
    for _ in range(2):
        loss, _ = step(g["pose"], g["uv"], g["intrinsics"], rgb_gt, depth_gt, epoch=0)
        assert bool(torch.isfinite(loss))
    for p, b in zip(frozen, before):
        assert torch.equal(p.detach(), b) and p.grad is None and len(model.optimizer.state.get(p, {})) == 0
    for p, b in zip(moving, moving_before):
        assert not torch.equal(p.detach(), b)
    assert float(model.optimizer.state[vf.layers[8].weight]["step"]) == 4 and float(model.optimizer.state[rn.layers[4].weight]["step"]) == 2
    # supervision only: the loss reaches the vector-field net through autograd although its first parameter is frozen
    model.optimizer.zero_grad()
    pts = torch.rand(64, 3, device=DEV)
    (model.vector_field_network(pts)[:, :3] ** 2).sum().backward()
    assert vf.layers[5][0].weight.grad is not None and float(vf.layers[5][0].weight.grad.abs().sum()) > 0


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit"])
def test_sampler_additional_depths_sorted_on_the_device(name):
    """RaySampler.sample(..., additional_depths) (ray_sampler.py:69-73: cat, torch.sort, points recomputed) through the device
    merge-sort (vfn_merge_sort_depths), against torch's own sort of the same depths; duplicates, a NaN (sorts last) and a number of
    depths that is not a power of two included."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device=DEV)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    sampler = model.ray_sampler
    sampler.deterministic = True
    n = fx["n_rays"]
    gen = torch.Generator().manual_seed(3)
    extra = (fx["near"] + (fx["far"] - fx["near"]) * torch.rand(n, 7, generator=gen)).to(DEV)
    extra[:, 3] = extra[:, 1]                       # duplicates
    extra[2, 5] = float("nan")
    pts, z = sampler.sample(g["directions"].reshape(n, 1, 3), g["cam_loc"], additional_depths=extra)
    z_plain = sampler.get_z_vals(g["directions"].reshape(n, 1, 3), g["cam_loc"])
    want, _ = torch.cat((z_plain, extra), dim=1).sort(dim=1)
    assert z.shape == (n, fx["n_samples"] + 7) and pts.shape == (n, fx["n_samples"] + 7, 3)
    same = (z == want) | (torch.isnan(z) & torch.isnan(want))
    assert bool(same.all()) and bool(torch.isnan(z[2, -1])) and int(torch.isnan(z).sum()) == 1
    ok = ~torch.isnan(z)
    want_pts = g["cam_loc"].unsqueeze(1) + want.unsqueeze(2) * g["directions"].reshape(n, 1, 3)
    assert float((pts - want_pts)[ok].abs().max()) < 1e-6


def test_fused_loss_matches_the_reference_vfloss_golden_on_the_device():
    """vf_nerf_amd.loss.VFLoss through the fused kernels (csrc/vfn_loss.hip) against the outputs of the reference's own VFLoss.forward
    (tests/golden/trainer_steps.npz, models/losses/vf_loss.py:34-87), every branch, values AND the gradients torch's autograd gives
    for the tensor-op formulation of the same terms."""
    fx, d = load_trainer_fixture()
    base = {k[len("loss.in."):]: v.to(DEV) for k, v in d.items() if k.startswith("loss.in.")}
    cfg = SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100)
    w = SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.3)
    for name in ("early", "late", "dd_before_start", "no_depth_no_sup"):
        epoch, dd, depth, sup = [int(v) for v in d[f"loss.{name}.case"]]
        grads = {}
        for fused in (True, False):
            crit = vloss.VFLoss(cfg, w)
            crit.fused = fused
            leaves = {k: base[k].clone().requires_grad_(True) for k in ("rgb", "depth", "normals", "sup")}
            pred = {"rgb": leaves["rgb"], "depth": leaves["depth"], "normals": leaves["normals"],
                    "supervised_normals": leaves["sup"] if sup else torch.empty(0, 3, device=DEV), "directional_derivatives": base["dd"] if dd else None}
            gt = {"rgb": base["rgb_gt"], "depth": base["depth_gt"] if depth else torch.empty(0, device=DEV),
                  "supervised_normals": base["sup_gt"] if sup else torch.empty(0, device=DEV)}
            loss, logs = crit(pred, gt, epoch)
            assert abs(float(loss) - float(d[f"loss.{name}.total"])) <= 1e-6 * max(1.0, float(d[f"loss.{name}.total"])), (name, fused)
            got = torch.tensor(list(logs.values()), dtype=torch.float64)
            assert list(logs) == list(vloss._NAMES) and float((got - d[f"loss.{name}.terms"]).abs().max()) <= 1e-6, (name, fused, got)
            (loss * 1.7).backward()
            grads[fused] = {k: (v.grad.clone() if v.grad is not None else None) for k, v in leaves.items()}
        for k in grads[True]:
            a, b = grads[True][k], grads[False][k]
            assert (a is None) == (b is None) or (a is None and float(b.abs().max()) == 0) or (b is None and float(a.abs().max()) == 0), (name, k)
            if a is not None and b is not None:
                assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max())), (name, k)


@pytest.mark.parametrize("n_rays", [24, 700])
def test_fused_loss_segments_and_centre_ball_selection(n_rays):
    """The inputs trainer.TrainStep hands the fused loss — separate supervision segments and the centre-ball selection made inside
    the kernels from (points, normals) — against the reference's formulation: boolean-mask compaction (functions.py:137-157), cat,
    VFLoss's means, torch autograd.  Values and every gradient, also with an upstream factor and with norm_smaller_than_one on."""
    gen = torch.Generator().manual_seed(n_rays)
    s_t = 20
    pts = (torch.rand(n_rays, s_t, 3, generator=gen) * 0.6 + torch.tensor([-0.3, -0.3, 0.25])).to(DEV)
    centroid, radius = (0.0, 0.0, 0.55), 0.15
    cfg = SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=5, directional_derivatives_start=100)
    w = SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0)
    base = {"rgb": torch.rand(n_rays, 3, generator=gen), "depth": torch.rand(n_rays, 1, generator=gen) * 1.5, "normals": torch.randn(n_rays, s_t, 3, generator=gen) * 0.8,
            "a": torch.randn(37, 3, generator=gen), "b": torch.randn(n_rays * 2, 3, generator=gen)}
    base["normals"][0, 0] = 0.0                                      # |n| = 0: gradient 0, like torch.linalg.vector_norm
    fixed = {k: torch.rand(v.shape, generator=gen).to(DEV) for k, v in base.items() if k != "normals"}
    for epoch in (0, 9):
        res = {}
        for fused in (True, False):
            crit = vloss.VFLoss(cfg, w)
            crit.fused = fused
            lv = {k: v.clone().to(DEV).requires_grad_(True) for k, v in base.items()}
            pred = {"rgb": lv["rgb"], "depth": lv["depth"], "normals": lv["normals"].reshape(-1, 3), "directional_derivatives": None}
            gt = {"rgb": fixed["rgb"], "depth": fixed["depth"]}
            if fused:
                pred["supervised_segments"] = [(lv["a"], fixed["a"]), (lv["b"], fixed["b"])]
                pred["supervised_normals"] = gt["supervised_normals"] = torch.empty(0, 3, device=DEV)
                pred["ray_center"] = (pts, centroid, radius)
            else:
                rc_n, rc_gt = supervision.get_center_indices_and_gt(pts, lv["normals"], torch.tensor(centroid, device=DEV), radius)
                assert 0 < rc_n.shape[0] < n_rays * s_t
                pred["supervised_normals"] = torch.cat([lv["a"], rc_n, lv["b"]])
                gt["supervised_normals"] = torch.cat([fixed["a"], rc_gt, fixed["b"]])
            loss, logs = crit(pred, gt, epoch)
            (loss * 0.6).backward()
            res[fused] = (float(loss), dict(logs), {k: v.grad.clone() for k, v in lv.items()})
        (lf, tf, gf), (lu, tu, gu) = res[True], res[False]
        assert abs(lf - lu) <= 2e-6 * max(1.0, abs(lu)), (epoch, lf, lu)
        for k in tf:
            assert abs(tf[k] - tu[k]) <= 2e-6 * max(1.0, abs(tu[k])), (epoch, k, tf[k], tu[k])
        assert (tf["norm_smaller_than_one_loss"] > 0) == (epoch >= 5)
        for k in gf:
            assert float((gf[k] - gu[k]).abs().max()) <= 2e-6 * max(1e-3, float(gu[k].abs().max())), (epoch, k)
        assert float(gf["normals"][0, 0].abs().max()) == 0.0


def test_train_step_is_the_same_step_with_and_without_the_fused_loss():
    """trainer.TrainStep with the fused loss (default), with the tensor-op loss on dense rows, and with the reference's compaction:
    the same loss terms at step 0 from the same state, and the same parameters after it (one clipped Adam update)."""
    from vf_nerf_amd import trainer
    fx, d = load_fixture("c1_perturb")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    gen = torch.Generator().manual_seed(1)
    rgb_gt, depth_gt = torch.rand(fx["n_rays"], 3, generator=gen).to(DEV), torch.rand(fx["n_rays"], 1, generator=gen).to(DEV)
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
    out = {}
    for mode in ("fused", "dense", "compact"):
        model = build_model(fx, d, device=DEV)
        model.step_sessions = False              # (launch by launch: no step session behind render() either)
        model.one_call_train_step = False        # the three LOSS formulations through the same (launch-by-launch) step; the one-call step
        supervision.manual_seed(5)               # has its own comparisons (test_one_call_*)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0, compact_selection=(mode == "compact"))
        step.criterion.fused = mode == "fused"
        loss, terms = step(g["pose"], g["uv"], g["intrinsics"], rgb_gt, depth_gt, epoch=0, uniforms=uni)
        out[mode] = (float(loss), dict(terms), float(step.last_total_norm),
                     model.vector_field_network.layers[5][0].weight.detach().clone(), model.rendering_network.layers[4].weight.detach().clone())
    for mode in ("dense", "compact"):
        assert abs(out[mode][0] - out["fused"][0]) < 1e-5 * max(1.0, abs(out["fused"][0])), mode
        for k, v in out["fused"][1].items():
            assert abs(out[mode][1][k] - v) < 1e-5 * max(1.0, abs(v)), (mode, k)
        assert abs(out[mode][2] - out["fused"][2]) < 1e-3 * out["fused"][2], (mode, out[mode][2], out["fused"][2])
        for a, b in zip(out[mode][3:], out["fused"][3:]):
            assert float((a - b).abs().max()) < 0.2 * 5e-4, mode                      # well inside one Adam update (lr 5e-4)
    assert out["fused"][1]["supervision_loss"] > 0



@pytest.mark.parametrize("colours", ["sparse", "dense"])
def test_training_step_is_the_same_step_on_one_stream_and_with_the_side_stream(colours):
    """vfn_train_step places part of its launches on an internal side stream (render.streams = 2, the default: the supervision forward
    behind the fine pass's forward, region 2's chain and weight gradients beside region 1's, the rendering net's re-packs; 3: the
    supervision forward beside the proposal pass as in round 4; 1: everything on the caller's stream).  Where a launch runs changes when it
    runs, never what it computes or in which order sums are taken: forward outputs bit-identical, and every parameter after three steps
    within rounding of the single-stream run (two kernels accumulate through atomicAdd)."""
    from vf_nerf_amd import trainer
    fx, d = load_fixture("c1_perturb")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    n = fx["n_rays"]
    gen = torch.Generator().manual_seed(1)
    rgb_gt, depth_gt = torch.rand(n, 3, generator=gen).to(DEV), torch.rand(n, 1, generator=gen).to(DEV)
    runs = {}
    for streams in (1, 2, 3):
        model = build_model(fx, d, device=DEV)
        model.train_step_streams = streams
        model.sparse_colour_training = colours == "sparse"
        model.rng_seed, model._rng_offset = 5, 0
        supervision.manual_seed(9)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
        rec = []
        for t in range(3):
            loss, terms = step(g["pose"], g["uv"], g["intrinsics"], rgb_gt, depth_gt, epoch=0)
            assert step.one_call.why_not is None, step.one_call.why_not
            o = step.last_outputs
            rec.append(dict(loss=float(loss), z=o.z_vals.detach().clone(), rgb=o.coarse_rgb_values.detach().clone(),
                            normals=o.coarse_normals.detach().clone()))
        runs[streams] = (rec, [p.detach().clone() for p in model.unique_parameters()])
    lr = float(build_model(fx, d, device=DEV).optimizer.param_groups[0]["lr"])
    for streams in (2, 3):
        for field in ("z", "rgb", "normals"):
            assert torch.equal(runs[streams][0][0][field], runs[1][0][0][field]), (streams, field)      # step 0: identical state, deterministic forward
        assert runs[streams][0][0]["loss"] == runs[1][0][0]["loss"]
        for t in (1, 2):
            assert abs(runs[streams][0][t]["loss"] - runs[1][0][t]["loss"]) <= 1e-4 * max(1.0, abs(runs[1][0][t]["loss"]))
        worst = max(float((a - b).abs().max()) for a, b in zip(runs[streams][1], runs[1][1]))
        print(f"[{colours}] streams {streams} vs 1: parameters after three steps differ by at most {worst / lr:.3f} lr")
        assert worst <= 0.5 * lr

@pytest.mark.parametrize("mode", ["default", "dense_colours", "fp32_storages", "single_product", "replayed_draws"])
def test_one_call_training_step_equals_the_launch_by_launch_step(mode):
    """vfn_train_step (csrc/vfn_train.hip, vf_nerf_amd/onecall.py) issues the launches of trainer.TrainStep's Python path from C out
    of one workspace.  Two models with the same weights, batches and random streams, one per path, three steps:

    * step 0 starts from identical state and its forward is deterministic: loss, the six terms, sampled depths, rgb / depth /
      normals are BIT-identical, and so are the colours wherever the C call evaluates them (by default only where a sample's weight
      is non-zero — the exact sparse colour branch of include/vfn.h; "dense_colours" switches it off);
    * gradients leave two kernels through atomicAdd (the density scalars, the loss reductions), so from there on the two runs agree
      to rounding, not bitwise: clip norm within 1e-5, parameters after each step within 2 % of one Adam update (lr) wherever the
      update is not a coin flip, losses of steps 1-2 within 1e-4;
    * the bookkeeping is the same: Adam step counters (2 per step on the aliased vector-field parameters, Q4), learning rate,
      render / supervision random-stream positions, and the weight packs are current after the call (a gradient-free render right
      after the steps returns the same image on both models)."""
    from vf_nerf_amd import trainer
    fx, d = load_fixture("c1_perturb")           # 48 rays x (32 + 32) = 3072 samples: whole groups of 32
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    n = fx["n_rays"]
    gen = torch.Generator().manual_seed(1)
    rgb_gt, depth_gt = torch.rand(n, 3, generator=gen).to(DEV), torch.rand(n, 1, generator=gen).to(DEV)
    n_sup = (n * 64) // 10
    runs = {}
    for path in ("one_call", "python"):
        model = build_model(fx, d, device=DEV)
        if mode == "fp32_storages":
            model.activation_storage = model.gradient_storage = "fp32"
        elif mode == "single_product":
            model.training_products = 1
        model.one_call_train_step = path == "one_call"
        model.step_sessions = path == "one_call"                    # (the launch-by-launch leg: no step session behind render() either)
        model.sparse_colour_training = mode != "dense_colours"      # (the C call's default: the colour branch only where w > 0; exact)
        model.rng_seed, model._rng_offset = 5, 0
        supervision.manual_seed(9)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
        rec = []
        ugen = torch.Generator().manual_seed(77)
        for t in range(3):
            uni = None
            if mode == "replayed_draws":
                uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
                supervision.replay_uniforms(torch.rand(n_sup, 3, generator=ugen).to(DEV), torch.rand(n_sup, 3, generator=ugen).to(DEV))
            loss, terms = step(g["pose"], g["uv"], g["intrinsics"], rgb_gt, depth_gt, epoch=0, uniforms=uni)
            o = step.last_outputs
            rec.append(dict(loss=float(loss), terms=dict(terms), norm=float(step.last_total_norm), z=o.z_vals.detach().clone(),
                            rgb=o.coarse_rgb_values.detach().clone(), depth=o.coarse_depth_map.detach().clone(),
                            normals=o.coarse_normals.detach().clone(), colors=o.coarse_colors.detach().clone(),
                            params=[p.detach().clone() for p in model.unique_parameters()]))
        took = step.one_call.why_not is None
        assert took == (path == "one_call"), (path, step.one_call.why_not)
        with torch.no_grad():
            img = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add")})
        runs[path] = dict(rec=rec, img=img.coarse_rgb_values.clone(), rng=(model._rng_offset, supervision._offset),
                          lr=model.optimizer.param_groups[0]["lr"],
                          steps=(float(model.optimizer.state[model.vector_field_network.layers[8].weight]["step"]),
                                 float(model.optimizer.state[model.rendering_network.layers[4].weight]["step"])))
    a, b = runs["one_call"], runs["python"]
    lr = 5e-4
    for k in ("z", "rgb", "depth", "normals"):
        assert torch.equal(a["rec"][0][k], b["rec"][0][k]), f"step 0 {k}"
    # colours: the sparse colour branch evaluates them only where the sample's weight is non-zero (zeros elsewhere: they multiply a zero
    # weight); where it does, they are the dense path's bit for bit — and rgb = sum w c above IS bit-identical
    ca, cb = a["rec"][0]["colors"], b["rec"][0]["colors"]
    evaluated = (ca != 0).any(dim=1)
    assert torch.equal(ca[evaluated], cb[evaluated]) and (mode == "dense_colours") == bool(evaluated.all())
    print(f"[{mode}] colour branch evaluated on {float(evaluated.float().mean()):.3f} of the samples")
    assert a["rec"][0]["loss"] == b["rec"][0]["loss"] and a["rec"][0]["terms"] == b["rec"][0]["terms"]
    for t in range(3):
        ra, rb = a["rec"][t], b["rec"][t]
        worst = max(float((x - y).abs().max()) for x, y in zip(ra["params"], rb["params"]))
        frac = sum(int(((x - y).abs() > 0.02 * lr).sum()) for x, y in zip(ra["params"], rb["params"])) / sum(x.numel() for x in ra["params"])
        print(f"[{mode}] step {t}: loss {ra['loss']:.6f} / {rb['loss']:.6f}; clip norm {ra['norm']:.6f} / {rb['norm']:.6f}; parameters differ by at most "
              f"{worst / lr:.3f} lr, {frac:.2e} of them by more than 0.02 lr")
        # "dense_colours" runs the Python path's launches one for one: agreement to the order of a few atomic additions.  The sparse colour
        # branch (the default) splits every sample's upstream gradient between two chain passes (d normals on region 1, d colours on
        # region 2), each rounded to the 16-bit gradient storage / the bf16x3 operands on its own instead of after their fp32 sum: the
        # parameter gradients agree to the storages' own precision (2.6e-5 in the clip norm observed), not to the last bit
        tight = mode == "dense_colours"
        assert abs(ra["norm"] - rb["norm"]) < (1e-5 if tight else 2e-4) * rb["norm"] * (1 if t == 0 else 100)
        assert abs(ra["loss"] - rb["loss"]) < (1e-4 if (tight or t == 0) else 1e-3) * max(1.0, abs(rb["loss"]))
        if mode == "single_product" and t > 0:
            # one product per K-block multiplies the weights' f16 ROUNDINGS: a weight that differs by 1e-6 between the two runs (the
            # atomics of step 0) can round to the neighbouring f16 value, 6e-5 away — the single-product arithmetic is discontinuous in
            # the weights, so from the second step on the runs are two trajectories of the same chaotic map (observed: 5 % of the
            # parameters more than 0.02 lr apart after step 2, none more than 3 lr).  Bounded, not pinned.
            assert frac < 0.25 and worst <= 4.1 * lr * 2
        else:
            # (a coin-flip sign is at most two updates of lr apart; the fraction of coin flips grows with every step as the two runs —
            # identical up to the order of a few atomic additions — drift apart: observed 0 / 5e-4 / 9e-3 on steps 0 / 1 / 2)
            assert frac < ((1e-5, 5e-3, 5e-2) if tight else (6e-3, 3e-2, 1.5e-1))[t] and worst <= 2.05 * lr * 2
    assert a["rng"] == b["rng"] and a["lr"] == b["lr"] and a["steps"] == b["steps"] == (6.0, 3.0)
    assert float((a["img"] - b["img"]).abs().max()) < 5e-3, "renders right after the steps: the re-packed weights are the updated ones"


def test_sparse_colour_branch_gives_the_dense_gradients():
    """vfn_train_step's sparse colour branch (include/vfn.h: the rendering net and the feature block evaluated, differentiated and summed
    into the weight gradients only for the samples with w > 0) against the dense step, on the GRADIENTS themselves: phase 1 of the call
    (forward + backward) with a stand-in for the gradient bucket that snapshots the flat gradient where a multi-rank run would all-reduce
    it.  512 rays x (64 + 64) + supervision, default 16-bit storages and fp32 storages: every parameter's gradient within the storage's own
    bound of the dense one-call step's and of the Python path's (1e-3 of the tensor's largest entry with the 16-bit storages, the bound
    test_render_gradients holds the dense path to against the oracle; 2e-4 with fp32 storages), the density scalars included; the loss
    and the outputs other than the un-evaluated colours bit-identical."""
    from vf_nerf_amd import trainer
    import bench
    dev = torch.device(DEV)

    class Snapshot:                                  # what trainer.TrainStep asks of a bucket
        def __init__(self, model):
            self.model, self.grads = model, None

        def zero(self):
            self.model.optimizer.zero_grad()

        def all_reduce_mean(self):
            self.grads = [p.grad.detach().clone() for p in self.model.unique_parameters()]

    worst_all = {}
    for storages in ("f16", "fp32"):
        got = {}
        for tag, one_call, sparse in (("sparse", True, True), ("dense", True, False), ("python", False, False)):
            model, uv, pose, K = bench.build_scene(dev, 512, 64, 64, seed=0)
            model.activation_storage = model.gradient_storage = storages
            model.one_call_train_step, model.sparse_colour_training = one_call, sparse
            model.step_sessions = one_call
            model.rng_seed, model._rng_offset = 3, 0
            supervision.manual_seed(21)
            gen = torch.Generator().manual_seed(2)
            rgb_gt, depth_gt = torch.rand(512, 3, generator=gen).to(dev), (0.2 + 0.6 * torch.rand(512, 1, generator=gen)).to(dev)
            snap = Snapshot(model)
            step = trainer.TrainStep(model, (0.0, 0.0, 0.6), border_radius=0.05, far=1.0, bucket=snap)
            loss, _ = step(pose, uv, K, rgb_gt, depth_gt, epoch=0)
            assert (step.one_call.why_not is None) == one_call, step.one_call.why_not
            got[tag] = (float(loss), snap.grads, step.last_outputs.coarse_rgb_values.clone(), step.last_outputs.coarse_depth_map.clone())
        names = [f"{net}.{k}" for net, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density))
                 for k, _ in mod.named_parameters()]
        assert got["sparse"][0] == got["dense"][0] == got["python"][0]
        assert torch.equal(got["sparse"][2], got["python"][2]) and torch.equal(got["sparse"][3], got["python"][3])
        tol = 1e-3 if storages == "f16" else 2e-4
        for a, b in (("sparse", "dense"), ("sparse", "python"), ("dense", "python")):
            worst = max((float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)), nm) for x, y, nm in zip(got[a][1], got[b][1], names))
            worst_all[storages, a, b] = worst
            print(f"[{storages} storages] {a} vs {b}: worst parameter-gradient difference {worst[0]:.2e} of the tensor's largest entry ({worst[1]})")
            assert worst[0] < (tol if "sparse" in (a, b) else 1e-5), (storages, a, b, worst)


@pytest.mark.parametrize("case", ["no_sample_selected", "oscillating_field", "ragged_selection_count"])
def test_sparse_colour_branch_edge_cases(case):
    """The device-side selection of vfn_train_step at its edges, against the dense Python path on the same weights, rays and draws:
    * no sample carries weight (a vector head that never flips: density identically zero, K = 0 — every launch of region 2 leaves at
      once, the colour branch's gradients are exactly zero);
    * a vector head twelve times as steep (an oscillating, saturated field: other rays and other samples carry the weight);
    * a selection count that is not a multiple of 32 / 128 (partial last group and workgroup of region 2) — true of any real batch, made
      explicit here by checking the count.
    Forward outputs bit-identical, parameter gradients within 1e-3 of a tensor's largest entry (the bound of the default storages)."""
    from vf_nerf_amd import trainer
    fx, d = load_fixture("c1_perturb")
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    n = fx["n_rays"]
    gen = torch.Generator().manual_seed(1)
    rgb_gt, depth_gt = torch.rand(n, 3, generator=gen).to(DEV), torch.rand(n, 1, generator=gen).to(DEV)
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}

    class Snapshot:
        def __init__(self, model):
            self.model, self.grads = model, None

        def zero(self):
            self.model.optimizer.zero_grad()

        def all_reduce_mean(self):
            self.grads = [p.grad.detach().clone() for p in self.model.unique_parameters()]

    got = {}
    for path in ("one_call", "python"):
        model = build_model(fx, d, device=DEV)
        head = model.vector_field_network.layers[8]
        with torch.no_grad():
            if case == "no_sample_selected":
                head.weight[:3].zero_()
                head.bias[:3] = torch.tensor([0.3, -0.2, 0.9])
            elif case == "oscillating_field":
                head.weight[:3].mul_(12.0)
        model.vector_field_network._invalidate_packs()
        model.one_call_train_step = path == "one_call"
        model.step_sessions = path == "one_call"
        supervision.manual_seed(4)
        snap = Snapshot(model)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0, bucket=snap)
        loss, _ = step(g["pose"], g["uv"], g["intrinsics"], rgb_gt, depth_gt, epoch=0, uniforms=uni)
        assert (step.one_call.why_not is None) == (path == "one_call"), step.one_call.why_not
        o = step.last_outputs
        counts = step.last_colour_counts.tolist() if path == "one_call" else None
        got[path] = (float(loss), snap.grads, o.coarse_rgb_values.clone(), o.coarse_depth_map.clone(), o.coarse_normals.clone(), counts)
    a, b = got["one_call"], got["python"]
    k, m = int(a[5][0]), int(a[5][1])
    print(f"[{case}] selected {k} of {m} samples ({k / m:.3f}); loss {a[0]:.6f} / {b[0]:.6f}")
    assert a[0] == b[0] and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])
    assert all(bool(torch.isfinite(x).all()) for x in a[1])
    worst = max(float((x - y).abs().max() / y.abs().max().clamp_min(1e-30)) for x, y in zip(a[1], b[1]) if float(y.abs().max()) > 0)
    print(f"[{case}] worst parameter-gradient difference to the dense Python path: {worst:.2e} of the tensor's largest entry")
    assert worst < 1e-3
    if case == "no_sample_selected":
        assert k == 0 and float(a[2].abs().max()) == 0.0
        rn_grads = a[1][len(list(build_model(fx, d).vector_field_network.parameters())):-3]
        assert all(float(x.abs().max()) == 0.0 for x in rn_grads), "no colour was used: the rendering net gets no gradient at all"
    elif case == "oscillating_field":
        assert 0 < k < m
    else:
        assert k % 32 != 0 and 0 < k < m


def test_lazy_loss_terms_survive_the_pinned_ring_wrapping_around():
    """The six log scalars travel through a ring of 64 pinned buffers and are fetched on first read: a dict that is still unread
    when its buffer comes round again is read before the buffer is handed on, so late readers get their own step's values."""
    crit = vloss.VFLoss(SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                        SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0))
    gen = torch.Generator().manual_seed(0)
    gt = {"rgb": torch.rand(8, 3, generator=gen).to(DEV), "depth": torch.rand(8, 1, generator=gen).to(DEV), "supervised_normals": torch.empty(0, device=DEV)}
    kept, want = [], []
    for i in range(150):
        pred = {"rgb": torch.full((8, 3), 0.01 * i, device=DEV), "depth": torch.rand(8, 1, generator=gen).to(DEV),
                "normals": torch.randn(16, 3, generator=gen).to(DEV), "supervised_normals": torch.empty(0, 3, device=DEV), "directional_derivatives": None}
        _, terms = crit(pred, gt, 0)
        kept.append(terms)
        want.append(float((pred["rgb"] - gt["rgb"]).abs().mean()))
    for i in (0, 1, 63, 64, 65, 100, 149):
        assert abs(kept[i]["rgb_loss"] - want[i]) < 1e-6, i
