"""The step session (vf_nerf_amd/stepengine.py): a grad-mode render() of the shipped regime and the calls the reference trainer makes
after it (train/vector_field_nerf_train.py:186-260) on the training step's workspace — against the launch-by-launch autograd path
(model.step_sessions = False, backward.py), which the other test files pin against the oracle and the reference's captured gradients.
The replay of the reference trainer's recorded steps through the session is tests/test_hip_trainer.py::
test_reference_call_sequence_replays_the_reference_trainer_steps."""
import pytest
import torch

from helpers import build_model, load_fixture, loss_coefficients
from vf_nerf_amd import lib, stepengine, supervision

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CENTROID = (0.0, 0.0, 0.55)


def _grads(model):
    return {f"{tag}.{k}": p.grad.detach().clone() for tag, net in (("vf", model.vector_field_network), ("rn", model.rendering_network),
                                                                   ("density", model.density)) for k, p in net.named_parameters()}


def _worst(a, b):
    worst = 0.0
    for k in b:
        scale = max(float(b[k].abs().max()), 1e-30)
        worst = max(worst, float((a[k] - b[k]).abs().max()) / scale)
    return worst


def _model_and_batch(name="bench_sizes", sessions=True):
    fx, d = load_fixture(name)
    model = build_model(fx, d, device=DEV)
    model.step_sessions = sessions
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    return fx, d, model, g, uni


@pytest.mark.parametrize("case", ["render_only", "with_supervision", "supervision_only", "full_matrix_of_a_forward", "one_batch_sampled_never_forwarded"])
def test_session_gradients_equal_the_launch_by_launch_path(case):
    """Same state, same draws, same loss through both paths: forward values bit-identical (rgb, depth, normals, sampled depths), every
    parameter gradient within 2e-3 of the tensor's largest entry (the sparse colour branch and the one-chain backward change the ORDER of
    the sums; the 16-bit storages' own bound against exact gradients is 1e-3).
      render_only                       no supervision call at all: the reserved rows stay unwritten and must not be walked
      with_supervision                  the trainer's two batches through vector_field_network(points)[:, :3]
      supervision_only                  a loss that never reaches the render's outputs: the parked rows are differentiated on their own
      full_matrix_of_a_forward          the [n, 3 + F] result used as a whole (features included): the lazy result materialises
      one_batch_sampled_never_forwarded a sampled region without a forward below a forwarded one"""
    got = {}
    for sessions in (True, False):
        fx, d, model, g, uni = _model_and_batch("bench_sizes", sessions)
        n, s_t = d["z_vals"].shape
        a, b, c = (t.to(DEV) for t in loss_coefficients(n, s_t))
        supervision.manual_seed(5)
        model.optimizer.zero_grad()
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
        eng = stepengine.StepEngine.of(model)
        assert (eng.session is not None and eng.why_not is None) == sessions, eng.why_not
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
        n_sup = (n * s_t) // 10
        cen = torch.tensor(CENTROID)                 # a host tensor: its values are known without a read-back
        vf = model.vector_field_network
        if case in ("with_supervision", "supervision_only", "full_matrix_of_a_forward", "one_batch_sampled_never_forwarded"):
            bp, bgt = supervision.sample_border_points(0.25, 1.0, n_sup, cen, DEV)
            cp, cgt = supervision.sample_center_points(cen, 0.15, n_sup, DEV)
            if sessions:
                assert len(eng.session.regions) == 2 and bp.data_ptr() == eng.session.sup_pts.data_ptr()
            if case == "full_matrix_of_a_forward":
                full = vf(bp)
                assert tuple(full.shape) == (n_sup, 3 + 256) and (type(full) is not torch.Tensor) == sessions
                sup = ((full[:, :3] - bgt) ** 2).mean() + 1e-3 * (full[:, 3:] ** 2).mean() + ((vf(cp)[:, :3] - cgt) ** 2).mean()
            elif case == "one_batch_sampled_never_forwarded":
                sup = ((vf(cp)[:, :3] - cgt) ** 2).mean()
            else:
                sup = ((vf(bp)[:, :3] - bgt) ** 2).mean() + ((vf(cp)[:, :3] - cgt) ** 2).mean()
            loss = sup if case == "supervision_only" else loss + sup
        loss.backward()
        if sessions and case != "supervision_only":
            assert eng.session.backward_done and not any(r["pending"] for r in eng.session.regions.values())
        got[sessions] = (float(loss), out, _grads(model))
    (l1, o1, g1), (l0, o0, g0) = got[True], got[False]
    if case != "supervision_only":
        for f in ("z_vals", "coarse_rgb_values", "coarse_depth_map", "coarse_normals", "points_coarse"):
            assert torch.equal(getattr(o1, f), getattr(o0, f)), f
    assert abs(l1 - l0) <= 1e-5 * max(1.0, abs(l0))
    worst = _worst(g1, g0)
    print(f"{case}: loss {l1:.6f} / {l0:.6f}; worst gradient difference {worst:.2e} of the tensor's largest entry")
    assert all(torch.isfinite(v).all() for v in g1.values()) and worst < 2e-3


def test_a_second_render_before_the_backward_takes_the_launch_by_launch_path():
    """One open step per model: while a render()'s outputs are alive and not yet differentiated, another grad-mode render() must not
    reuse the workspace — it takes the launch-by-launch path (and says why); both graphs then differentiate correctly in one backward."""
    fx, d, model, g, uni = _model_and_batch("bench_sizes", True)
    n, s_t = d["z_vals"].shape
    a, b, c = (t.to(DEV) for t in loss_coefficients(n, s_t))
    eng = stepengine.StepEngine.of(model)
    model.optimizer.zero_grad()
    out1 = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    first = eng.session
    assert first is not None and first.open
    out2 = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    assert eng.session is first and "still waiting" in eng.why_not
    assert torch.equal(out1.coarse_rgb_values, out2.coarse_rgb_values)
    ((out1.coarse_rgb_values * a).sum() + (out2.coarse_rgb_values * a).sum() + (out2.coarse_normals * c).sum()).backward()
    both = _grads(model)
    ref = _model_and_batch("bench_sizes", False)[2]
    ref.optimizer.zero_grad()
    r1 = ref.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    r2 = ref.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    ((r1.coarse_rgb_values * a).sum() + (r2.coarse_rgb_values * a).sum() + (r2.coarse_normals * c).sum()).backward()
    assert _worst(both, _grads(ref)) < 2e-3
    # the first step is differentiated: the next render opens a session again
    out3 = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    assert eng.session is not first and eng.why_not is None
    del out3


def test_deferred_centre_rows_need_the_drop_in_loss():
    """functions.get_center_indices_and_gt on an open step defers the centre-ball rows to loss.VFLoss (no boolean-mask indexing, no
    synchronisation).  A loss that does not know about them would silently train without that term: the step's backward refuses instead;
    model.defer_center_rows = False gives the compacted rows back."""
    fx, d, model, g, uni = _model_and_batch("bench_sizes", True)
    cen = torch.tensor(CENTROID)
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    rows, gt = supervision.get_center_indices_and_gt(out.points_coarse, out.coarse_normals, cen, 0.15)
    assert rows.shape == (0, 3) and gt.shape == (0, 3) and stepengine.find_marker(torch.cat([torch.empty(0, 3, device=DEV), rows])) is not None
    loss = out.coarse_rgb_values.abs().mean() + ((rows - gt) ** 2).sum()
    with pytest.raises(RuntimeError, match="deferred the centre-ball rows"):
        loss.backward()
    model2 = _model_and_batch("bench_sizes", True)[2]
    model2.defer_center_rows = False
    out = model2.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    rows, gt = supervision.get_center_indices_and_gt(out.points_coarse, out.coarse_normals, cen.to(DEV), 0.15)
    keep = torch.linalg.vector_norm(out.points_coarse - cen.to(DEV), dim=2) < 0.15
    assert rows.shape[0] == int(keep.sum()) and rows.shape == gt.shape
    (out.coarse_rgb_values.abs().mean() + ((rows - gt) ** 2).mean()).backward()


def test_batch_tensors_are_checked_before_their_pointers_are_used():
    """the step's C calls read the batch through raw pointers.  Host-resident pixels / intrinsics are moved to the device, a
    pose batch of the wrong row count raises — neither reaches a kernel as a bad pointer."""
    fx, d, model, g, uni = _model_and_batch("bench_sizes", True)
    out = model.render(g["pose"], g["uv"].cpu(), g["intrinsics"].cpu(), epoch=0, uniforms=uni)       # moved, not dereferenced on the host
    ref = _model_and_batch("bench_sizes", True)[2].render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    assert torch.equal(out.coarse_rgb_values, ref.coarse_rgb_values)
    del out, ref
    model = _model_and_batch("bench_sizes", True)[2]
    with pytest.raises(lib.VfnError, match="rows"):
        model.render(g["pose"][:5], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    with pytest.raises(lib.VfnError, match="shape"):
        model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={k: v[:3] for k, v in uni.items()})


def test_session_render_returns_every_samples_colour_like_the_reference():
    """models/nerf/vector_field_nerf.py:315-338: ``coarse_colors`` is the rendering net's colour of EVERY sample.  A step session evaluates
    the colour branch on the SELECTED samples only (the sparse colour branch) — the field is completed on first access by a dense
    gradient-free launch (render_output.LazyColours), so a reader sees the reference's tensor: against the reference's own captured
    colours (tests/golden/bench_sizes.npz) every row is inside 2e-5, selected or not; the rows the step computed are the step's values bit
    for bit (the zeros elsewhere are what the C call left); reading the field does not disturb the step (gradients equal to a step whose
    colours were never read); and after optimizer.step() — new weights — a first read raises instead of returning other weights' colours."""
    fx, d, model, g, uni = _model_and_batch("bench_sizes", True)
    n, s_t = d["z_vals"].shape
    a, b, c = (t.to(DEV) for t in loss_coefficients(n, s_t))
    model.optimizer.zero_grad()
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    eng = stepengine.StepEngine.of(model)
    assert eng.session is not None and eng.why_not is None and eng.params.sparse_colours == 1
    raw = eng.session.colors.view(n * s_t, 3).clone()                       # what the C call wrote: selected rows, zeros elsewhere
    w = eng.session.weights.view(n, s_t)
    ran = (raw != 0).any(dim=1)
    positive = (w > 0).reshape(-1)
    assert bool((ran | ~positive).all()), "a sample with w > 0 was not selected"
    assert 0 < int(ran.sum()) < n * s_t // 2, "the selection is neither empty nor dense on this fixture"
    colours = out.coarse_colors                                              # first access: the dense fill
    assert type(colours) is torch.Tensor and tuple(colours.shape) == (n * s_t, 3)
    assert torch.equal(colours[ran], raw[ran])
    err = (colours.cpu() - d["colors"]).abs().max(dim=1)[0]
    print(f"session coarse_colors vs the reference's: selected rows {int(ran.sum())} of {n * s_t}, worst error on selected rows "
          f"{float(err[ran.cpu()].max()):.2e}, on the filled rows {float(err[~ran.cpu()].max()):.2e}")
    assert float(err.max()) < 2e-5
    assert out.coarse_colors is colours                                      # filled once
    ((out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()).backward()
    got = _grads(model)
    fx, d, ref, g, uni = _model_and_batch("bench_sizes", True)
    ref.optimizer.zero_grad()
    o2 = ref.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    ((o2.coarse_rgb_values * a).sum() + (o2.coarse_depth_map * b).sum() + (o2.coarse_normals * c).sum()).backward()
    # (equal up to the order of the atomic sums of two runs of the same step)
    assert _worst(got, _grads(ref)) < 1e-4
    torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
    ref.optimizer.step()
    with pytest.raises(RuntimeError, match="before optimizer.step"):
        o2.coarse_colors
    # eager fill: the same tensor, made inside render()
    fx, d, eager, g, uni = _model_and_batch("bench_sizes", True)
    eager.eager_session_colours = True
    o3 = eager.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    assert torch.equal(o3.coarse_colors, colours)
    del o3


def test_training_selection_keeps_samples_whose_weight_underflowed_but_whose_derivative_did_not():
    """The sparse colour branch's selection (csrc/vfn_rays.hip, sel_predicate): w > 0, or w = 0 by an underflowed alpha alone (sigma > 0,
    delta > 0, T > 0) — there d w / d sigma = T delta is not zero and the dense step's (d rgb . c) term needs the colour.  Constructed
    directly: sigma tiny (sigma delta < 6e-8 rounds alpha to 0) on otherwise empty rays."""
    n, s = 8, 64
    z = torch.linspace(0.0, 1.0, s, device=DEV).repeat(n, 1).contiguous()
    sigma = torch.zeros(n, s, device=DEV)
    sigma[0, 10] = 1e-6                # alpha = 1 - exp(-1e-6 / 63) == 0 in fp32, T = 1
    sigma[1, 5] = 1e4                  # opaque: everything behind it has T = 0
    sigma[1, 20] = 1e-6                # ... so this one is NOT needed
    sigma[2, 7] = 3.0                  # an ordinary surface sample: w > 0
    delta = torch.cat([z[:, 1:] - z[:, :-1], torch.full((n, 1), 1e10, device=DEV)], 1)
    e = sigma * delta
    T = torch.exp(-(torch.cumsum(e.double(), 1) - e.double())).float()
    w = (1.0 - torch.exp(-e)) * T
    assert float(w[0, 10]) == 0.0 and float(w[2, 7]) > 0.0
    pts = torch.rand(n, s, 3, device=DEV)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, device=DEV), dim=1)
    got_train = lib.select_samples(w, pts, dirs, sigma=sigma, z_vals=z)
    got_fwd = lib.select_samples(w, pts, dirs)
    want_fwd = sorted(int(i) for i in torch.nonzero(w.reshape(-1) > 0).reshape(-1))
    assert got_fwd == want_fwd and 2 * s + 7 in got_fwd and 0 * s + 10 not in got_fwd
    assert 0 * s + 10 in got_train and 1 * s + 20 not in got_train and set(got_fwd) <= set(got_train)
    assert got_train == sorted(got_train)


def test_clip_inside_optimizer_step_equals_the_wrapped_clip():
    """``dropin.install(patch_clip=False)``: torch.nn.utils.clip_grad_norm_ is PyTorch's own function, a step session
    parks its gradient (every param.grad None until the step), the trainer's clip call scales nothing, and optimizer.step() clips with the
    configured norm before the update — the same parameters, bit for bit, as the default drop-in (wrapped clip, then step), SURVEY Q4's
    double clip and double Adam update of the aliased vector-field parameters included."""
    from vf_nerf_amd import dropin, optim
    after = {}
    for patched in (True, False):
        try:
            dropin.install(patch_clip=patched)
            fx, d, model, g, uni = _model_and_batch("bench_sizes", True)
            n, s_t = d["z_vals"].shape
            a, b, c = (t.to(DEV) for t in loss_coefficients(n, s_t))
            clip = float(model.config.scheduler_config.clip_norm)
            for _ in range(2):                          # two steps: the second starts from re-bound gradient views
                model.optimizer.zero_grad()
                out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
                assert stepengine.StepEngine.of(model).why_not is None
                ((out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()).backward()
                if not patched:
                    assert all(p.grad is None for p in model.unique_parameters()) and "parked_max_norm" in model.optimizer.flat()
                norm = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
                if not patched:
                    assert float(norm) == 0.0          # PyTorch's function saw no gradient
                model.optimizer.step()
                model.scheduler.step()
                assert all(p.grad is not None for p in model.unique_parameters()) and "parked_max_norm" not in model.optimizer.flat()
            after[patched] = {k: p.detach().clone() for net in (model.vector_field_network, model.rendering_network, model.density)
                              for k, p in net.named_parameters(prefix=type(net).__name__)}
            step8 = float(model.optimizer.state[model.vector_field_network.layers[8].weight]["step"])
            assert step8 == 4.0                         # Q4: two updates per step for the aliased net
        finally:
            dropin.install()
    assert optim.CLIP_INSIDE_STEP is False
    for k, v in after[True].items():
        assert torch.equal(v, after[False][k]), k
