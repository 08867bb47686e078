"""Training path on the GPU: gradients of a fixed linear functional of (rgb, depth, normals) from the HIP backward
kernels against the CPU oracle's autograd and against gradients captured from the reference's own backward pass.
Tolerances are PER MODE, at about three times what each arithmetic is observed to deliver (``MODE_TOL``; sums over
thousands of points in a different order); observed errors are printed."""
import pytest
import torch

from helpers import (FIXTURE_NAMES, GRAD_KEYS, build_model, grad_rel_err, load_fixture, loss_coefficients,
                     oracle_gradients)

from vf_nerf_amd import lib

pytestmark = pytest.mark.gpu
TOL = 1e-3
# Worst parameter-gradient error (of the tensor's largest entry) against the oracle with this implementation's ReLU masks pinned, per
# arithmetic; observed over the eight fixtures on MI355X (round 4, gpurun_out/r04/grad_errs.txt) in brackets.  A 5x regression of
# any mode fails its line; the blanket 1e-3 of rounds 1-3 would have let the exact modes slip by an order of magnitude.
MODE_TOL = {
    "fp32": 1e-4,                      # exact fp32 MFMA forward + backward                              [1.5e-6 .. 2.8e-5]
    "f16x3": 2e-4,                     # split-f16 forward, bf16x3 chain, fp32 storages                  [6.7e-5 .. 7.7e-5]
    "f16x3+rows": 2e-4,                #   ... row-major workspace                                       [6.0e-5 .. 7.2e-5]
    "f16x3+f16dy": 1e-3,               # scaled-f16 gradient storage                                     [2.3e-4 .. 7.9e-4]
    "f16x3+f16act": 1e-3,              # f16 activation storage                                          [2.0e-4 .. 7.3e-4]
    "f16x3+f16act+rows": 1e-3,         #                                                                 [2.0e-4 .. 7.1e-4]
    "f16x3+f16act+f16dy": 1e-3,        # the DEFAULT 16-bit storages                                     [2.8e-4 .. 7.4e-4]
    "f16x3+f16act+f16dy+c2": 1e-3,     #   ... with the saving forward's colour branch on two products   [3.7e-4 .. 7.3e-4]
    "f16x3+f16act+bf16dy": 1e-2,       # bf16 gradient storage (opt-in, 8 significant bits)              [1.7e-3 .. 4.5e-3]
}
# the 64 + 64-sample fixture: two fp32 evaluations of this step already differ by 8e-4 (exact-fp32 kernels against the CPU oracle:
# the density amplifies the normals' rounding into the weights, and the gradient follows) — the FLOOR of every mode there
# [fp32 8.2e-4, f16x3 8.0e-4, 16-bit storages 9.3e-4 .. 1.1e-3, bf16 3.6e-3]
BENCH_SIZES_TOL = {"exact": 1.5e-3, "stored16": 2e-3}


def _hip_gradients(fx, d, model):
    g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    for p in model.unique_parameters():
        p.grad = None
    model._keep_saved = True
    model._debug_dst = None
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    a, b, c = (t.to("cuda:0") for t in loss_coefficients(*d["z_vals"].shape))
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    return float(loss), out


@pytest.mark.parametrize("precision", ["f16x3", "fp32", "f16x3+f16act", "f16x3+f16act+bf16dy", "f16x3+f16act+f16dy", "f16x3+f16dy", "f16x3+rows", "f16x3+f16act+rows",
                                       "f16x3+f16act+f16dy+c2"])
@pytest.mark.parametrize("name", FIXTURE_NAMES)
def test_render_gradients(name, precision):
    """``precision`` selects the arithmetic of the activation-saving forward (split-half f16 products or exact fp32
    MFMA); "+f16act" stores the hidden activations as f16 for the weight-gradient kernels (``model.activation_storage``, the
    default), "+bf16dy" the pre-activation gradients as bf16 (``model.gradient_storage``), "+rows" keeps the row-major
    workspace and its kernels instead of the fragment-ordered one (``model.workspace_layout``), "+f16dy" the scaled f16 gradient
    storage (the default), "+c2" the saving forward with its colour branch on two products (``model.training_colour_products``,
    opt-in): the same 1e-3 bound must hold for all of them, the observed error is printed."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device="cuda:0")
    opts = precision.split("+")
    f16act = "f16act" in opts
    model.precision = opts[0]
    model.activation_storage = "f16" if f16act else "fp32"
    model.gradient_storage = "bf16" if "bf16dy" in opts else ("f16" if "f16dy" in opts else "fp32")
    model.workspace_layout = "rows" if "rows" in opts else "fragment"
    model.training_colour_products = 2 if "c2" in opts else 3      # (opt-in: the saving forward's colour branch on two products)
    loss, out = _hip_gradients(fx, d, model)
    assert (out.z_vals.cpu() == d["z_vals"]).all(), "sampling must replay exactly for the comparison to be meaningful"
    ref_loss, ref = oracle_gradients(fx, d, build_model(fx, d))
    assert abs(loss - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    # An assertion that does NOT depend on this implementation's ReLU masks: the whole parameter gradient against the oracle's own
    # (unpinned) autograd, per network, as a direction and a length.  Looser than the per-tensor bounds below — a unit on the other
    # side of zero moves a layer's gradient by percents on these small fixtures — but independent of anything the kernels report.
    for tag, net in (("vf", model.vector_field_network), ("rn", model.rendering_network)):
        a = torch.cat([p.grad.detach().reshape(-1).double().cpu() for _, p in net.named_parameters()])
        b = torch.cat([ref[f"{tag}.{k}"].reshape(-1).double() for k, _ in net.named_parameters()])
        cos = float(torch.dot(a, b) / (a.norm() * b.norm()))
        print(f"{name}/{precision}: {tag} gradient vs the oracle's unpinned autograd: cosine {cos:.6f}, length ratio {float(a.norm() / b.norm()):.5f}")
        # observed over the eight fixtures: cosine >= 0.99997 (0.9998 with the opt-in two-product forward), length within 6e-4
        assert cos > (0.9995 if "c2" in opts or "bf16dy" in opts else 0.9999) and abs(float(a.norm() / b.norm()) - 1.0) < 3e-3, (tag, cos)
    # ReLU kinks: two fp32-accurate implementations can disagree on the sign of a pre-activation that is ~1e-7 from
    # zero, which legitimately changes the gradient of everything below (one unit of a 64-ray fixture moves a layer's
    # gradient by percents).  Count such flips from the saved activations; when there are any, the tight comparison is
    # made against the oracle re-run with THIS implementation's masks pinned.
    # the masks the backward applied: the sign-bit words of the 16-bit path (exact: a positive activation below the f16 range
    # is stored as 0 with f16 storage but its unit is open), the saved values of the fp32 path (row-major [slots][M][256])
    open_units = lib.unpack_sign_words(model._debug_masks).cpu() if model._debug_masks is not None else model._debug_saved.cpu() > 0
    if getattr(model, "_debug_dst", None) is not None:     # one VF evaluation per distinct sample: the workspace is in STORAGE order
        dst = model._debug_dst.cpu().long()                 # (proposal samples, then the new ones); row r is sorted sample dst[r]
        sorted_units = torch.empty_like(open_units)
        sorted_units[:, dst] = open_units
        open_units = sorted_units
    slots = list(range(8)) + list(range(9, 13))          # VF hidden 0..7, rendering hidden 0..3 (slot 8 = features)
    masks, flips = [], 0
    for slot, act in zip(slots, ref["_hidden"]):
        w = act.shape[1]
        masks.append(open_units[slot][:, :w])
        flips += int((masks[-1] != (act > 0)).sum())
    print(f"{name}/{precision}: ReLU sign flips between HIP and CPU activations: {flips}")
    # (an 11-bit forward in the colour branch moves more units across zero — measured 0.013 .. 0.078 flips per point over the eight
    # fixtures, i.e. one in ~15 000 of the colour branch's units — the oracle re-runs with these masks)
    assert flips <= (max(4, d["z_vals"].numel() // 1000) if "c2" not in opts else max(8, d["z_vals"].numel() // 10))
    if flips:
        ref_loss, ref = oracle_gradients(fx, d, build_model(fx, d), masks=masks)
        assert abs(loss - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    worst = ("", 0.0)
    errs = []
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    for tag, net in nets.items():
        for pname, p in net.named_parameters():
            assert p.grad is not None, (tag, pname)
            err = grad_rel_err(p.grad, ref[f"{tag}.{pname}"])
            if err > worst[1]:
                worst = (f"{tag}.{pname}", err)
            errs.append((tag, pname, err))
    for pname, p in model.density.named_parameters():
        err = grad_rel_err(p.grad.reshape(1), ref[f"density.{pname}"].reshape(1))
        print(f"density.{pname}: hip {float(p.grad):.6e} ref {float(ref['density.' + pname]):.6e}")
        assert err < (1e-2 if "bf16dy" in opts else TOL), (pname, err)
    print(f"{name}/{precision}: worst parameter-gradient error {worst[1]:.3e} at {worst[0]}")
    # bf16 gradient storage (opt-in): 8 significant bits in one factor of every dW term; the rounding is unbiased and averages
    # out over the points of a batch, so on these fixtures of a few hundred to a few thousand points it is still visible
    # (Backward): bounded at 1e-2 here, ~1e-4 at the 524 288 points of a full batch
    tol = MODE_TOL[precision]
    if name == "bench_sizes" and "bf16dy" not in opts:
        tol = BENCH_SIZES_TOL["stored16" if ("f16act" in opts or "f16dy" in opts) else "exact"]
    assert all(e < tol for _, _, e in errs), [x for x in errs if x[2] >= tol]
    # and against the reference's own backward pass (captured in the fixture): the mode's own bound when no unit flipped; 1e-2 when
    # at most four did (one unit on the other side of zero moves a layer's gradient by ~0.2 % on fixtures of a few hundred points);
    # a sanity bound beyond that (the opt-in 11-bit colour branch and bf16 storage move dozens of units)
    # (observed, round 4: 1.1e-2 with ONE flipped unit on the smallest fixtures — a BatchNorm gain of the first layer, whose gradient is a
    # sum over a few hundred points — hence 2e-2, not 1e-2; the density scalars, sums over every ray of terms the scale of 100 amplifies,
    # sit at 1.1e-4 .. 1.5e-4 from the reference's with the exact kernels and no flip: bounded at 5e-4)
    tol_ref = max(tol, 1e-3 if name == "bench_sizes" else 0.0) if flips == 0 else (max(tol, 2e-2) if flips <= 4 else 1e-1)
    worst_ref = 0.0
    for tag, key in GRAD_KEYS:
        err = grad_rel_err(dict(nets[tag].named_parameters())[key].grad, d[f"grad.{tag}.{key}"])
        worst_ref = max(worst_ref, err)
        assert err < tol_ref, ("vs reference", tag, key, err, flips)
    for k in ("beta", "mean", "scale"):
        err = grad_rel_err(getattr(model.density, k).grad.reshape(1), d[f"grad.density.{k}"])
        worst_ref = max(worst_ref, err)
        assert err < max(tol_ref, 5e-4), ("vs reference", k, err, flips)
    print(f"{name}/{precision}: worst error vs the REFERENCE's captured gradients {worst_ref:.3e} with {flips} flipped units (bound {tol_ref:.0e})")


def test_training_forward_workspace_f16x3_matches_fp32():
    """The two activation-saving forwards (vfn_vf_render_fused16_fwd_train / vfn_vf_render_fused_fwd_train and the VF-only
    pair) fill the same workspace: every hidden layer's output, both encoding tiles, normals and colours."""
    from vf_nerf_amd import lib
    from vf_nerf_amd.backward import _Workspace, _entries
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device="cuda:0")
    vf, rn = model.vector_field_network, model.rendering_network
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    n, s_t = 37, 12                                    # 444 points: ragged last workgroup
    pts = (torch.rand(n * s_t, 3, generator=gen) * 2 - 1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1).to(dev)
    m, slots = n * s_t, len(_entries(vf)) + len(_entries(rn))
    ws32, ws16 = _Workspace(m, slots, dev), _Workspace(m, slots, dev)
    ws16.saved.zero_()          # columns 224..255 of the 217-wide layer's slot are never written: define them for the sign-word check
    n32, c32 = lib.vf_render_fused_fwd_train(vf.geometry(), vf.packed_weights(), rn.geometry(), rn.packed_weights(), pts,
                                             dirs, s_t, ws32.saved, ws32.aux_vf, ws32.aux_rn)
    n16, c16 = lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(),
                                               pts, dirs, s_t, ws16.saved, ws16.aux_vf, ws16.aux_rn, ws16.masks)
    assert float((n16 - n32).abs().max()) < 2e-5 and float((c16 - c32).abs().max()) < 2e-5
    # the sign-bit words next to the saved activations (what the bf16 chain reads instead of them) are exactly their signs
    relu_slots = [sl for sl in range(13) if sl != 8]          # slot 8 = the tanh'ed feature block: no mask, the chain reads its values
    assert torch.equal(ws16.masks[relu_slots], lib.relu_sign_words(ws16.saved[relu_slots]))
    assert float((ws16.aux_vf[:, :39] - ws32.aux_vf[:, :39]).abs().max()) < 1e-6
    assert float((ws16.aux_rn[:, :33] - ws32.aux_rn[:, :33]).abs().max()) < 2e-5      # holds the normals
    widths = [256, 256, 256, 217, 256, 256, 256, 256, 256, 256, 256, 256, 256]
    for slot, w in enumerate(widths):
        a, b = ws16.saved[slot][:, :w], ws32.saved[slot][:, :w]
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) < 2e-5 * scale, (slot, float((a - b).abs().max()), scale)
    # VF-only variants (supervision points), with and without the feature block
    for with_feat in (False, True):
        w32, w16 = _Workspace(m, len(_entries(vf)), dev), _Workspace(m, len(_entries(vf)), dev)
        w16.saved.zero_()
        o32 = lib.vf_mlp_fwd_train(vf.geometry(), vf.packed_weights(), pts, 259 if with_feat else 3, w32.saved, w32.aux_vf)
        o16 = lib.vf_mlp16_fwd_train(vf.geometry(), vf.packed16_weights(), pts, with_feat, w16.saved, w16.aux_vf, w16.masks)
        assert float((o16 - o32[:, :3]).abs().max()) < 2e-5
        assert torch.equal(w16.masks[:8], lib.relu_sign_words(w16.saved[:8]))
        for slot in range(9 if with_feat else 8):
            w = widths[slot]
            scale = max(1.0, float(w32.saved[slot][:, :w].abs().max()))
            assert float((w16.saved[slot][:, :w] - w32.saved[slot][:, :w]).abs().max()) < 2e-5 * scale, (with_feat, slot)


def test_supervision_forward_gradients():
    """vector_field_network(points)[:, :3] as the trainer uses it for border / centre supervision
    (train/vector_field_nerf_train.py:191,203,215), full-row and vector-only variants."""
    from oracle import vfnerf_oracle as O
    fx, d = load_fixture("odd_orbit")
    model = build_model(fx, d, device="cuda:0")
    vf = model.vector_field_network
    gen = torch.Generator().manual_seed(3)
    pts = (torch.rand(333, 3, generator=gen) - 0.5) * 2
    coef = torch.randn(333, 3, generator=gen)
    cpu_sd = {k: v.detach().cpu().clone() for k, v in vf.state_dict().items()}
    for name, _ in vf.named_parameters():
        cpu_sd[name].requires_grad_(True)
    (O.vf_mlp(pts, cpu_sd)[:, :3] * coef).sum().backward()
    for variant in ("full", "vector_only"):
        for p in vf.parameters():
            p.grad = None
        out = vf(pts.to("cuda:0")) if variant == "full" else vf(pts.to("cuda:0"), vector_only=True)
        (out[:, :3] * coef.to("cuda:0")).sum().backward()
        worst = 0.0
        for name, p in vf.named_parameters():
            ref = cpu_sd[name].grad
            if variant == "vector_only" and name.startswith("layers.8"):
                err = grad_rel_err(p.grad[:3], ref[:3])
            else:
                err = grad_rel_err(p.grad, ref)
            worst = max(worst, err)
            assert err < TOL, (variant, name, err)
        print(f"supervision forward ({variant}): worst gradient error {worst:.3e}")


def test_one_adam_step_matches_oracle_step():
    """zero_grad -> backward -> clip_grad_norm_ (with the duplicated parameter list, Q4) -> Adam.step, as
    train/vector_field_nerf_train.py:251-260 does, against the same sequence driven by the oracle's gradients."""
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    # exact-fp32 forward: this test is about the optimizer sequence; a ReLU unit landing on the other side of zero
    # (which the split-f16 forward does on this fixture, see test_render_gradients) would change the gradients by percents
    model.precision = "fp32"
    ref_model = build_model(fx, d)
    # GPU side: the facade's own optimizer (optim.SequentialAdam) and optim.clip_grad_norm_ — multi-tensor kernels, one
    # pass per multiplicity; CPU side: PyTorch's sequential per-entry loops, i.e. the semantics the reference ran with.
    from vf_nerf_amd import optim
    lr = model.config.scheduler_config.lr
    assert isinstance(model.optimizer, optim.SequentialAdam)
    ref_model.optimizer = torch.optim.Adam(ref_model.parameters(), lr=lr, foreach=False)
    _hip_gradients(fx, d, model)
    optim.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
    model.optimizer.step()
    _, ref = oracle_gradients(fx, d, ref_model)
    for tag, net in (("vf", ref_model.vector_field_network), ("rn", ref_model.rendering_network)):
        for pname, p in net.named_parameters():
            p.grad = ref[f"{tag}.{pname}"].clone()
    for pname, p in ref_model.density.named_parameters():
        p.grad = ref[f"density.{pname}"].reshape(p.shape).clone()
    torch.nn.utils.clip_grad_norm_(ref_model.parameters(), ref_model.config.scheduler_config.clip_norm, foreach=False)
    ref_model.optimizer.step()
    # Adam's first step is lr * g / (|g| + 1e-8) = +-lr per application (twice for the aliased VF net, Q4): it is
    # ill-conditioned where |g| is at rounding-noise level, so compare where the gradient is significant.
    worst = 0.0
    for tag, net_a, net_b in (("vf", model.vector_field_network, ref_model.vector_field_network),
                              ("rn", model.rendering_network, ref_model.rendering_network)):
        for (n1, p1), (n2, p2) in zip(net_a.named_parameters(), net_b.named_parameters()):
            gref = ref[f"{tag}.{n1}"]
            sig = gref.abs() > 1e-3 * gref.abs().max()
            diff = (p1.detach().cpu() - p2.detach()).abs()
            worst = max(worst, float(diff[sig].max()))
    print(f"max parameter difference after one clipped Adam step (significant gradients): {worst:.3e}")
    assert worst < 2e-5
    w0 = build_model(fx, d).vector_field_network.layers[8].weight.detach()
    step = (ref_model.vector_field_network.layers[8].weight.detach() - w0).abs().max()
    assert abs(float(step) - 2 * lr) < 0.1 * lr, "the aliased VF parameters receive two Adam updates per step (Q4)"


def test_training_loop_tracks_the_cpu_path():
    """Config 3 in miniature: the trainer's step (train/vector_field_nerf_train.py:177-260 — render, supervision
    forward, VFLoss, zero_grad, backward, clip_grad_norm_, Adam with the duplicated parameter list, ExponentialLR) run
    on the HIP path and on the CPU oracle from identical weights, draws and targets.  The loss curves must agree while
    the two trajectories are the same trajectory: Adam's normalised steps amplify rounding differences, and as soon as
    one proposal argmax (a discrete event) differs between the paths the fine samples of that ray differ and the
    curves part (observed around step 6-10 with this deliberately jumpy synthetic setup) — so the first six steps are
    asserted and the rest is reported."""
    from helpers import oracle_settings
    from oracle import vfnerf_oracle as O
    fx, d = load_fixture("c1_perturb")
    n, s_c, n_f = 32, fx["n_samples"], fx["n_importance"]
    steps = 12
    gen = torch.Generator().manual_seed(77)
    uv, pose, K = d["uv"][:n], d["pose"][:n], d["intrinsics"][:n]
    rgb_gt, depth_gt = torch.rand(n, 3, generator=gen), 0.2 + 0.6 * torch.rand(n, 1, generator=gen)
    sup_pts = torch.rand(200, 3, generator=gen) * 2 - 1
    sup_gt = torch.nn.functional.normalize(torch.randn(200, 3, generator=gen), dim=1)
    draws = [dict(u_coarse=torch.rand(n, s_c, generator=gen), u_fine=torch.rand(n, n_f, generator=gen),
                  u_add=torch.rand(n, n_f, generator=gen)) for _ in range(steps)]
    w = O.LossWeights()

    def run(device):
        model = build_model(fx, d, device=device)
        if device != "cpu":
            model.precision = "fp32"     # a trajectory comparison: keep every discrete event (ReLU kinks) on the CPU's side
        lr = model.config.scheduler_config.lr
        if device == "cpu":
            model.optimizer = torch.optim.Adam(model.parameters(), lr=lr, foreach=False)  # PyTorch's sequential loop
        # (the GPU side keeps the facade's optim.SequentialAdam)
        model.scheduler = torch.optim.lr_scheduler.ExponentialLR(model.optimizer, 0.1 ** (1. / 50000))
        losses = []
        for t in range(steps):
            if device == "cpu":
                live = lambda net: {**dict(net.named_parameters()), **dict(net.named_buffers())}
                out = O.render(uv, pose, K, live(model.vector_field_network), live(model.rendering_network),
                               oracle_settings(fx), beta=model.density.beta, mean=model.density.mean,
                               scale=model.density.scale, **draws[t])
                rgb, depth, normals = out["rgb"], out["depth"], out["normals"]
                sup = O.vf_mlp(sup_pts, live(model.vector_field_network))[:, :3]
                tgt = (rgb_gt, depth_gt, sup_gt)
            else:
                dev = torch.device(device)
                o = model.render(pose.to(dev), uv.to(dev), K.to(dev), 0, uniforms={k: v.to(dev) for k, v in draws[t].items()})
                rgb, depth, normals = o.coarse_rgb_values, o.coarse_depth_map, o.coarse_normals
                sup = model.vector_field_network(sup_pts.to(dev))[:, :3]
                tgt = (rgb_gt.to(dev), depth_gt.to(dev), sup_gt.to(dev))
            loss = O.vf_loss(rgb, depth, normals, sup, tgt[0], tgt[1], tgt[2], w, epoch=0)
            model.optimizer.zero_grad()
            loss.backward()
            if device == "cpu":
                torch.nn.utils.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm, foreach=False)
            else:
                from vf_nerf_amd import optim
                optim.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
            model.optimizer.step()
            model.scheduler.step()
            losses.append(float(loss))
        return losses, model

    hip_losses, hip_model = run("cuda:0")
    cpu_losses, cpu_model = run("cpu")
    rel = [abs(a - b) / max(abs(b), 1e-6) for a, b in zip(hip_losses, cpu_losses)]
    print("loss curve (hip | cpu | rel diff):")
    for t in range(steps):
        print(f"  step {t:2d}: {hip_losses[t]:.6f} | {cpu_losses[t]:.6f} | {rel[t]:.2e}")
    assert rel[0] < 1e-5, "first step: identical weights, so the forward must agree to rounding"
    assert max(rel[:6]) < 1e-3
    assert all(map(lambda v: v == v and abs(v) < 1e6, hip_losses + cpu_losses)), "no NaN / blow-up on either path"
    drift = max(float((p.detach().cpu() - q.detach()).abs().max())
                for p, q in zip(hip_model.unique_parameters(), cpu_model.unique_parameters()))
    print(f"max parameter drift after {steps} steps: {drift:.3e} (each step moves a weight by <= 2 lr = 1e-3)")


def test_weight_gradient_bf16_split_matches_fp32():
    """vfn_weight_grad_partials_bf16 against the exact-fp32 kernel on the same operands: exact on small integers (which
    pins the transposed-read operand maps), ~2^-16 per product on random data spanning gradient-like magnitudes, ragged
    point counts and more groups than steps."""
    from vf_nerf_amd import lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    for m, groups, kind in ((64, 1, "int"), (1000, 3, "int"), (4099, 7, "rand"), (50, 4, "rand"), (70000, 64, "rand")):
        if kind == "int":
            dy = torch.randint(-3, 4, (m, 256), generator=gen).float()
            x = torch.randint(-3, 4, (m, 256), generator=gen).float()
        else:
            dy = torch.randn(m, 256, generator=gen) * torch.logspace(-9, -3, 256)[None, :]     # columns of very different scale
            x = torch.relu(torch.randn(m, 256, generator=gen)) * 30.0
        dy, x = dy.to(dev), x.to(dev)
        p32, b32 = torch.empty(groups, 256, 256, device=dev), torch.empty(groups, 256, device=dev)
        p16, b16 = torch.empty(groups, 256, 256, device=dev), torch.empty(groups, 256, device=dev)
        lib.weight_grad_partials(0, dy, 256, 256, x, 256, 256, m, groups, p32, b32)
        lib.weight_grad_partials_bf16(dy, x, m, groups, p16, b16)
        g32, g16 = p32.sum(0), p16.sum(0)
        ref = (dy.double().t() @ x.double()).float()
        if kind == "int":
            assert torch.equal(g16, ref) and torch.equal(g32, ref), (m, groups)
        else:
            row_scale = ref.abs().amax(dim=1, keepdim=True).clamp_min(1e-30)      # per output row: rows differ by 1e6
            err16 = float(((g16 - ref).abs() / row_scale).max())
            err32 = float(((g32 - ref).abs() / row_scale).max())
            print(f"m={m} groups={groups}: bf16-split err {err16:.2e}, fp32 kernel err {err32:.2e} (relative to each row's max)")
            assert err16 < 2e-4, (m, groups, err16)
        assert float((b16.sum(0) - dy.sum(0)).abs().max()) <= 1e-5 * max(1.0, float(dy.abs().sum(0).max()))


def test_dx_chain_bf16_split_matches_fp32():
    """vfn_mlp_bwd_chain_bf16 against the exact-fp32 chain on the same workspace: every dY slot and both head gradients,
    in the fused mode and the two VF-only modes (with / without a gradient on the features).  Tolerance 2e-4 of each
    slot's largest entry (16-bit operands under 256-term sums; observed ~3e-5)."""
    from vf_nerf_amd import lib
    from vf_nerf_amd.backward import _Workspace, _entries, _packed_bwd, _packed_bwd16, _head_rows
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device="cuda:0")
    vf, rn = model.vector_field_network, model.rendering_network
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    n, s_t = 41, 13                                    # 533 points: ragged last workgroup
    m = n * s_t
    pts = (torch.rand(m, 3, generator=gen) * 2 - 1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1).to(dev)
    vf_h, rn_h = len(_entries(vf)), len(_entries(rn))
    ws = _Workspace(m, vf_h + rn_h, dev)
    normals, colors = lib.vf_render_fused_fwd_train(vf.geometry(), vf.packed_weights(), rn.geometry(), rn.packed_weights(), pts,
                                                    dirs, s_t, ws.saved, ws.aux_vf, ws.aux_rn)
    dc = (torch.randn(m, 3, generator=gen) * 1e-4).to(dev)
    dn = (torch.randn(m, 3, generator=gen) * 1e-5).to(dev)

    def compare(tag, a, b, slots):
        for sl in slots:
            scale = float(b[sl].abs().max())
            err = float((a[sl] - b[sl]).abs().max())
            assert err <= 2e-4 * scale + 1e-30, (tag, sl, err, scale)

    dy32, dy16 = torch.zeros(vf_h + rn_h, m, 256, device=dev), torch.zeros(vf_h + rn_h, m, 256, device=dev)
    zr32, zv32, zr16, zv16 = (torch.empty(m, 4, device=dev) for _ in range(4))
    lib.mlp_bwd_chain(vf.geometry(), vf.packed_weights(), _packed_bwd(vf), rn.geometry(), rn.packed_weights(), _packed_bwd(rn),
                      ws.saved, dy32, dc, colors, dn, normals, None, 3, m, zr32, zv32)
    masks = lib.relu_sign_words(ws.saved)      # the workspace of this test comes from the fp32 forward: derive the words
    lib.mlp_bwd_chain_bf16(vf.geometry(), _packed_bwd16(vf), _head_rows(vf), rn.geometry(), _packed_bwd16(rn), _head_rows(rn),
                           ws.saved, masks, dy16, dc, colors, dn, normals, None, 3, m, zr16, zv16)
    assert torch.equal(zr16, zr32) and torch.equal(zv16, zv32)
    widths = [256, 256, 256, 217, 256, 256, 256, 256, 256, 256, 256, 256, 256]
    compare("fused", [dy16[s][:, :w] for s, w in enumerate(widths)], [dy32[s][:, :w] for s, w in enumerate(widths)], range(13))
    # VF-only: full rows (gradient on vector + features) and vector-only
    out = torch.cat([normals, ws.saved[vf_h - 1]], dim=1).contiguous()
    d_out = (torch.randn(m, 259, generator=gen) * 1e-3).to(dev)
    for cols in (259, 3):
        dy32, dy16 = torch.zeros(vf_h, m, 256, device=dev), torch.zeros(vf_h, m, 256, device=dev)
        o = out if cols == 259 else normals
        g = d_out if cols == 259 else d_out[:, :3].contiguous()
        from vf_nerf_amd.backward import _offset_view
        dfe = _offset_view(g, 3) if cols == 259 else None
        lib.mlp_bwd_chain(vf.geometry(), vf.packed_weights(), _packed_bwd(vf), None, None, None, ws.saved, dy32, None, None, g, o,
                          dfe, cols, m, None, zv32)
        lib.mlp_bwd_chain_bf16(vf.geometry(), _packed_bwd16(vf), _head_rows(vf), None, None, None, ws.saved, masks, dy16, None, None, g, o,
                               dfe, cols, m, None, zv16)
        assert torch.equal(zv16, zv32)
        slots = range(9) if cols == 259 else range(8)
        compare(f"vf-only/{cols}", [dy16[s][:, :w] for s, w in enumerate(widths[:9])], [dy32[s][:, :w] for s, w in enumerate(widths[:9])], slots)


def test_fragment_ordered_workspace_holds_the_row_major_values():
    """The f16x3 training forward and the bf16 chain with a FRAGMENT-ORDERED workspace (include/vfn.h) against the same
    launches with the row-major one: bit-identical activations (fp32 and f16 storage), sign words, outputs and dY slots;
    bf16 dY = the fp32 gradients rounded to nearest even.  Ragged point counts (partial last group)."""
    from vf_nerf_amd import lib
    from vf_nerf_amd.backward import _Workspace, _entries, _packed_bwd16, _head_rows
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device="cuda:0")
    vf, rn = model.vector_field_network, model.rendering_network
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    widths = [256, 256, 256, 217, 256, 256, 256, 256, 256, 256, 256, 256, 256]
    for n, s_t in ((37, 12), (1, 3), (64, 16)):        # 444 (ragged), 3 (one partial group), 1024 (whole groups) points
        m = n * s_t
        pts = (torch.rand(m, 3, generator=gen) * 2 - 1).to(dev)
        dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1).to(dev)
        slots = len(_entries(vf)) + len(_entries(rn))
        dc = (torch.randn(m, 3, generator=gen) * 1e-4).to(dev)
        dn = (torch.randn(m, 3, generator=gen) * 1e-5).to(dev)
        for f16 in (False, True):
            rows_ws, frag_ws = _Workspace(m, slots, dev, f16=f16), _Workspace(m, slots, dev, f16=f16, frag=True)
            outs = []
            for ws in (rows_ws, frag_ws):
                ws.saved.fill_(float("nan"))
                outs.append(lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(),
                                                            pts, dirs, s_t, ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags()))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
            assert torch.equal(rows_ws.masks, frag_ws.masks)
            for slot, w in enumerate(widths):
                a, b = frag_ws.rows(slot)[:, :w], rows_ws.rows(slot)[:, :w]
                assert torch.equal(a, b), (m, f16, slot)
            if f16:
                continue
            zr, zv = torch.empty(m, 4, device=dev), torch.empty(m, 4, device=dev)
            dys = []
            for ws, flags in ((rows_ws, 0), (frag_ws, lib.DY_FRAG), (frag_ws, lib.DY_FRAG | lib.DY_BF16), (frag_ws, lib.DY_FRAG | lib.DY_F16S)):
                dy = ws.new_dy()
                dy.fill_(float("nan"))
                lib.mlp_bwd_chain_bf16_ws(vf.geometry(), _packed_bwd16(vf), _head_rows(vf), rn.geometry(), _packed_bwd16(rn), _head_rows(rn),
                                          ws.feats(8), ws.masks, dy, flags, dc, outs[0][1], dn, outs[0][0], None, 3, m, zr, zv)
                dys.append(dy)
            for slot, w in enumerate(widths):
                want = dys[0][slot][:, :w]
                assert torch.equal(lib.frag_to_rows(dys[1][slot], m)[:, :w], want), (m, slot)
                got16 = lib.frag_to_rows(dys[2][slot], m, torch.bfloat16)[:, :w]
                assert torch.equal(got16, want.to(torch.bfloat16).float()), (m, slot, "bf16")
                # scaled f16 (dY form 3): exactly what the host-side encoder makes of the fp32 gradients, exponent bytes included
                # (padding lanes of the last group carry 255; the values of those lanes are never read)
                full = lib.frag_to_rows(dys[1][slot], m)            # all columns as the fragment-ordered fp32 run holds them
                tiles = (w + 31) // 32                              # (slot 3 has seven tiles: the eighth is never written; its exponent
                full[:, 32 * tiles:] = 0                            # bytes must say "all zero")
                enc = lib.rows_to_frag_f16s(full).view(-1, lib.GROUP_FLOATS)
                got = dys[3][slot].view(-1, lib.GROUP_FLOATS)
                assert torch.equal(lib.frag_f16s_to_rows(got, m)[:, :32 * tiles], lib.frag_f16s_to_rows(enc, m)[:, :32 * tiles]), (m, slot, "f16s values")
                eb_got = got.view(torch.uint8)[:, lib.F16S_EXP_OFF:lib.F16S_EXP_OFF + 512].reshape(-1, 8, 2, 32)
                eb_enc = enc.view(torch.uint8)[:, lib.F16S_EXP_OFF:lib.F16S_EXP_OFF + 512].reshape(-1, 8, 2, 32)
                assert torch.equal(eb_got, eb_enc), (m, slot, "f16s exponents")
                rel = float((lib.frag_f16s_to_rows(got, m)[:, :w] - want).abs().max() / want.abs().max().clamp_min(1e-30))
                assert rel < 2 ** -11, (m, slot, rel)


@pytest.mark.parametrize("m,groups", [(32 * 40, 5), (1000, 7), (33, 1), (5000, 64)])
def test_weight_grad_frag_matches_float64(m, groups):
    """vfn_weight_grad_frag (csrc/vfn_dwf.hip) in every operand form against a float64 product of the same operands: the three
    shapes (256 x 256, 256 x 64 with the encoding tile, 32 x 256 with the head gradient), fragment fp32 / f16 / bf16 / row-major
    operands, ragged point counts, more groups than steps.  Small integers make the 16-bit splits exact (this pins the operand
    maps, the swizzled LDS images and the transposed reads); random data bounds the split error."""
    from vf_nerf_amd import lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(m)

    def operands(exact):
        if exact:
            dy = torch.randint(-3, 4, (m, 256), generator=gen).float()
            x = torch.randint(-4, 5, (m, 256), generator=gen).float()
            aux = torch.randint(-2, 3, (m, 40), generator=gen).float()
            dz = torch.randint(-3, 4, (m, 4), generator=gen).float()
        else:
            dy = torch.randn(m, 256, generator=gen) * torch.logspace(-6, -2, 256)[None, :]
            x = torch.relu(torch.randn(m, 256, generator=gen)) * 3
            aux = torch.randn(m, 40, generator=gen)
            dz = torch.randn(m, 4, generator=gen) * 1e-3
        return dy.to(dev), x.to(dev), aux.to(dev), dz.to(dev)

    for exact in (True, False):
        dy, x, aux, dz = operands(exact)
        dz[:, 3] = 0
        f16s = lib.rows_to_frag_f16s(dy)                # tile-scaled f16 (what the chain writes with gradient_storage = "f16")
        forms_dy = {lib.DYF_FRAG32: (lib.rows_to_frag(dy), dy), lib.DYF_FRAGBF16: (lib.rows_to_frag(dy, torch.bfloat16), dy.to(torch.bfloat16).float()),
                    lib.DYF_FRAGF16S: (f16s, lib.frag_f16s_to_rows(f16s, m))}
        if exact:
            assert torch.equal(forms_dy[lib.DYF_FRAGF16S][1], dy)
        else:
            assert float((forms_dy[lib.DYF_FRAGF16S][1] - dy).abs().max() / dy.abs().max()) < 2 ** -11
        forms_x = {lib.XF_FRAG32: (lib.rows_to_frag(x), x), lib.XF_FRAG16: (lib.rows_to_frag(x, torch.float16), x.half().float()),
                   lib.XF_ROWS32: (x.contiguous(), x)}
        tol = 0.0 if exact else 3e-4

        def check(tag, got, want, got_b, want_b):
            scale = float(want.abs().max())
            err = float((got.double() - want).abs().max())
            assert err <= tol * scale + (0 if not exact else 0), (tag, m, groups, exact, err, scale)
            if got_b is not None:
                assert float((got_b.double() - want_b).abs().max()) <= (1e-5 if not exact else 0) * max(1e-30, float(want_b.abs().max())), (tag, "db")

        for fdy, (bdy, vdy) in forms_dy.items():
            for fx_, (bx, vx) in forms_x.items():
                part = torch.full((groups, 256, 256), float("nan"), device=dev)
                dbp = torch.full((groups, 256), float("nan"), device=dev)
                lib.weight_grad_frag(0, bdy, fdy, bx, fx_, m, groups, part, dbp)
                check(f"shape0/{fdy}/{fx_}", part.sum(0), vdy.double().T @ vx.double(), dbp.sum(0), vdy.double().sum(0))
            part = torch.full((groups, 256, 64), float("nan"), device=dev)
            dbp = torch.full((groups, 256), float("nan"), device=dev)
            lib.weight_grad_frag(1, bdy, fdy, aux, lib.XF_AUX40, m, groups, part, dbp)
            check(f"shape1/{fdy}", part.sum(0)[:, :40], vdy.double().T @ aux.double(), dbp.sum(0), vdy.double().sum(0))
            assert float(part.sum(0)[:, 40:].abs().max()) == 0.0
        for fx_, (bx, vx) in forms_x.items():
            part = torch.full((groups, 32, 256), float("nan"), device=dev)
            dbp = torch.full((groups, 32), float("nan"), device=dev)
            lib.weight_grad_frag(2, dz, lib.DYF_DZ4, bx, fx_, m, groups, part, dbp)
            check(f"shape2/{fx_}", part.sum(0)[:4], dz.double().T @ vx.double(), dbp.sum(0)[:4], dz.double().sum(0))
            assert float(part.sum(0)[4:].abs().max()) == 0.0 and float(dbp.sum(0)[4:].abs().max()) == 0.0


@pytest.mark.parametrize("storage", ["f16", "fp32", "f16+bf16dy", "f16+f16dy", "fp32+f16dy"])
def test_one_call_weight_gradients_equal_the_launch_by_launch_path(storage):
    """vfn_net_weight_grads_frag (csrc/vfn_wgrad.hip: every weight-gradient launch of a net and the un-fold issued from C out of
    one scratch buffer, results added into the parameters' .grad) against the facade's launch-by-launch sequence: the same
    products on the same data — since round 3 the C path issues the products of one shape as ONE launch with fewer, larger partial
    slabs each, so the sums over points run in another order: every gradient within 1e-5 of the tensor's largest entry (observed
    ~1e-6) — through the fused fine pass (both nets), a vector-only supervision forward (feature block skipped) and a full VF
    forward, accumulated into the same .grad tensors."""
    fx, d = load_fixture("shipped_sizes")
    g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    a, b, c = (t.to("cuda:0") for t in loss_coefficients(*d["z_vals"].shape))
    sup = torch.rand(777, 3, generator=torch.Generator().manual_seed(5)).to("cuda:0") - 0.5
    opts = storage.split("+")
    grads = {}
    for one_call in (True, False):
        model = build_model(fx, d, device="cuda:0")
        model.activation_storage = opts[0]
        model.gradient_storage = "bf16" if "bf16dy" in opts else ("f16" if "f16dy" in opts else "fp32")
        for net in (model.vector_field_network, model.rendering_network):
            net.one_call_weight_grads = one_call
        calls = []
        real = lib.net_weight_grads_frag
        lib.net_weight_grads_frag = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            model.optimizer.zero_grad()                      # FlatAdam: .grad tensors exist (views of one buffer), zeroed
            assert all(p.grad is not None for p in model.unique_parameters())
            out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
            vf = model.vector_field_network
            loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum() + \
                vf(sup, vector_only=True).pow(2).sum() + vf(sup[:300]).sum()
            loss.backward()
        finally:
            lib.net_weight_grads_frag = real
        assert len(calls) == (4 if one_call else 0), calls   # fine pass: two nets; two stand-alone VF forwards
        grads[one_call] = {f"{tag}.{n}": p.grad.detach().clone()
                           for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density))
                           for n, p in mod.named_parameters() if p.grad is not None}
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 40
    worst = 0.0
    for name in grads[True]:
        x, y = grads[True][name], grads[False][name]
        if name.startswith("density."):                     # float atomics across workgroups (vfn_density_bwd_kernel): not bit-stable
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-6), name
        else:
            err = float((x - y).abs().max()) / max(float(y.abs().max()), 1e-30)
            worst = max(worst, err)
            assert err < 1e-5, (name, err)
    print(f"{storage}: worst difference between the one-call and the launch-by-launch weight gradients {worst:.2e} of a tensor's largest entry")


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit", "shipped_sizes"])
def test_training_render_with_one_vf_evaluation_per_sample(name):
    """backward.StoredFinePass (the activation-saving forward on the proposal samples first, on the new samples after the
    sampler, one workspace in storage order) against the fused forward over the sorted samples with its separate gradient-free
    proposal pass (``model.reuse_proposal_training = False``): every forward output bit-identical (same per-sample arithmetic,
    same draws), parameter gradients equal up to the order of the sums over points (1e-5 of each tensor's largest entry; the
    fixtures whose proposal block is not a multiple of 32 points fall back to the sorted path and must agree exactly)."""
    fx, d = load_fixture(name)
    g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    a, b, c = (t.to("cuda:0") for t in loss_coefficients(*d["z_vals"].shape))
    res = {}
    for stored in (True, False):
        model = build_model(fx, d, device="cuda:0")
        model.gradient_storage = "fp32"                 # (the comparison is about the ORDER of the sums, not their storage)
        model.reuse_proposal_training = stored
        model._keep_saved, model._debug_dst = True, None
        model.optimizer.zero_grad()
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
        loss.backward()
        grads = {f"{tag}.{n}": p.grad.detach().clone()
                 for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density))
                 for n, p in mod.named_parameters()}
        res[stored] = (out, grads, model._debug_dst is not None)
    (o1, g1, used), (o0, g0, _) = res[True], res[False]
    n_rays, s_c = d["z_vals"].shape[0], fx["n_samples"]
    assert used == ((n_rays * s_c) % 32 == 0), "the stored path runs when the proposal block is whole groups of 32 points"
    for f in ("z_vals", "points_coarse", "coarse_normals", "coarse_colors", "coarse_rgb_values", "coarse_depth_map"):
        assert torch.equal(getattr(o1, f), getattr(o0, f)), f
    worst = 0.0
    for k in g1:
        scale = float(g0[k].abs().max())
        err = float((g1[k] - g0[k]).abs().max()) / max(scale, 1e-30)
        worst = max(worst, err)
        assert err < (1e-5 if used else 1e-12) or k.startswith("density."), (k, err)
    print(f"{name}: stored path used: {used}; worst gradient difference {worst:.2e}")


@pytest.mark.parametrize("scale", [1e-30, 1e-12, 1e-5, 1.0, 1e12, 1e25, 1e33])
def test_scaled_f16_gradients_at_any_magnitude(scale):
    """dY form 3 (csrc/vfn_dwf.hip) carries its own power-of-two scale per lane and tile, so the weight gradient it yields must
    be as accurate for gradients of 1e-30 as for gradients of 1e+25 (a batch-mean loss puts dY far below the f16 range; nothing
    bounds it from above either), with rows of wildly different magnitude in one slab, all-zero tiles and all-zero points."""
    from vf_nerf_amd import lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    m, groups = 3000, 9
    dy = torch.randn(m, 256, generator=gen) * torch.logspace(-3, 0, 256)[None, :] * torch.logspace(0, -4, m)[:, None]
    dy[:, 64:96] = 0                       # an all-zero tile everywhere
    dy[100:164] = 0                        # two all-zero groups
    dy = (dy * scale).to(dev)
    x = (torch.relu(torch.randn(m, 256, generator=gen)) * 3).to(dev)
    slot = lib.rows_to_frag_f16s(dy)
    held = lib.frag_f16s_to_rows(slot, m)
    assert float((held - dy).abs().max() / dy.abs().max()) < 2 ** -11
    part = torch.full((groups, 256, 256), float("nan"), device=dev)
    dbp = torch.full((groups, 256), float("nan"), device=dev)
    lib.weight_grad_frag(0, slot, lib.DYF_FRAGF16S, lib.rows_to_frag(x, torch.float16), lib.XF_FRAG16, m, groups, part, dbp)
    want = held.double().T @ x.half().double()
    got = part.sum(0).double()
    assert torch.isfinite(got).all()
    err = float((got - want).abs().max() / want.abs().max())
    print(f"scale {scale:g}: dW error {err:.2e} of the largest entry")
    assert err < 1e-5
    assert float((dbp.sum(0).double() - held.double().sum(0)).abs().max() / held.double().sum(0).abs().max()) < 1e-5


@pytest.mark.parametrize("name", ["c1_perturb", "bench_sizes"])
def test_shared_step_workspace_gives_the_same_gradients(name):
    """backward.StepWorkspace: vector-field forwards that follow a render() under autograd append their points to the render's
    workspace (ragged batches padded with points of zero upstream gradient), every backward runs its own dX chain, and ONE
    sequence of weight-gradient launches at the end of the backward pass covers all of them.  Against private workspaces
    (``model.shared_step_workspace = False``): forward values bit-identical, every parameter gradient equal up to the order of
    the sums (1e-5 of the tensor's largest entry) — with a full forward (features evaluated, their gradient zero: what the
    trainer's ``vector_field_network(points)[:, :3]`` is), a vector-only forward, a batch too large for the room that is left
    (falls back to its own workspace) and a forward that takes no part in the loss."""
    fx, d = load_fixture(name)
    g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    a, b, c = (t.to("cuda:0") for t in loss_coefficients(*d["z_vals"].shape))
    gen = torch.Generator().manual_seed(3)
    m_fine = d["z_vals"].numel()
    sup_a = (torch.rand(max(33, m_fine // 10) + 5, 3, generator=gen) - 0.5).to("cuda:0")          # ragged
    sup_b = (torch.rand(64, 3, generator=gen) - 0.5).to("cuda:0")
    sup_big = (torch.rand(m_fine, 3, generator=gen) - 0.5).to("cuda:0")                            # does not fit behind the fine pass
    sup_unused = (torch.rand(40, 3, generator=gen) - 0.5).to("cuda:0")
    res = {}
    for shared in (True, False):
        model = build_model(fx, d, device="cuda:0")
        model.shared_step_workspace = shared
        model.step_sessions = False                                   # (the pool of the launch-by-launch autograd path is what is tested here)
        vf = model.vector_field_network
        model.optimizer.zero_grad()                                   # gradient views exist: results are added in place
        out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
        pool = getattr(vf, "_step_ws", None)
        assert (pool is not None) == (shared and m_fine % 32 == 0 or shared)       # (a ragged fine pass still owns a pool, without room)
        full = vf(sup_a)                                              # [M, 259]
        vec = vf(sup_b, vector_only=True)
        big = vf(sup_big)[:, :3]
        _ = vf(sup_unused)                                            # appended, never differentiated
        if shared and m_fine % 32 == 0:
            assert pool.next > _round32_(m_fine), "the supervision forwards joined the render's workspace"
        loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum() + \
            full[:, :3].pow(2).sum() + vec.pow(2).sum() + 0.1 * big.sum()
        loss.backward()
        res[shared] = (out, full.detach(), vec.detach(), {n: p.grad.detach().clone() for n, p in _named(model)})
    (o1, f1, v1, g1), (o0, f0, v0, g0) = res[True], res[False]
    assert torch.equal(o1.coarse_rgb_values, o0.coarse_rgb_values) and torch.equal(o1.coarse_normals, o0.coarse_normals)
    assert torch.equal(f1, f0) and torch.equal(v1, v0)
    worst = 0.0
    for k in g1:
        if k.startswith("density."):
            continue
        scale = max(float(g0[k].abs().max()), 1e-30)
        e1 = float((g1[k] - g0[k]).abs().max()) / scale
        worst = max(worst, e1)
        assert e1 < 1e-5, (k, e1)
    print(f"{name}: shared vs private workspaces: worst gradient difference {worst:.2e}")


def _round32_(n):
    return (n + 31) // 32 * 32


def _named(model):
    for tag, mod in (("vf", model.vector_field_network), ("rn", model.rendering_network), ("density", model.density)):
        for n, p in mod.named_parameters():
            yield f"{tag}.{n}", p
