"""The opt-in 16-bit-native training mode (``model.training_products = 1``; BASELINE.json configs[2], "bf16 MFMA MLPs" as written).

The saving forward multiplies f16 roundings (one product per K-block, csrc/vfn_mlp16.hip M16_P1), the dX chain bf16 roundings
(csrc/vfn_bwd16.hip BM_P1); the weight-gradient kernels are the default ones.  The mode is OUTSIDE the 1e-4 / 1e-3 contracts by
construction, so nothing here compares it with the oracle at those tolerances: the tests pin what it must still be —
 * the same function to within what 11 / 8 significant bits per operand allow (outputs, loss, gradient direction and length
   against the default three-product kernels on the same state and the same draws);
 * invisible to gradient-free renders (bit-identical outputs);
 * able to learn: the trainer's step brings the loss on teacher-rendered targets down as the default path does.
Observed figures are printed; the long three-stream comparison is tools/train_curve.py -> profiles/r03/train_curve_p1.json."""
import pytest
import torch

from helpers import build_model, load_fixture, loss_coefficients

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _step(model, d, products):
    model.training_products = products
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    for p in model.unique_parameters():
        p.grad = None
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    a, b, c = (t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape))
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    grads = {tag: torch.cat([p.grad.detach().reshape(-1).double() for p in net.parameters()])
             for tag, net in (("vf", model.vector_field_network), ("rn", model.rendering_network))}
    return float(loss), out, grads


def _cos_ratio(a, b):
    return float(torch.dot(a, b) / (a.norm() * b.norm())), float(a.norm() / b.norm())


@pytest.mark.parametrize("name", ["c1_perturb", "bench_sizes", "trained_256"])
def test_single_product_step_against_the_default_kernels(name):
    """One differentiated render with single products against the same call on the default kernels (same state, same draws).
    Yardstick for the gradients: how far the DEFAULT kernels' gradient moves when every weight moves by one f16 rounding (uniform in
    +-2^-11 relative) — on the 64 + 64-sample fixture that is a lot (cosine 0.74, length x1.8: the Laplace density of scale 100 turns a
    normal 5e-3 away into a derivative e^0.5 away), on the others 1e-3; the single-product step must stay within three such moves."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device=DEV)
    assert model.training_products == 3
    loss3, out3, g3 = _step(model, d, 3)
    loss1, out1, g1 = _step(model, d, 1)
    keep = (out1.z_vals == out3.z_vals).all(dim=1)
    same_z = keep.float().mean().item()
    nrm = (out1.coarse_normals[keep] - out3.coarse_normals[keep]).abs().max().item()
    rgb_same = (out1.coarse_rgb_values[keep] - out3.coarse_rgb_values[keep]).abs().max().item()
    depth_same = (out1.coarse_depth_map[keep] - out3.coarse_depth_map[keep]).abs().max().item()
    print(f"{name}: rays with identical samples {same_z:.4f}; on those: max |normal diff| {nrm:.2e}, |rgb diff| {rgb_same:.2e}, |depth diff| "
          f"{depth_same:.2e}; loss {loss1:.6f} vs {loss3:.6f}")
    # 11-bit operands: the vector columns move by ~5e-3, which moves an argmax (and with it a ray's fine samples) now and then
    assert same_z > 0.8 and nrm < 2e-2 and rgb_same < 2e-2 and depth_same < 5e-2
    # the yardstick: default kernels, weights one f16 rounding away
    gen = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for p in model.unique_parameters():
            if p.dim() == 2:
                p.mul_(1.0 + ((torch.rand(p.shape, generator=gen) - 0.5) * 2.0 ** -10).to(p.device))
    _, _, gy = _step(model, d, 3)
    for tag in ("vf", "rn"):
        cos, ratio = _cos_ratio(g1[tag], g3[tag])
        ycos, yratio = _cos_ratio(gy[tag], g3[tag])
        print(f"{name}: {tag} gradient, single product vs three: cosine {cos:.5f}, length ratio {ratio:.4f}   (default kernels with weights one "
              f"f16 rounding away: {ycos:.5f}, {yratio:.4f})")
        assert 1.0 - cos <= 3.0 * (1.0 - ycos) + 1e-3 and abs(ratio - 1.0) <= 3.0 * abs(yratio - 1.0) + 0.02, (tag, cos, ratio, ycos, yratio)


@pytest.mark.parametrize("n,s_t", [(37, 12), (1, 3), (64, 16)])
def test_single_product_kernels_against_the_default_ones(n, s_t):
    """Kernel level, on one workspace each (444 points: ragged last workgroup; 3: one partial group; 1 024: whole groups): the saving
    forward with one f16 product per K-block against the three-product one — outputs and every saved activation within 11-bit
    arithmetic, the sign words all but identical — and the chain with one bf16 product per K-block against the three-product chain ON
    THE SAME WORKSPACE: every gradient slot within 8-bit arithmetic accumulated over the layers below it, head gradients identical."""
    from vf_nerf_amd import lib
    from vf_nerf_amd.backward import _Workspace, _entries, _packed_bwd16, _head_rows
    fx, d = load_fixture("trained_256")
    model = build_model(fx, d, device=DEV)
    vf, rn = model.vector_field_network, model.rendering_network
    dev = torch.device(DEV)
    gen = torch.Generator().manual_seed(5)
    widths = [256, 256, 256, 217, 256, 256, 256, 256, 256, 256, 256, 256, 256]
    m = n * s_t
    pts = (torch.rand(m, 3, generator=gen) * 2 - 1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1).to(dev)
    slots = len(_entries(vf)) + len(_entries(rn))
    ws3 = _Workspace(m, slots, dev, f16=True, frag=True, dy16="f16")
    ws1 = _Workspace(m, slots, dev, f16=True, frag=True, dy16="f16p1")
    assert ws1.single and not ws3.single and ws1.dy16 == "f16"
    outs = {}
    for tag, ws, products in (("three", ws3, 3), ("single", ws1, 1)):
        ws.saved.zero_()
        outs[tag] = lib.vf_render_fused16_fwd_train(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, s_t,
                                                    ws.saved, ws.aux_vf, ws.aux_rn, ws.masks, save_f16=ws.fwd_flags(), colour_products=products)
    dn = float((outs["single"][0] - outs["three"][0]).abs().max())
    dc = float((outs["single"][1] - outs["three"][1]).abs().max())
    assert 0 < dn < 2e-2 and dc < 2e-2, (dn, dc)
    assert torch.equal(ws1.aux_vf, ws3.aux_vf) and torch.equal(ws1.aux_rn[:, :30], ws3.aux_rn[:, :30])     # (columns 30..32 are the normal)
    worst = 0.0
    for slot, w in enumerate(widths):
        a, b = ws1.rows(slot)[:, :w], ws3.rows(slot)[:, :w]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        assert rel < 1e-2, (slot, rel)
    flips = int((lib.unpack_sign_words(ws1.masks) != lib.unpack_sign_words(ws3.masks)).sum())
    print(f"{m} points: normals {dn:.2e}, colours {dc:.2e}, worst slot (Frobenius, relative) {worst:.2e}, sign flips {flips} of {13 * m * 256}")
    assert flips <= max(8, 13 * m * 256 // 200)
    # vector-only forward (supervision points)
    wv3, wv1 = _Workspace(m, len(_entries(vf)), dev, f16=True, frag=True, dy16="f16"), _Workspace(m, len(_entries(vf)), dev, f16=True, frag=True, dy16="f16p1")
    v3 = lib.vf_mlp16_fwd_train(vf.geometry(), vf.packed16_weights(), pts, False, wv3.saved, wv3.aux_vf, wv3.masks, save_f16=wv3.fwd_flags())
    v1 = lib.vf_mlp16_fwd_train(vf.geometry(), vf.packed16_weights(), pts, False, wv1.saved, wv1.aux_vf, wv1.masks, save_f16=wv1.fwd_flags())
    assert torch.equal(v3, outs["three"][0]) and torch.equal(v1, outs["single"][0]), "the vector columns do not depend on the launch shape"
    with pytest.raises(lib.VfnError):
        lib.vf_mlp16_fwd_train(vf.geometry(), vf.packed16_weights(), pts, True, wv1.saved, wv1.aux_vf, wv1.masks, save_f16=wv1.fwd_flags())
    # the chain, both arithmetics on the THREE-product workspace
    dcol = (torch.randn(m, 3, generator=gen) * 1e-4).to(dev)
    dnrm = (torch.randn(m, 3, generator=gen) * 1e-5).to(dev)
    res = {}
    for tag, flags, rounded in (("three", ws3.dy_flags(), False), ("single", ws3.dy_flags() | lib.DY_P1, True)):
        dy = ws3.new_dy()
        dy.zero_()
        zr, zv = torch.empty(m, 4, device=dev), torch.empty(m, 4, device=dev)
        lib.mlp_bwd_chain_bf16_ws(vf.geometry(), _packed_bwd16(vf, rounded), _head_rows(vf), rn.geometry(), _packed_bwd16(rn, rounded), _head_rows(rn),
                                  ws3.feats(8), ws3.masks, dy, flags, dcol, outs["three"][1], dnrm, outs["three"][0], None, 3, m, zr, zv)
        res[tag] = (dy, zr, zv)
    assert torch.equal(res["single"][1], res["three"][1]) and torch.equal(res["single"][2], res["three"][2])
    worst = 0.0
    for slot, w in enumerate(widths):
        a = lib.frag_f16s_to_rows(res["single"][0][slot].view(-1, lib.GROUP_FLOATS), m)[:, :w]
        b = lib.frag_f16s_to_rows(res["three"][0][slot].view(-1, lib.GROUP_FLOATS), m)[:, :w]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
        worst = max(worst, rel)
        assert torch.isfinite(a).all() and rel < 3e-2, (slot, rel)
        if slot == 12:           # the slot the fused chain starts from: no matrix product yet
            assert rel < 2e-3, (slot, rel)
    print(f"{m} points: chain, worst gradient slot (Frobenius, relative) {worst:.2e}")


def test_single_product_mode_leaves_gradient_free_renders_alone():
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device=DEV)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    outs = []
    for products in (3, 1):
        model.training_products = products
        with torch.no_grad():
            outs.append(model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni))
    for key in ("coarse_rgb_values", "coarse_depth_map", "coarse_normals", "z_vals"):
        assert torch.equal(getattr(outs[0], key), getattr(outs[1], key)), key
    with pytest.raises(ValueError):
        model.training_products = 2


def test_single_product_supervision_forward():
    """The vector-only forward on supervision points (train/vector_field_nerf_train.py:196-214) under autograd: values within 11-bit
    arithmetic of the default kernels', the parameter gradient pointing the same way; a [vector | features] forward keeps three
    products (the single-product kernels evaluate the vector columns only)."""
    fx, d = load_fixture("trained_256")
    model = build_model(fx, d, device=DEV)
    net = model.vector_field_network
    gen = torch.Generator().manual_seed(3)
    pts = (torch.rand(3000, 3, generator=gen) * 2 - 1).to(DEV)
    w = torch.randn(3000, 3, generator=gen).to(DEV)
    res = {}
    for products in (3, 1):
        model.training_products = products
        for p in net.parameters():
            p.grad = None
        out = net(pts, vector_only=True)
        (out * w).sum().backward()
        res[products] = (out.detach().clone(), torch.cat([p.grad.reshape(-1).double() for p in net.parameters() if p.grad is not None]))
    diff = (res[1][0] - res[3][0]).abs().max().item()
    cos = float(torch.dot(res[1][1], res[3][1]) / (res[1][1].norm() * res[3][1].norm()))
    ratio = float(res[1][1].norm() / res[3][1].norm())
    print(f"vector columns: max |single - three| {diff:.2e}; gradient cosine {cos:.5f}, length ratio {ratio:.4f}")
    assert 0 < diff < 2e-2 and cos > 0.99 and abs(ratio - 1.0) < 0.05
    full3 = net(pts)
    model.training_products = 3
    assert torch.equal(full3, net(pts)), "a forward that returns the features runs on three products in either mode"


def test_single_product_training_converges():
    """trainer.TrainStep on teacher-rendered targets (as test_training_converges_on_teacher_targets): 300 steps of 1 024 rays from
    the same state and batches with the default kernels and with single products — both bring the loss below 0.75 of its start and
    end within the same wide band of each other."""
    import vf_nerf_amd
    from vf_nerf_amd import supervision, synthetic, trainer
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
    final, first_loss = {}, {}
    for products in (3, 1):
        model = scene(0)
        model.training_products = products
        model.rng_seed, model._rng_offset = 11, 0
        supervision.manual_seed(3)
        step = trainer.TrainStep(model, (0.0, 0.0, 0.55), border_radius=0.15, far=1.0)
        losses = []
        for t in range(300):
            pose, uv, K, rgb_gt, depth_gt = pool.batch(t, 1024)
            losses.append(step(pose, uv, K, rgb_gt, depth_gt, epoch=0)[0])
        losses = [float(x) for x in losses]
        first, last = sum(losses[:10]) / 10, sum(losses[-50:]) / 50
        print(f"[products={products}] loss first 10 steps {first:.4f} -> last 50 steps {last:.4f} (x{last / first:.3f}); PSNR to the teacher "
              f"{pool.psnr(model):.2f} dB")
        assert all(x == x for x in losses), "no NaN"
        assert last < 0.75 * first, (products, first, last)       # (observed over repeated runs: 0.42 .. 0.59 for either arithmetic)
        final[products], first_loss[products] = last, losses[0]
    print(f"final loss single / three = {final[1] / final[3]:.4f}; first-step losses {first_loss[3]:.6f} / {first_loss[1]:.6f}")
    assert abs(first_loss[1] - first_loss[3]) < 2e-2 * first_loss[3]
    assert 0.5 < final[1] / final[3] < 2.0
