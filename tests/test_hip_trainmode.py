"""Networks in TRAINING mode on the GPU (``VectorFieldNerf.train()``: batch-statistics BatchNorm, the VF forward's three
autograd.grad rows, analytic directional derivatives) against the CPU oracle and against vectors the
reference itself produced (tests/golden/train_mode.npz).  Tolerances are written at each comparison; fp32 throughout."""
import pytest
import torch

from helpers import GRAD_KEYS, build_model, grad_rel_err, load_fixture, loss_coefficients, oracle_gradients, rel_err
from oracle import vfnerf_oracle as O
from vf_nerf_amd import lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("m,k,n", [(1, 39, 256), (130, 256, 217), (257, 256, 259), (1000, 289, 256), (77, 256, 3), (4096, 256, 256)])
@pytest.mark.parametrize("transpose", [False, True])
@pytest.mark.parametrize("arith", ["exact", "split_f16", "split_bf16", "bf16x6"])
def test_linear_rows_against_float64(m, k, n, transpose, arith):
    """C = tanh(A W^T + b) / C = A W on the matrix cores vs float64 on the CPU.  Exact fp32 instruction: 1e-6 of the largest |z|
    (products are exact fp32, the sums of k terms round in fp32).  Split f16 (three f16 products per product, 22 significant bits,
    what the forward GEMMs of the training-mode networks run on): the same 1e-6.  Split bf16 (16 significant bits with fp32's range,
    16 bits): 3e-5 — and, scaled down by 1e-6 like a gradient, still 3e-5 of ITS largest entry (no underflow).  bf16 in three parts (six
    products, 24 bits: the backward GEMMs of the split mode): 1e-6 at any scale.
    Column sums of z and z^2 within 1e-5 / the arithmetic's own bound."""
    torch.manual_seed(m + k + n)
    kp, ldc = (k + 7) & ~7, (n + 7) & ~7
    a = torch.zeros(m, kp)
    a[:, :k] = torch.randn(m, k)
    w = torch.randn(k, n) if transpose else torch.randn(n, k)
    b = None if transpose else torch.randn(n)
    c = torch.full((m, ldc), 7.0, device=DEV)
    parts = lib.linear_rows_stat_parts(m)
    part = torch.empty(parts, 2, n, device=DEV)
    mode = {"exact": lib.GEMM_EXACT, "split_f16": lib.GEMM_SPLIT_F16, "split_bf16": lib.GEMM_SPLIT_BF16, "bf16x6": lib.GEMM_BF16X6}[arith]
    tol = 3e-5 if arith == "split_bf16" else 1e-6
    lib.linear_rows(a.to(DEV), w.to(DEV), None if b is None else b.to(DEV), m, n, k, c, act=lib.ACT_NONE if transpose else lib.ACT_TANH,
                    transpose_w=transpose, stats_part=part, arith=mode)
    z = a[:, :k].double() @ (w.double() if transpose else w.double().t()) + (0 if b is None else b.double())
    want = z if transpose else torch.tanh(z)
    got = c.cpu()
    err = float((got[:, :n].double() - want).abs().max()) / max(1.0, float(z.abs().max()))
    print(f"{arith} m={m} k={k} n={n} transpose={transpose}: {err:.2e} of the largest |z|")
    assert err <= tol
    assert bool((got[:, n:] == 7.0).all()), "columns past n_out must not be written"
    sums = torch.empty(2 * n, dtype=torch.float64, device=DEV)
    lib.colsum_finish(part, parts, 2 * n, sums)
    s = sums.cpu().view(2, n)
    assert float((s[0] - z.sum(0)).abs().max()) <= max(1e-5, 10 * tol) * float(z.abs().sum(0).max())
    assert float((s[1] - (z * z).sum(0)).abs().max()) <= max(1e-5, 10 * tol) * float((z * z).sum(0).max())
    if arith in ("split_bf16", "bf16x6") and transpose:          # gradient-sized operands: bf16 parts keep fp32's exponent range
        c2 = torch.zeros((m, ldc), device=DEV)
        lib.linear_rows((a * 1e-6).to(DEV), w.to(DEV), None, m, n, k, c2, transpose_w=True, arith=mode)
        assert float((c2.cpu()[:, :n].double() - 1e-6 * z).abs().max()) <= tol * 1e-6 * max(1.0, float(z.abs().max()))


@pytest.mark.parametrize("arith", [lib.GEMM_SPLIT_F16, lib.GEMM_SPLIT_BF16], ids=["f16x3", "bf16x3"])
@pytest.mark.parametrize("m,k,n,transpose", [(4096, 256, 256, False), (5000, 256, 256, True), (300, 289, 256, False), (4096, 256, 259, False),
                                             (1000, 295, 256, True), (130, 39, 256, False), (4096, 256, 512, False)])
def test_layer_product_with_presplit_weight_planes_is_bit_identical(m, k, n, transpose, arith):
    """vfn_linear_rows_ws: W split ONCE per call into its 16-bit planes (one small launch) instead of by every workgroup for every chunk —
    the same operands, hence the same C bit for bit and the same column-sum partials; shapes with ragged rows, a K that is not a whole number
    of chunks, more than one block of 256 output columns, both orientations of W."""
    torch.manual_seed(m + k + n)
    kp = (k + 7) & ~7
    a = torch.zeros(m, kp, device=DEV)
    a[:, :k] = torch.randn(m, k, device=DEV) * (1e-4 if transpose else 1.0)
    w = (torch.randn(k, n, device=DEV) if transpose else torch.randn(n, k, device=DEV)) * 0.1
    bias = None if transpose else torch.randn(n, device=DEV)
    ldc = (n + 7) & ~7
    c1, c2 = torch.zeros(m, ldc, device=DEV), torch.zeros(m, ldc, device=DEV)
    parts = lib.linear_rows_stat_parts(m)
    p1, p2 = torch.zeros(parts, 2, n, device=DEV), torch.zeros(parts, 2, n, device=DEV)
    lib.linear_rows(a, w, bias, m, n, k, c1, transpose_w=transpose, stats_part=p1, arith=arith)
    lib.linear_rows(a, w, bias, m, n, k, c2, transpose_w=transpose, stats_part=p2, arith=arith, planes=lib.wplanes(n, k, DEV))
    assert torch.equal(c1, c2) and torch.equal(p1, p2) and float(c1.abs().max()) > 0


@pytest.mark.parametrize("arith", [lib.GEMM_SPLIT_BF16, lib.GEMM_BF16X6], ids=["bf16x3", "bf16x6"])
@pytest.mark.parametrize("m,k_out,n,n_prev", [(4096, 256, 256, 256), (4096, 259 + 5, 259, 256), (5000, 295, 256, 217), (4096, 256, 256, 217),
                                              (130, 256, 256, 64), (1, 289, 256, 256)])
def test_dx_product_with_the_batchnorm_backward_sums(m, k_out, n, n_prev, arith):
    """vfn_linear_rows_dx_sums: C = dZ W bit-identical to vfn_linear_rows (the same kernel body), and the two column sums of the previous
    layer's BatchNorm backward, taken from C in registers, equal the pass of their own (vfn_bstat_relu_bwd_sums) and float64 within 1e-6
    of the largest sum (fp32 partial sums of 128 / 64 rows, finished in double by both).  Shapes: a hidden layer, the last Linear
    (259 outputs), the skip layer (295 inputs of which 217 come from the BatchNorm'ed layer), ragged row counts."""
    torch.manual_seed(m + k_out + n_prev)
    up8 = lambda v: (v + 7) & ~7
    dz = torch.zeros(m, up8(n), device=DEV)
    dz[:, :n] = torch.randn(m, n, device=DEV)
    w = torch.randn(n, k_out, device=DEV) * 0.1
    zp = torch.randn(m, up8(n_prev), device=DEV)
    coef = torch.stack([torch.rand(n_prev, device=DEV) + 0.5, torch.randn(n_prev, device=DEV) * 0.3, torch.randn(n_prev, device=DEV) * 0.1,
                        torch.rand(n_prev, device=DEV) + 0.5]).contiguous()
    post = 0.7
    g1 = torch.zeros(m, up8(k_out), device=DEV)
    g2 = torch.zeros_like(g1)
    lib.linear_rows(dz, w, None, m, k_out, n, g1, transpose_w=True, arith=arith)
    p1 = lib.bstat_row_parts(m)
    part1 = torch.empty(p1, 2, n_prev, device=DEV)
    lib.bstat_relu_bwd_sums(g1, zp, coef, m, n_prev, post, part1)
    s1 = torch.empty(2, n_prev, dtype=torch.float64, device=DEV)
    lib.colsum_finish(part1, p1, 2 * n_prev, s1)
    p2 = lib.linear_rows_stat_parts(m)
    part2 = torch.empty(p2, 2, n_prev, device=DEV)
    lib.linear_rows_dx_sums(dz, w, m, k_out, n, g2, zp, coef, n_prev, post, part2, arith=arith,
                            planes=lib.wplanes(k_out, n, DEV) if arith == lib.GEMM_SPLIT_BF16 else None)
    s2 = torch.empty(2, n_prev, dtype=torch.float64, device=DEV)
    lib.colsum_finish(part2, p2, 2 * n_prev, s2)
    assert torch.equal(g1, g2)
    gd, zd, c = g1[:, :n_prev].double(), zp[:, :n_prev].double(), coef.double()
    mask = torch.addcmul(coef[1], zp[:, :n_prev], coef[0]) > 0          # (fp32, like the kernels; ties at exactly 0 have measure zero here)
    gp = torch.where(mask, post * gd, torch.zeros_like(gd))
    want = torch.stack([gp.sum(0), (gp * ((zd - c[2]) * c[3])).sum(0)])
    scale = float(want.abs().max())
    e_fused, e_pass = float((s2 - want).abs().max()) / scale, float((s1 - want).abs().max()) / scale
    print(f"column sums vs float64: fused {e_fused:.2e}, pass of its own {e_pass:.2e}")
    assert e_fused <= 1e-6 and e_pass <= 1e-6


@pytest.mark.parametrize("m,ld_dy,c_dy,ld_x,c_x", [(4096, 256, 0, 256, 0), (3000, 264, 0, 256, 0), (4096, 256, 0, 296, 0), (1000, 264, 0, 296, 32),
                                                     (33, 256, 0, 256, 0)])
def test_weight_gradient_blocks_of_wider_matrices_on_the_bf16_cores(m, ld_dy, c_dy, ld_x, c_x):
    """vfn_weight_grad_partials_bf16_ld: dW[n][k] = sum_m dY[m][c_dy + n] X[m][c_x + k] over 256 x 256 column blocks of matrices whose rows
    are wider than 256 floats (the 259-wide output gradient, the 289-wide rendering-net input), and db = column sums of the dY block, vs
    float64: 3e-5 of the largest entry (bf16 split operands, three products: 16 significant bits per operand under a sum over m)."""
    torch.manual_seed(m + ld_dy + ld_x)
    dy = torch.randn(m, ld_dy, device=DEV) * 1e-3
    x = torch.randn(m, ld_x, device=DEV)
    G = 4
    part = torch.empty(G, 256, 256, device=DEV)
    db = torch.empty(G, 256, device=DEV)
    lib.weight_grad_partials_bf16_cols(lib.Cols(dy, c_dy), lib.Cols(x, c_x), m, G, part, db)
    got_w, got_b = part.double().sum(0).cpu(), db.double().sum(0).cpu()
    dyb, xb = dy[:, c_dy:c_dy + 256].double().cpu(), x[:, c_x:c_x + 256].double().cpu()
    want_w, want_b = dyb.t() @ xb, dyb.sum(0)
    e_w = float((got_w - want_w).abs().max() / want_w.abs().max())
    e_b = float((got_b - want_b).abs().max() / want_b.abs().max())
    print(f"dW block rel err {e_w:.2e}, db {e_b:.2e}")
    assert e_w <= 3e-5 and e_b <= 1e-5


def test_split_f16_layer_product_reports_operands_outside_its_range():
    """The split f16 form carries A at 64x its value as two f16 halves: |a| >= 1023 overflows the high half.  Like the fused f16x3
    kernels the launch then sets bit 0 of the calling thread's range-report word (vfn_f16x3_set_status), which the facade's range guard
    turns into a move to the exact products (guard.py; batchstat._split reads model.precision); in range, the word stays clear; the
    bf16 and exact forms do not report (they have fp32's range)."""
    torch.manual_seed(0)
    m, k, n = 300, 256, 256
    a = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.1
    c = torch.empty(m, n, device=DEV)
    word = torch.zeros(4, dtype=torch.int32, device=DEV)
    lib.f16x3_set_status(word)
    try:
        lib.linear_rows(a, w, None, m, n, k, c, arith=lib.GEMM_SPLIT_F16)
        assert int(word[0]) == 0
        a[137, 200] = 1500.0
        lib.linear_rows(a, w, None, m, n, k, c, arith=lib.GEMM_BF16X6)
        lib.linear_rows(a, w, None, m, n, k, c, arith=lib.GEMM_EXACT)
        assert int(word[0]) == 0
        want = a.double() @ w.double().t()
        assert float((c.double() - want).abs().max() / want.abs().max()) <= 1e-6          # (the exact form's result, the last one written)
        lib.linear_rows(a, w, None, m, n, k, c, arith=lib.GEMM_SPLIT_F16)
        assert int(word[0]) & lib.STATUS_ACT_SATURATED
    finally:
        lib.f16x3_set_status(None)


def test_embed_rows_and_its_derivative():
    """Positional encoding rows (and the scaled copy the skip layer reads) bit-match the oracle's; the backward is the
    analytic derivative (checked against float64 autograd, 1e-5 of the largest entry)."""
    torch.manual_seed(3)
    m, L = 300, 6
    pts = torch.rand(m, 3) * 2 - 1
    dst = torch.zeros(m, 48, device=DEV)
    lib.embed_rows(pts.to(DEV), m, L, lib.Cols(dst, 4))
    want = O.positional_encoding(pts, L)
    assert float((dst.cpu()[:, 4:43] - want).abs().max()) <= 1e-6
    assert float(dst.cpu()[:, :4].abs().max()) == 0.0 and float(dst.cpu()[:, 43:].abs().max()) == 0.0
    ga, gb = torch.randn(m, 40), torch.randn(m, 256)
    p64 = pts.double().requires_grad_(True)
    pe = O.positional_encoding(p64, L)
    (pe * ga[:, :39].double()).sum().backward(retain_graph=True)
    g1 = p64.grad.clone()
    p64.grad = None
    (pe * 0.5 * gb[:, 217:256].double()).sum().backward()
    d = torch.empty(m, 3, device=DEV)
    lib.embed_rows_bwd(pts.to(DEV), m, L, ga.to(DEV), 1.0, lib.Cols(gb.to(DEV), 217), 0.5, d)
    want_d = g1 + p64.grad
    assert float((d.cpu().double() - want_d).abs().max()) <= 1e-5 * float(want_d.abs().max())


def _train_model(fx, d):
    model = build_model(fx, d, device=DEV)
    model.train()
    assert model.vector_field_network.training and model.rendering_network.training
    return model


def test_training_mode_render_outside_the_split_f16_range():
    """(a) A camera 5 000 units from the origin: raw coordinates beyond the split f16 form's 1 023 enter the first layer and the skip layer
    only, whose forward products run in the wide-range form (batchstat._arith): the render stays on the split products, finite.
    (b) A BatchNorm gain of 3 000 pushes hidden activations beyond the range: the launch reports it, the strict range guard repeats the
    call on the exact fp32 products (model.precision = "fp32", which batchstat._split honours) and returns what a model set to the exact
    products from the start returns."""
    import warnings
    fx, d = load_fixture("train_mode")
    uni = {k: d[k].to(DEV) for k in ("u_coarse", "u_fine", "u_add")}

    def render(model, pose):
        with torch.no_grad():
            return model.render(pose.to(DEV), d["uv"].to(DEV), d["intrinsics"].to(DEV), 0, False, uniforms=uni)

    def pair(prepare):
        out = []
        for precision in ("fp32", "f16x3"):
            model = _train_model(fx, d)
            prepare(model)
            model.precision = precision
            model.f16x3_guard = "strict"
            out.append(model)
        return out

    # (a)
    far = d["pose"].clone()
    far[..., :3, 3] += 5000.0
    exact, model = pair(lambda m: None)
    got = render(model, far)
    assert model.precision == "f16x3" and model.f16x3_disabled is None
    assert bool(torch.isfinite(got.coarse_rgb_values).all()) and bool(torch.isfinite(got.coarse_normals).all())
    # (no accuracy statement out there: with batch statistics, a unit-sized scene 5 000 units from the origin cancels several of fp32's seven
    #  digits in the first BatchNorm, whatever the arithmetic: the split and the exact products give O(1) different outputs there)

    # (b)
    def blow_up(m):
        with torch.no_grad():
            m.vector_field_network._bn(2).weight.mul_(3000.0)
    exact, model = pair(blow_up)
    want = render(exact, d["pose"])
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = render(model, d["pose"])
    assert model.precision == "fp32" and model.f16x3_disabled is not None and any("f16x3" in str(w.message) for w in caught)
    assert bool(torch.isfinite(got.coarse_normals).all()) and torch.equal(got.z_vals, want.z_vals)
    assert float((got.coarse_normals - want.coarse_normals).abs().max()) <= 1e-6
    assert float((got.coarse_rgb_values - want.coarse_rgb_values).abs().max()) <= 1e-6
    # ... and the flagged attempt left no trace in the BatchNorm running statistics: they are finite, they are the exact
    # model's after its ONE call, and every layer counts the batches of one render, not of two
    for net_e, net_g in ((exact.vector_field_network, model.vector_field_network), (exact.rendering_network, model.rendering_network)):
        for i in range(net_g.num_layers):
            bn_e, bn_g = net_e._bn(i), net_g._bn(i)
            if bn_g is None:
                continue
            assert bool(torch.isfinite(bn_g.running_mean).all()) and bool(torch.isfinite(bn_g.running_var).all())
            assert int(bn_g.num_batches_tracked) == int(bn_e.num_batches_tracked)
            scale = float(bn_e.running_var.abs().max()) + float(bn_e.running_mean.abs().max()) + 1.0
            assert float((bn_g.running_mean - bn_e.running_mean).abs().max()) <= 1e-5 * scale
            assert float((bn_g.running_var - bn_e.running_var).abs().max()) <= 1e-5 * scale
    # lazy mode: the flagged call returns clamped but FINITE values, and so are the statistics it advanced
    lazy = _train_model(fx, d)
    blow_up(lazy)
    lazy.precision, lazy.f16x3_guard = "f16x3", "lazy"
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        got_lazy = render(lazy, d["pose"])
    assert bool(torch.isfinite(got_lazy.coarse_normals).all())
    for net in (lazy.vector_field_network, lazy.rendering_network):
        for i in range(net.num_layers):
            bn = net._bn(i)
            if bn is not None:
                assert bool(torch.isfinite(bn.running_mean).all()) and bool(torch.isfinite(bn.running_var).all())


def test_vf_forward_training_mode_against_oracle_and_reference():
    """[M, 3 + 256 + 9] of the VF net on the fixture's proposal samples: network columns within 1e-4 (the '1e-4 rel fp32'
    bar), Jacobian rows within 1e-3 of their largest entry (sums of M per-row terms through 8 BatchNorm backwards), both vs
    the reference's own output; running statistics after one batch vs the oracle's."""
    fx, d = load_fixture("train_mode")
    model = _train_model(fx, d)
    cpu = build_model(fx, d)
    sd = {k: v.clone() for k, v in cpu.vector_field_network.state_dict().items()}
    cfg = O.RenderSettings(n_samples=fx["n_samples"], near=fx["near"], far=fx["far"], perturb=True)
    directions, _, cam_loc = O.ray_directions(d["uv"], d["pose"], d["intrinsics"])
    z_c = O.uniform_z_vals(fx["n_rays"], cfg.n_samples, cfg.near, cfg.far, d["u_coarse"])
    pts = O.points_along_rays(cam_loc, directions, z_c).reshape(-1, 3)
    orc = O.vf_mlp_train(pts.clone(), sd).detach()       # the oracle on THIS host's CPU (it advances sd's running statistics)
    want = d["vf_out_coarse"]                            # the reference's output, captured in the build container
    e_orc = float((orc - want).abs().max() / want.abs().max())
    print(f"oracle on this host vs reference fixture: {e_orc:.2e}")
    assert e_orc <= 1e-4
    got = model.vector_field_network(pts.to(DEV)).detach().cpu()
    assert got.shape == want.shape == (pts.shape[0], 268)
    e_net = float((got[:, :259] - want[:, :259]).abs().max())
    e_jac = float((got[:, 259:] - want[:, 259:]).abs().max()) / float(want[:, 259:].abs().max())
    print(f"train-mode VF forward: network columns max abs err {e_net:.2e}, Jacobian rows rel err {e_jac:.2e}")
    assert e_net <= 1e-4 and e_jac <= 1e-3
    for i in (0, 3, 7):
        bn = model.vector_field_network.layers[i][1]
        assert int(bn.num_batches_tracked) == 1
        for stat in ("running_mean", "running_var"):
            w = sd[f"layers.{i}.1.{stat}"]
            assert float((getattr(bn, stat).cpu() - w).abs().max()) <= 1e-5 * max(1.0, float(w.abs().max())), (i, stat)


def test_render_training_mode_against_reference():
    """render() after train(): outputs, directional derivatives (proposal values listed twice, Q10), gradients of the fixed
    functional through the batch statistics of both networks, running statistics (VF: two batches, rendering net: one) —
    against what the reference produced for the same draws."""
    fx, d = load_fixture("train_mode")
    model = _train_model(fx, d)
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
    for p in model.unique_parameters():
        p.grad = None
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    assert torch.equal(out.z_vals.cpu(), d["z_vals"]), "sampling must replay exactly (argmax of the proposal weights)"
    errs = {}
    for name, got in (("normals", out.coarse_normals), ("colors", out.coarse_colors), ("rgb", out.coarse_rgb_values),
                      ("depth", out.coarse_depth_map)):
        errs[name] = rel_err(got.detach().reshape(d[name].shape), d[name])
    dd, dd_want = out.directional_derivtives.cpu(), d["directional_derivatives"]
    assert dd.shape == dd_want.shape and not out.directional_derivtives.requires_grad
    errs["dd"] = float((dd - dd_want).abs().max()) / float(dd_want.abs().max())
    print("train-mode render vs reference:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs[k] for k in ("normals", "colors", "rgb", "depth")) <= 1e-4 and errs["dd"] <= 1e-3
    a, b, c = (t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape))
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    assert abs(float(loss) - float(d["loss"])) <= 1e-4 * max(1.0, abs(float(d["loss"])))
    nets = {"vf": model.vector_field_network, "rn": model.rendering_network}
    worst = 0.0
    for net, key in GRAD_KEYS:
        got, want = dict(nets[net].named_parameters())[key].grad.cpu(), d[f"grad.{net}.{key}"]
        if key.endswith(".0.bias"):          # a bias in front of batch statistics: exactly zero gradient up to rounding
            assert float(got.abs().max()) <= 1e-4
            continue
        e = grad_rel_err(got, want)
        worst = max(worst, e)
        # (no masks pinned here: one ReLU unit of this 312-row batch on the other side of zero moves a layer's gradient by 0.3-1 %.  The
        # exact fp32 GEMMs land at 1.7e-3 of the reference's gradients, the split ones — fp32-equivalent products, another rounding, other
        # units on the edge — at 1e-2 on the skip layer's weight; the bound is the one test_hip_backward.py uses for <= 4 flipped units.  The
        # comparison with THIS implementation's masks pinned (test_training_mode_gradients_against_oracle_all_parameters) holds both arithmetics to 2e-3)
        assert e <= 2e-2, (net, key, e)
    for name, p in model.density.named_parameters():
        e = grad_rel_err(p.grad.cpu().reshape(1), d[f"grad.density.{name}"])
        worst = max(worst, e)
        assert e <= 2e-2, (name, e)
    print(f"train-mode gradients vs reference: worst rel err {worst:.2e}")
    for net, i in (("vf", 0), ("vf", 3), ("vf", 7), ("rn", 0), ("rn", 3)):
        bn = nets[net].layers[i][1]
        assert int(bn.num_batches_tracked) == int(d[f"bn.{net}.{i}.num_batches_tracked"])
        for stat in ("running_mean", "running_var"):
            want = d[f"bn.{net}.{i}.{stat}"]
            assert float((getattr(bn, stat).cpu() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), (net, i, stat)


def test_training_mode_gradients_against_oracle_all_parameters():
    """Every parameter's gradient (not only the sampled keys of the fixture) vs the oracle's autograd, 2e-3 of each tensor's
    largest entry.  A pre-activation ~1e-7 from zero may land on either side of the ReLU in two fp32 implementations, and in a
    312-row batch one such unit moves a layer's gradient by percents: flips are counted, and when there are any the oracle is
    re-run with THIS implementation's ReLU masks pinned (oracle._relu), as the eval-mode gradient tests do."""
    fx, d = load_fixture("train_mode")
    model = _train_model(fx, d)
    vf, rn = model.vector_field_network, model.rendering_network
    vf._keep_state = rn._keep_state = True
    g = {k: v.to(DEV) for k, v in d.items() if isinstance(v, torch.Tensor)}
    out = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms={k: g[k] for k in ("u_coarse", "u_fine", "u_add")})
    a, b, c = (t.to(DEV) for t in loss_coefficients(*d["z_vals"].shape))
    loss = (out.coarse_rgb_values * a).sum() + (out.coarse_depth_map * b).sum() + (out.coarse_normals * c).sum()
    loss.backward()
    ref_loss, ref = oracle_gradients(fx, d, build_model(fx, d))
    masks = []
    for net in (vf, rn):
        st = net._debug_state
        for i in range(1, net.num_layers):
            width = net._linear(i - 1).out_features
            masks.append((st.x[i][:, :width] > 0).cpu())
    flips = sum(int((mk != (act > 0)).sum()) for mk, act in zip(masks, ref["_hidden"]))
    print(f"train mode: ReLU sign flips between HIP and CPU activations: {flips} of {sum(mk.numel() for mk in masks)}")
    assert len(masks) == len(ref["_hidden"]) == 12 and flips <= 8
    if flips:
        ref_loss, ref = oracle_gradients(fx, d, build_model(fx, d), masks=masks)
    assert abs(float(loss) - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    worst = ("", 0.0)
    for tag, net in (("vf", vf), ("rn", rn)):
        for name, p in net.named_parameters():
            want = ref[f"{tag}.{name}"]
            if float(want.abs().max()) <= 1e-5:      # Linear biases in front of BatchNorm: zero
                assert float(p.grad.abs().max()) <= 1e-4, (tag, name)
                continue
            e = grad_rel_err(p.grad, want)
            worst = max(worst, (f"{tag}.{name}", e), key=lambda t: t[1])
            assert e <= 2e-3, (tag, name, e)
    for name, p in model.density.named_parameters():
        assert grad_rel_err(p.grad.reshape(1), ref[f"density.{name}"].reshape(1)) <= 2e-3, name
    print("train-mode gradients vs oracle: worst", worst)


def test_supervision_forward_in_training_mode_is_differentiable():
    """The trainer's extra calls ``vector_field_network(points)[:, :3]`` (train/vector_field_nerf_train.py:191,203,215) in
    training mode: gradient of sum(out[:, :3] * c) wrt two parameters vs the oracle."""
    fx, d = load_fixture("train_mode")
    model = _train_model(fx, d)
    torch.manual_seed(1)
    pts = torch.rand(500, 3) * 2 - 1
    coef = torch.randn(500, 3)
    out = model.vector_field_network(pts.to(DEV))
    assert out.shape == (500, 268)
    (out[:, :3] * coef.to(DEV)).sum().backward()
    cpu = build_model(fx, d)
    sd = {k: v.clone() for k, v in cpu.vector_field_network.state_dict().items()}
    for k in ("layers.2.0.weight", "layers.8.weight", "layers.5.1.weight"):
        sd[k].requires_grad_(True)
    y = O.vf_mlp(pts, sd, train=True)
    (y[:, :3] * coef).sum().backward()
    for k in ("layers.2.0.weight", "layers.8.weight", "layers.5.1.weight"):
        got = dict(model.vector_field_network.named_parameters())[k].grad
        assert grad_rel_err(got, sd[k].grad) <= 2e-3, (k, grad_rel_err(got, sd[k].grad))


def test_batch_statistics_layer_at_full_size():
    """One BatchNorm'ed layer at the size of the headline chunk (524 288 rows): GEMM, column statistics, finalize and the
    ReLU against torch on the same device in float64 — the partial-sum tree (4 096 workgroup partials per column, finished
    in double) must hold the 1e-6 level of the small cases."""
    from vf_nerf_amd import batchstat
    torch.manual_seed(11)
    m, k, n = 524288, 256, 256
    x = torch.randn(m, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.1
    b, gamma, beta = torch.randn(n, device=DEV), torch.rand(n, device=DEV) + 0.5, torch.randn(n, device=DEV)
    z = torch.empty(m, n, device=DEV)
    parts = lib.linear_rows_stat_parts(m)
    part = torch.empty(parts, 2, n, device=DEV)
    lib.linear_rows(x, w, b, m, n, k, z, stats_part=part)
    sums = torch.empty(2, n, dtype=torch.float64, device=DEV)
    lib.colsum_finish(part, parts, 2 * n, sums)
    rm, rv = torch.zeros(n, device=DEV), torch.ones(n, device=DEV)
    coef = torch.empty(4, n, device=DEV)
    lib.bstat_finalize(sums, m, n, gamma, beta, batchstat.BN_EPS, batchstat.BN_MOMENTUM, rm, rv, coef)
    h = torch.empty(m, n, device=DEV)
    lib.bstat_relu_rows(z, coef, m, n, 1.0, h)
    z64 = x.double() @ w.double().t() + b.double()
    assert float((z.double() - z64).abs().max()) <= 2e-6 * float(z64.abs().max())
    mean, var = z64.mean(0), z64.var(0, unbiased=False)
    assert float((coef[2].double() - mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max()))
    assert float((coef[3].double() - 1 / torch.sqrt(var + 1e-5)).abs().max()) <= 1e-5 * float((1 / torch.sqrt(var + 1e-5)).max())
    want_h = torch.relu((z64 - mean) / torch.sqrt(var + 1e-5) * gamma.double() + beta.double())
    assert float((h.double() - want_h).abs().max()) <= 2e-5
    assert float((rm.double() - 0.1 * mean).abs().max()) <= 1e-6 * max(1.0, float(mean.abs().max()))
    assert float((rv.double() - (0.9 + 0.1 * var * m / (m - 1))).abs().max()) <= 1e-5 * max(1.0, float(var.max()))


def test_standalone_rendering_net_is_differentiable_in_eval_mode():
    """``RenderingNetwork.forward`` with gradients and eval-mode BatchNorm (the facade's get_colors / get_weights_and_color
    under autograd): colours and the gradients wrt two parameters and the features against the oracle's autograd."""
    fx, d = load_fixture("c1_det")
    model = build_model(fx, d, device=DEV)          # eval mode
    rn = model.rendering_network
    torch.manual_seed(2)
    m = 700
    pts, nrm = torch.rand(m, 3) * 2 - 1, torch.nn.functional.normalize(torch.randn(m, 3), dim=1)
    dirs, feats = torch.nn.functional.normalize(torch.randn(m, 3), dim=1), torch.tanh(torch.randn(m, 256))
    coef = torch.randn(m, 3)
    f_gpu = feats.to(DEV).requires_grad_(True)
    out = rn(pts.to(DEV), nrm.to(DEV), dirs.to(DEV), f_gpu)
    (out * coef.to(DEV)).sum().backward()
    cpu = build_model(fx, d)
    sd = {k: v.clone() for k, v in cpu.rendering_network.state_dict().items()}
    keys = ("layers.0.0.weight", "layers.2.0.bias", "layers.3.1.weight", "layers.4.weight")
    for k in keys:
        sd[k].requires_grad_(True)
    f_cpu = feats.clone().requires_grad_(True)
    want = O.render_mlp(pts, nrm, dirs, f_cpu, sd)
    (want * coef).sum().backward()
    assert float((out.detach().cpu() - want.detach()).abs().max()) <= 1e-5
    assert grad_rel_err(f_gpu.grad, f_cpu.grad) <= 1e-3
    for k in keys:
        got = dict(rn.named_parameters())[k].grad
        assert grad_rel_err(got, sd[k].grad) <= 1e-3, (k, grad_rel_err(got, sd[k].grad))


@pytest.mark.parametrize("n_prev,post,m", [(256, 1.0, 5000), (217, 2 ** -0.5, 4133)])
def test_folded_products_equal_the_row_pass_then_the_product(n_prev, post, m):
    """``vfn_linear_rows_fold`` / ``vfn_weight_grad_partials_bf16_fold`` (round 6): the previous layer's BatchNorm + ReLU formed inside the
    operand read — against the row pass ``vfn_bstat_relu_rows`` followed by the plain product: the SAME operand values (the row pass's own
    expression), hence outputs, column statistics and weight-gradient slabs bit for bit.  (217, 1/sqrt 2): the skip layer's shape — 217
    BatchNorm'ed columns and 39 re-injected encoding columns behind them that pass through scaled, without a ReLU; m is not a multiple of
    the 128-row blocks (rows past the end must neither be stored nor reach the range report: a first version flagged max(shift, 0) there)."""
    from vf_nerf_amd import batchstat
    torch.manual_seed(5)
    k, n = 256, 256
    z_prev = torch.randn(m, k, device=DEV) * 3.0
    coef = torch.stack([torch.rand(n_prev, device=DEV) + 0.5, torch.randn(n_prev, device=DEV) * 40.0, torch.zeros(n_prev, device=DEV),
                        torch.ones(n_prev, device=DEV)]).contiguous()               # (shifts of +-40: a row of zeros would activate to 40)
    w, b = torch.randn(n, k, device=DEV) * 0.1, torch.randn(n, device=DEV)
    # the materialised operand
    x = torch.zeros(m, k, device=DEV)
    lib.bstat_relu_rows(z_prev, coef, m, n_prev, post, x)
    if n_prev < k:
        x[:, n_prev:] = z_prev[:, n_prev:] * post
    parts = lib.linear_rows_stat_parts(m)
    outs = []
    for folded in (False, True):
        z, part = torch.empty(m, n, device=DEV), torch.empty(parts, 2, n, device=DEV)
        planes = lib.wplanes(n, k, torch.device(DEV))
        if folded:
            lib.linear_rows_fold(z_prev, coef, n_prev, post, w, b, m, n, k, z, stats_part=part, arith=lib.GEMM_SPLIT_F16, planes=planes)
        else:
            lib.linear_rows(x, w, b, m, n, k, z, stats_part=part, arith=lib.GEMM_SPLIT_F16, planes=planes)
        outs.append((z, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    want = x.double() @ w.double().t() + b.double()
    assert float((outs[1][0].double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    # the weight-gradient product with the same fold on its X operand
    dy = torch.randn(m, n, device=DEV) * 1e-3
    G = batchstat._groups(m)
    slabs = []
    for folded in (False, True):
        dw, db = torch.empty(G, 256, 256, device=DEV), torch.empty(G, 256, device=DEV)
        if folded:
            lib.weight_grad_partials_bf16_fold(dy, z_prev, coef, n_prev, post, m, G, dw, db)
        else:
            lib.weight_grad_partials_bf16_cols(dy, x, m, G, dw, db)
        slabs.append((dw, db))
    assert torch.equal(slabs[0][0], slabs[1][0]) and torch.equal(slabs[0][1], slabs[1][1])
    want_dw = dy.double().t() @ x.double()
    got_dw = slabs[1][0].double().sum(0)
    assert float((got_dw - want_dw).abs().max()) <= 1e-4 * float(want_dw.abs().max())
    with pytest.raises(lib.VfnError, match="coefficients"):
        lib.linear_rows_fold(z_prev, coef[:1], n_prev, post, w, b, m, n, k, outs[0][0], arith=lib.GEMM_SPLIT_F16, planes=lib.wplanes(n, k, torch.device(DEV)))


def test_training_mode_step_is_the_same_with_and_without_the_activation_fold():
    """batchstat.FOLD_ACTIVATIONS: a training-mode render + backward with the fold and with the row pass everywhere — outputs and every
    parameter gradient bit for bit (the folded operands ARE the row pass's values; nothing else changes)."""
    from vf_nerf_amd import batchstat
    fx, d = load_fixture("train_mode")
    uni = {k: d[k].to(DEV) for k in ("u_coarse", "u_fine", "u_add")}
    got = {}
    keep = batchstat.FOLD_ACTIVATIONS
    try:
        for fold in (True, False):
            batchstat.FOLD_ACTIVATIONS = fold
            model = _train_model(fx, d)
            model.optimizer.zero_grad()
            out = model.render(d["pose"].to(DEV), d["uv"].to(DEV), d["intrinsics"].to(DEV), 0, False, uniforms=uni)
            loss = out.coarse_rgb_values.abs().mean() + out.coarse_depth_map.mean() + 0.1 * out.coarse_normals.pow(2).mean() + \
                0.05 * out.directional_derivtives.mean()
            loss.backward()
            got[fold] = (out, {k: p.grad.detach().clone() for net in (model.vector_field_network, model.rendering_network)
                               for k, p in net.named_parameters(prefix=type(net).__name__) if p.grad is not None},
                         {k: v.detach().clone() for k, v in model.vector_field_network.state_dict().items() if "running" in k})
    finally:
        batchstat.FOLD_ACTIVATIONS = keep
    (o1, g1, r1), (o0, g0, r0) = got[True], got[False]
    for f in ("coarse_rgb_values", "coarse_depth_map", "coarse_normals", "directional_derivtives", "coarse_colors"):
        assert torch.equal(getattr(o1, f), getattr(o0, f)), f
    assert g1.keys() == g0.keys() and len(g1) > 40
    for k in g1:
        assert torch.equal(g1[k], g0[k]), k
    for k in r1:
        assert torch.equal(r1[k], r0[k]), k
