"""The f16x3 (split-half, fp32-accumulate) inference kernels against the reference's golden vectors, and the
exact-fp32 kernels as a second opinion.  Same tolerances as the fp32 path: the split carries 22 significant bits
per operand and the products of halves are exact in fp32."""
import pytest
import torch

from helpers import FIXTURE_NAMES, build_model, load_fixture, rel_err

pytestmark = pytest.mark.gpu
TIGHT = 2e-5


@pytest.fixture(scope="module", params=FIXTURE_NAMES)
def case(request):
    fx, d = load_fixture(request.param)
    model = build_model(fx, d, device="cuda:0")
    return fx, d, {k: v.to("cuda:0") for k, v in d.items()}, model


def test_vector_only_f16x3(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    pts = (g["cam_loc"][:, None, :] + g["z_coarse"][:, :, None] * g["directions"][:, None, :]).reshape(-1, 3).contiguous()
    vf = model.vector_field_network
    out = lib.vf_mlp16_fwd(vf.geometry(), vf.packed16_weights(), pts)
    ref32 = lib.vf_mlp_fwd(vf.geometry(), vf.packed_weights(), pts, 3)
    e_gold, e_32 = rel_err(out, d["normals_coarse"].reshape(-1, 3)), rel_err(out, ref32)
    print(f"f16x3 vector-only: vs reference {e_gold:.3e}, vs exact-fp32 kernel {e_32:.3e}")
    assert e_gold < TIGHT and e_32 < TIGHT


def test_fused_fine_pass_f16x3(case):
    from vf_nerf_amd import lib
    fx, d, g, model = case
    vf, rn = model.vector_field_network, model.rendering_network
    s_t = fx["n_samples"] + fx["n_importance"]
    normals, colors = lib.vf_render_fused16_fwd(vf.geometry(), vf.packed16_weights(), rn.geometry(),
                                                rn.packed16_weights(), g["points"].reshape(-1, 3).contiguous(),
                                                g["ray_dirs"].contiguous(), s_t)
    en, ec = rel_err(normals, d["normals"].reshape(-1, 3)), rel_err(colors, d["colors"])
    print(f"f16x3 fused: normals {en:.3e} colors {ec:.3e}")
    assert en < TIGHT and ec < TIGHT


def test_precision_switch_end_to_end(case):
    fx, d, g, model = case
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    outs = {}
    for prec in ("fp32", "f16x3"):
        model.precision = prec
        with torch.no_grad():
            outs[prec] = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    model.precision = "f16x3"
    same = (outs["f16x3"].z_vals == outs["fp32"].z_vals).all(dim=1)
    assert float(same.float().mean()) >= 0.9
    err = (outs["f16x3"].coarse_rgb_values - outs["fp32"].coarse_rgb_values).abs().max(dim=1)[0]
    print(f"f16x3 vs fp32 render: identical sampling on {float(same.float().mean()):.3f} of rays, max rgb diff {float(err[same].max()):.3e}")
    assert float(err[same].max()) < 1e-4


def test_partial_tile_and_large_batch():
    """Point counts that are not multiples of the 128-point workgroup, and a batch spanning many workgroups."""
    from vf_nerf_amd import lib
    fx, d = load_fixture("w1_det")
    model = build_model(fx, d, device="cuda:0")
    vf = model.vector_field_network
    gen = torch.Generator().manual_seed(1)
    for m in (1, 31, 129, 5000):
        pts = (torch.rand(m, 3, generator=gen) * 2 - 1).to("cuda:0")
        a = lib.vf_mlp16_fwd(vf.geometry(), vf.packed16_weights(), pts)
        b = lib.vf_mlp_fwd(vf.geometry(), vf.packed_weights(), pts, 3)
        assert rel_err(a, b) < TIGHT, m


def test_proposal_reuse_equals_fused_pass():
    """render() with the VF net evaluated once per distinct sample (vfn_vf_feat16_fwd on the proposal samples, again on the
    N_f new ones, vfn_render16_from_blocks over the stored rows, scattering to the sorted positions) against the single fused launch over all S_c+N_f
    samples: same per-sample arithmetic, so every output must agree to the last bit — including ragged ray counts, the
    all-zero-weights branch (argmax 0 -> uniform extra samples) and per-ray far."""
    import torch
    from helpers import build_model, load_fixture
    for name in ("c1_perturb", "odd_orbit", "w1_det"):
        fx, d = load_fixture(name)
        model = build_model(fx, d, device="cuda:0")
        g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
        uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
        outs = []
        for reuse in (True, False):
            model.reuse_proposal = reuse
            with torch.no_grad():
                outs.append(model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni))
        a, b = outs
        assert torch.equal(a.z_vals, b.z_vals) and torch.equal(a.points_coarse, b.points_coarse)
        for f in ("coarse_normals", "coarse_colors", "coarse_rgb_values", "coarse_depth_map"):
            x, y = getattr(a, f), getattr(b, f)
            assert torch.equal(x, y), (name, f, float((x - y).abs().max()))


def test_fine_sampler_provenance_index():
    """vfn_range_fine_sample_indexed: src maps every sorted sample back to its proposal / new sample, new_points are the
    new samples in generation order; z and points equal the plain sampler's."""
    import torch
    from vf_nerf_amd import lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(4)
    n, s_c, n_f = 37, 16, 9
    z_c = torch.sort(torch.rand(n, s_c, generator=gen), dim=1)[0].to(dev)
    imax = torch.randint(0, s_c, (n,), generator=gen).to(dev)
    imax[::5] = 0
    dirs = torch.randn(n, 3, generator=gen).to(dev)
    cam = torch.randn(n, 3, generator=gen).to(dev)
    u_add, u_fine = torch.rand(n, n_f, generator=gen).to(dev), torch.rand(n, n_f, generator=gen).to(dev)
    z0, p0 = lib.range_fine_sample(z_c, imax, dirs, cam, n_f, 0.0, 1.0, 0.3, u_add, u_fine)
    z1, p1, src, newp = lib.range_fine_sample_indexed(z_c, imax, dirs, cam, n_f, 0.0, 1.0, 0.3, u_add, u_fine)
    assert torch.equal(z0, z1) and torch.equal(p0, p1)
    allp = torch.cat([(cam[:, None, :] + z_c[..., None] * dirs[:, None, :]).reshape(-1, 3), newp.reshape(-1, 3)])
    assert torch.equal(allp[src.reshape(-1).long()].view(n, s_c + n_f, 3), p1)
    rows = torch.sort(src, dim=1)[0].cpu()
    expect = torch.cat([torch.arange(n)[:, None] * s_c + torch.arange(s_c)[None, :],
                        n * s_c + torch.arange(n)[:, None] * n_f + torch.arange(n_f)[None, :]], dim=1).int()
    assert torch.equal(rows, expect)
    # the inverse map, with the new samples' rows moved to a 32-row group boundary (as render() stores them): padding
    # rows stay -1, every other row names the sorted position it went to
    row0 = lib.block_rows(n * s_c)
    assert row0 > n * s_c                      # 37 * 16 = 592 -> 608: the layout has padding rows in this case
    z2, p2, src2, newp2, dst = lib.range_fine_sample_indexed(z_c, imax, dirs, cam, n_f, 0.0, 1.0, 0.3, u_add, u_fine,
                                                             new_row0=row0, want_dst=True)
    assert torch.equal(z2, z1) and torch.equal(newp2, newp)
    assert dst.shape[0] == row0 + n * n_f and bool((dst[n * s_c:row0] == -1).all())
    assert torch.equal(torch.where(src >= n * s_c, src + (row0 - n * s_c), src), src2)
    assert torch.equal(dst[src2.reshape(-1).long()].cpu(), torch.arange(n * (s_c + n_f), dtype=torch.int32))


def test_block_launches_equal_fused_launch():
    """vfn_vf_feat16_fwd (VF net, feature operand blocks out) + vfn_render16_from_blocks (rendering net over the stored rows,
    outputs scattered through an index) against the fused launch on the same samples: bit-identical normals and colours, for
    a row count with padding rows in the last 32-row group and a non-trivial permutation; and vfn_scatter_rows3."""
    import torch
    from helpers import build_model, load_fixture
    from vf_nerf_amd import lib
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device="cuda:0")
    vf, rn = model.vector_field_network, model.rendering_network
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(9)
    n_rays, s = 13, 23                                    # 299 samples: 9 whole groups + 11 rows
    m = n_rays * s
    pts = (torch.rand(m, 3, generator=gen) * 2 - 1).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=gen), dim=1).to(dev)
    want_n, want_c = lib.vf_render_fused16_fwd(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, s)
    # store the samples in a permuted order (rays stay together so that ray_dirs[dst // s] is the sample's direction)
    perm = torch.cat([r * s + torch.randperm(s, generator=gen) for r in range(n_rays)]).to(dev)     # stored row -> sample
    stored = pts[perm].contiguous()
    vecs = torch.empty(m, 3, device=dev)
    blocks = torch.empty(lib.block_rows(m), lib.BLOCK_BYTES, dtype=torch.uint8, device=dev)
    lib.vf_feat16_fwd(vf.geometry(), vf.packed16_weights(), stored, vecs, blocks)
    got_n, got_c = lib.render16_from_blocks(rn.geometry(), rn.packed16_weights(), blocks, vecs, perm.to(torch.int32), pts, dirs, s)
    assert torch.equal(got_n, want_n) and torch.equal(got_c, want_c)
    # the scatter launch and the row scatter against plain indexing
    out_n, out_c = torch.zeros(m, 3, device=dev), torch.zeros(m, 3, device=dev)
    idx = perm.to(torch.int32).clone()
    idx[::7] = -1
    lib.vf_render_fused16_scatter(vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), stored,
                                  dirs[perm // s].contiguous(), 1, idx, out_n, out_c)
    keep = idx >= 0
    assert torch.equal(out_n[perm[keep]], want_n[perm[keep]]) and torch.equal(out_c[perm[keep]], want_c[perm[keep]])
    assert float(out_n[perm[~keep]].abs().max()) == 0.0
    a, b = torch.randn(m, 3, device=dev), torch.randn(m, 3, device=dev)
    oa, ob = torch.zeros(m, 3, device=dev), torch.zeros(m, 3, device=dev)
    lib.scatter_rows3(a, b, idx, oa, ob)
    assert torch.equal(oa[perm[keep]], a[keep]) and torch.equal(ob[perm[keep]], b[keep]) and float(oa[perm[~keep]].abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# range guard (vf_nerf_amd/guard.py): networks / inputs outside the range the split-f16 operands cover
# ------------------------------------------------------------------------------------------------
def _out_of_family(case_name, model, points):
    """Push the fixture's network / inputs out of the default-initialised family."""
    vf, rn = model.vector_field_network, model.rendering_network
    with torch.no_grad():
        if case_name == "gain8_gamma30":            # hidden gain 2 -> 8, BatchNorm gamma spread over 1..30
            for net in (vf, rn):
                for i in range(net.num_layers - 1):
                    net._linear(i).weight.mul_(4.0)
                    bn = net._bn(i)
                    bn.weight.copy_(torch.linspace(1.0, 30.0, bn.weight.numel(), device=bn.weight.device))
        elif case_name == "weights_1e-3":           # every Linear weight scaled down: folded weights deep in the f16 denormals
            for net in (vf, rn):
                for i in range(net.num_layers):
                    net._linear(i).weight.mul_(1e-3)
        elif case_name == "points_50":              # a scene 50 units across (the reference normalises nothing)
            points = points * 50.0
        elif case_name == "points_2000":            # beyond the f16 range once scaled by 2^6
            points = points * 2000.0
    return points


@pytest.mark.parametrize("case_name", ["in_family", "gain8_gamma30", "weights_1e-3", "points_50", "points_2000"])
def test_range_guard_never_returns_unflagged_garbage(case_name):
    """the reference's MLPs have no range restriction (vector_field_network.py:177-208); the f16x3 kernels
    do (|activation|, |coordinate| < ~937, folded weights not down in the f16 denormals).  For each out-of-family case:
    * strict guard: the vector-field query and render() return what the exact-fp32 kernels / the oracle return (the flagged
      call is repeated on the fp32 kernels before it returns);
    * lazy guard: whenever the f16x3 result is more than 1e-4 off, the guard's report says so (and names the reason);
    * in-family inputs never trip it."""
    import warnings
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import lib
    fx, d = load_fixture("c1_perturb")
    dev = "cuda:0"
    g = {k: v.to(dev) for k, v in d.items()}

    def fresh():
        m = build_model(fx, d, device=dev)
        pts = _out_of_family(case_name, m, g["points"].reshape(-1, 3).contiguous())
        return m, pts.contiguous()

    model, pts = fresh()
    cpu_sd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    want = O.vf_mlp(pts.cpu(), cpu_sd, 6, (4,))[:, :3]
    vf = model.vector_field_network
    raw16 = lib.vf_mlp16_fwd(vf.geometry(), vf.packed16_weights(), pts)           # the unguarded kernel
    raw32 = lib.vf_mlp_fwd(vf.geometry(), vf.packed_weights(), pts, 3)
    e16, e32 = rel_err(raw16, want), rel_err(raw32, want)
    print(f"{case_name}: f16x3 kernel vs oracle {e16:.2e}; exact-fp32 kernel vs oracle {e32:.2e}")
    assert e32 < 1e-4, "the exact-fp32 kernels follow the reference everywhere"

    # lazy: the report exists whenever the result is off
    model.f16x3_guard = "lazy"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            out = vf(pts, vector_only=True)
            reason = model.range_guard.check_now(torch.device(dev))
    print(f"{case_name}: lazy guard report: {reason}")
    assert rel_err(out, want) < 1e-4 or reason is not None
    if case_name == "in_family":
        assert reason is None and model.precision == "f16x3" and e16 < TIGHT
    if reason is not None:
        assert model.precision == "fp32" and model.f16x3_disabled == reason and not model.uses_f16x3()
        with torch.no_grad():
            assert rel_err(vf(pts, vector_only=True), want) < 1e-4, "after the switch the model runs the fp32 kernels"

    # strict: no call returns values the clamp touched
    model, pts = fresh()
    model.f16x3_guard = "strict"
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        with torch.no_grad():
            out = model.vector_field_network(pts, vector_only=True)
    assert rel_err(out, want) < 1e-4, (case_name, rel_err(out, want))
    if e16 >= 1e-4:
        assert any("f16x3" in str(w.message) for w in caught) and model.f16x3_disabled is not None

    # strict, through render(): same draws on both paths, normals / colours of identically sampled rays against the oracle
    model, _ = fresh()
    model.f16x3_guard = "strict"
    scale = {"points_50": 50.0, "points_2000": 2000.0}.get(case_name, 1.0)
    pose = g["pose"].clone()

    model.ray_sampler.far = model.fine_sampler.far = fx["far"] * scale
    model.fine_sampler.range = fx["fine_range"] * scale
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():
            out = model.render(pose, g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    settings = O.RenderSettings(n_samples=fx["n_samples"], n_fine=fx["n_importance"], near=fx["near"], far=fx["far"] * scale,
                                fine_range=fx["fine_range"] * scale, perturb=True, n_window=fx["n_window"], dir_to_normal_th=fx["th"],
                                density=O.DensityParams(beta=0.5, mean=0.7, scale=100.0, beta_bounds=(1e-4, 1e9), mean_bounds=(0.6, 1.0), scale_min=1.0))
    rn_sd = {k: v.detach().cpu() for k, v in model.rendering_network.state_dict().items()}
    vf_sd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    ref = O.render(d["uv"], d["pose"], d["intrinsics"], vf_sd, rn_sd, settings, **{k: d[k] for k in uni})
    same = (out.z_vals.cpu() == ref["z_vals"]).all(dim=1)
    assert torch.isfinite(out.coarse_rgb_values).all() and torch.isfinite(out.coarse_normals).all()
    n_same = int(same.sum())
    print(f"{case_name}: strict render(): {n_same}/{same.numel()} rays sampled identically; model now on {model.precision}")
    if n_same:
        en = float((out.coarse_normals.cpu()[same] - ref["normals"][same]).abs().max())
        ec = float((out.coarse_colors.cpu().reshape(ref["normals"].shape)[same] - ref["colors"].reshape(ref["normals"].shape)[same]).abs().max())
        assert en < 1e-4 and ec < 1e-4, (case_name, en, ec)
    if case_name == "in_family":
        assert n_same == same.numel() and model.precision == "f16x3"


def test_range_guard_costs_nothing_on_the_hot_path():
    """lazy mode adds no synchronisation: render() under the guard returns the same values as with the guard off, and the
    status word is read back at most once per guard.LAZY_EVERY calls."""
    from vf_nerf_amd import guard as vguard
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d, device="cuda:0")
    g = {k: v.to("cuda:0") for k, v in d.items()}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}
    outs = {}
    for mode in ("off", "lazy", "strict"):
        model.f16x3_guard = mode
        with torch.no_grad():
            outs[mode] = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni).coarse_rgb_values.clone()
    assert torch.equal(outs["off"], outs["lazy"]) and torch.equal(outs["off"], outs["strict"])
    model.f16x3_guard = "lazy"
    reads = 0
    real = model.range_guard._read_back

    def counting(st, dev):
        nonlocal reads
        reads += 1
        return real(st, dev)

    model.range_guard._read_back = counting
    with torch.no_grad():
        for _ in range(2 * vguard.LAZY_EVERY):
            model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    torch.cuda.synchronize()
    assert reads <= 3 and model.f16x3_disabled is None


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit", "c1_det", "shipped_sizes", "bench_sizes"])
def test_one_call_render_equals_the_launch_by_launch_path(name):
    """vfn_render_fwd (csrc/vfn_render.hip: the whole gradient-free render() issued from C out of one workspace) against the
    facade's launch-by-launch path: every output bit-identical — with the reference's draws replayed, with the device Philox
    stream (same seed and offset: the draws and the advance of the stream must match), with per-ray far values, a shared
    [4,4] pose, the white background — and against the reference's golden outputs."""
    fx, d = load_fixture(name)
    model = build_model(fx, d, device="cuda:0")
    g = {k: v.to("cuda:0") for k, v in d.items()}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    fields = ("points_coarse", "coarse_normals", "coarse_rgb_values", "coarse_depth_map", "z_vals", "ray_dirs", "coarse_colors")

    def both(**kw):
        outs = []
        # two halves on two streams inside the call (default) | five merged launches on one stream | Python | eight launches from C
        for one_call, separate, streams in ((True, False, 2), (True, False, 1), (False, False, 1), (True, True, 1)):
            model.one_call_render, model.render_separate_launches, model.render_streams = one_call, separate, streams
            model.rng_seed, model._rng_offset = 17, 5
            with torch.no_grad():
                outs.append((model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, **kw), model._rng_offset))
        (a, off_a) = outs[0]
        model.render_separate_launches, model.render_streams = False, 0
        for (b, off_b) in outs[1:]:
            assert off_a == off_b
            for f in fields:
                assert torch.equal(getattr(a, f), getattr(b, f)), (name, f, kw.keys())
        return a

    out = both(uniforms=uni)
    # (bench_sizes: 4e-5 in rgb with any kernel — the weights' conditioning, DESIGN.md section 4 — inside the 1e-4 contract)
    assert torch.equal(out.z_vals.cpu(), d["z_vals"]) and rel_err(out.coarse_rgb_values, d["rgb"]) < (1e-4 if name == "bench_sizes" else TIGHT)
    both()                                           # device Philox draws
    both(uniforms=uni, white=True)
    if name == "c1_det":                             # one pose / intrinsics for the whole batch
        model.one_call_render = True
        with torch.no_grad():
            shared = model.render(g["pose"][0], g["uv"], g["intrinsics"][0], epoch=0, uniforms=uni)
        assert torch.equal(shared.coarse_rgb_values, out.coarse_rgb_values)


# ------------------------------------------------------------------------------------------------
# colour branch on two products (csrc/vfn_mlp16.hip, M16_C2; the facade's default for gradient-free renders)
# ------------------------------------------------------------------------------------------------
def vguard_tol() -> float:
    from vf_nerf_amd import guard
    return guard.COLOUR_CHECK_TOL


COLOUR2_TOL = 4e-5          # colours of the two-product branch against the reference's outputs (measured 1.6e-5 .. 2.2e-5); contract 1e-4


@pytest.mark.parametrize("name", FIXTURE_NAMES + ("attached_normals",))
def test_two_product_colour_branch(name):
    """vfn_vf_render_fused16_products with colour_products = 2: the vector head and everything before it keep three products, so
    the normals are the 3-product launch's to the last bit; the colours stay within COLOUR2_TOL of the reference's golden
    outputs (3 products: 2e-7) and of the exact-fp32 kernels; the scattering variant writes the same values to the rows it is
    told to; render() with the two settings agrees bit for bit on everything but the colours / rgb."""
    from vf_nerf_amd import lib
    fx, d = load_fixture(name)
    model = build_model(fx, d, device="cuda:0")
    g = {k: v.to("cuda:0") for k, v in d.items()}
    vf, rn = model.vector_field_network, model.rendering_network
    s_t = fx["n_samples"] + fx["n_importance"]
    pts, dirs = g["points"].reshape(-1, 3).contiguous(), g["ray_dirs"].contiguous()
    args = (vf.geometry(), vf.packed16_weights(), rn.geometry(), rn.packed16_weights(), pts, dirs, s_t)
    n3, c3 = lib.vf_render_fused16_fwd(*args, colour_products=3)
    n2, c2 = lib.vf_render_fused16_fwd(*args, colour_products=2)
    _, c32, _ = lib.vf_render_fused_fwd(vf.geometry(), vf.packed_weights(), rn.geometry(), rn.packed_weights(), pts, dirs, s_t)
    assert torch.equal(n2, n3)
    e3, e2, e2_32 = rel_err(c3, d["colors"]), rel_err(c2, d["colors"]), rel_err(c2, c32)
    print(f"{name}: colours vs the reference: 3 products {e3:.2e}, 2 products {e2:.2e}; 2 products vs exact-fp32 kernels {e2_32:.2e}")
    # trained_far is the state the opt-in does NOT survive (8 000 optimizer steps of the reference trainer: a saturating rendering
    # net, pre-activations of 10-90; tests/golden/make_trained_golden.py --far): there the two-product colours are asserted to be
    # OUTSIDE the guard's tolerance — the finding that made three products the default — and still a rounding-level effect
    far = name == "trained_far"
    assert e3 < 2e-6
    if far:
        assert vguard_tol() < e2_32 < 5e-3 and e2 < 5e-3, (e2, e2_32)
    else:
        assert e2 < COLOUR2_TOL and e2_32 < COLOUR2_TOL
    # scatter: a permutation with dropped rows
    m = pts.shape[0]
    perm = torch.randperm(m, generator=torch.Generator().manual_seed(3)).to(torch.int32)
    perm[::7] = -1
    out_n = torch.full((m, 3), 7.0, device="cuda:0")
    out_c = torch.full((m, 3), 7.0, device="cuda:0")
    lib.vf_render_fused16_scatter(*args[:6], s_t, perm.to("cuda:0"), out_n, out_c, colour_products=2)
    keep = (perm >= 0).to("cuda:0")
    idx = perm.to("cuda:0").long()[keep]
    assert torch.equal(out_n[idx], n2[keep]) and torch.equal(out_c[idx], c2[keep])
    untouched = torch.ones(m, dtype=torch.bool, device="cuda:0")
    untouched[idx] = False
    assert bool((out_c[untouched] == 7.0).all())
    # render(): products only move the colours
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    outs = {}
    for k in (3, 2):
        model.colour_products = k
        for one_call in (True, False):
            model.one_call_render = one_call
            with torch.no_grad():
                outs[k, one_call] = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    for k in (3, 2):
        for f in ("z_vals", "coarse_normals", "coarse_colors", "coarse_rgb_values", "coarse_depth_map", "coarse_weights"):
            if hasattr(outs[k, True], f):
                assert torch.equal(getattr(outs[k, True], f), getattr(outs[k, False], f)), (k, f)
    a, b = outs[3, True], outs[2, True]
    for f in ("z_vals", "points_coarse", "coarse_normals", "coarse_depth_map"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    # what the colour branch's two products add to the composite: against the three-product render (same weights, bit for bit);
    # against the reference both renders carry the weights' own error (the density amplifies the 6e-6 of the normals: 4e-5 in
    # rgb on the 64 + 64-sample fixture with ANY kernel, the exact-fp32 ones included) and stay inside the 1e-4 contract
    e_23 = rel_err(b.coarse_rgb_values, a.coarse_rgb_values)
    e_rgb, e_rgb3 = rel_err(b.coarse_rgb_values, d["rgb"]), rel_err(a.coarse_rgb_values, d["rgb"])
    print(f"{name}: composited rgb, two vs three products {e_23:.2e}; vs the reference: two {e_rgb:.2e}, three {e_rgb3:.2e}")
    assert torch.equal(b.z_vals.cpu(), d["z_vals"]) and e_rgb3 < 1e-4
    if far:
        assert e_23 < 2e-3          # (composited: the weights average the colour error down; reported above, not within the 4e-5 of in-family nets)
    else:
        assert e_23 < COLOUR2_TOL and e_rgb < 1e-4 and e_rgb < e_rgb3 + COLOUR2_TOL


def test_two_product_colours_are_measured_by_the_guard(monkeypatch):
    """guard.py, colour self-check: on the calls whose status is read back, a few rays are evaluated again with three products;
    a difference above guard.COLOUR_CHECK_TOL moves the model back to colour_products = 3 — in strict mode before the call
    returns.  In-family networks stay on two products; with the tolerance lowered under the measured 2e-5 the switch happens and
    the strict call returns exactly the three-product render; a rendering net pushed out of the family (hidden weights x 16,
    BatchNorm gamma up to 30) is caught at the shipped tolerance or is within it."""
    import warnings
    from vf_nerf_amd import guard as vguard
    fx, d = load_fixture("c1_perturb")
    g = {k: v.to("cuda:0") for k, v in d.items()}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add")}

    def render(model):
        with torch.no_grad():
            return model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)

    model = build_model(fx, d, device="cuda:0")
    assert model.colour_products == 3          # the default since round 4: two products are an opt-in (they do not survive training)
    model.colour_products = 2
    model.f16x3_guard = "strict"
    two = render(model)
    assert model.colour_products == 2 and model.range_guard.colour_products_reason is None
    model.colour_products = 3
    three = render(model)
    diff = float((two.coarse_colors - three.coarse_colors).abs().max())
    print(f"in family: two- vs three-product colours {diff:.2e} (tolerance {vguard.COLOUR_CHECK_TOL:.0e})")
    assert 0 < diff < vguard.COLOUR_CHECK_TOL

    monkeypatch.setattr(vguard, "COLOUR_CHECK_TOL", 1e-6)
    for mode in ("strict", "lazy"):
        model = build_model(fx, d, device="cuda:0")
        model.colour_products = 2
        model.f16x3_guard = mode
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            out = render(model)
            if mode == "lazy":                     # the report arrives asynchronously; the switch happens when it is consumed
                assert model.range_guard.check_now(torch.device("cuda:0")) is None
                out = render(model)
        assert model.colour_products == 3 and model.precision == "f16x3" and model.f16x3_disabled is None
        assert any("colour_products" in str(w.message) for w in caught)
        assert torch.equal(out.coarse_colors, three.coarse_colors) and torch.equal(out.coarse_rgb_values, three.coarse_rgb_values)
    monkeypatch.undo()

    # a rendering net out of the family: hidden weights x 2 and x 4 (fresh BatchNorm statistics normalise nothing, so the
    # activations grow 16- and 256-fold by the head and the rounding errors of the weights with them).  Measured unguarded:
    # 2.4e-4 and 3.2e-3 — outside the contract — and the self-check sends both back to three products.
    for wmul in (2.0, 4.0):
        model = build_model(fx, d, device="cuda:0")
        rn = model.rendering_network
        with torch.no_grad():
            for i in range(rn.num_layers - 1):
                rn._linear(i).weight.mul_(wmul)
        rn._invalidate_packs()
        model.colour_products = 2
        model.f16x3_guard = "off"
        model.precision = "fp32"
        want = render(model)
        model.precision = "f16x3"
        raw = render(model)
        assert torch.equal(raw.z_vals, want.z_vals)
        raw_diff = float((raw.coarse_colors - want.coarse_colors).abs().max())
        model.f16x3_guard = "strict"
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = render(model)
        got_diff = float((got.coarse_colors - want.coarse_colors).abs().max())
        print(f"rendering-net weights x{wmul:g}: unguarded two-product colours {raw_diff:.2e} off the exact-fp32 kernels; the strict guard "
              f"returns {got_diff:.2e} (colour_products now {model.colour_products}, precision {model.precision})")
        assert raw_diff > 1e-4, "the case is meant to be outside the contract without the guard"
        assert got_diff < 1e-4 and model.colour_products == 3 and model.precision == "f16x3"
        assert model.range_guard.colour_products_reason is not None


@pytest.mark.parametrize("name", ["c1_perturb", "odd_orbit", "bench_sizes", "trained_256", "trained_far"])
def test_sparse_colours_render_equals_the_dense_render(name):
    """model.sparse_colours (vfn_render_params.sparse_colours; what evaluator.render_view asks for): the vector-field net on every sample
    with its vector-only launch, the fused VF + rendering launch only on the compacted list of samples whose weight is non-zero (the
    count never leaves the device).  Against the dense plan on the same draws: sample depths, points, normals, rgb, depth BIT-identical;
    colours bit-identical wherever they were evaluated, zero elsewhere (completed by a dense launch when the field is READ) — and every
    sample they were NOT evaluated for carries zero weight (checked through the composite: rgb is unchanged to the bit)."""
    import os
    from helpers import GOLDEN_DIR
    if not os.path.exists(os.path.join(GOLDEN_DIR, f"{name}.npz")):
        pytest.skip(f"tests/golden/{name}.npz has not been generated")
    fx, d = load_fixture(name)
    g = {k: v.to("cuda:0") for k, v in d.items() if isinstance(v, torch.Tensor)}
    uni = {k: g[k] for k in ("u_coarse", "u_fine", "u_add") if k in g}
    model = build_model(fx, d, device="cuda:0")
    outs = {}
    for sparse in (False, True):
        model.sparse_colours = sparse
        with torch.no_grad():
            outs[sparse] = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0, uniforms=uni)
    a, b = outs[False], outs[True]
    for f in ("z_vals", "points_coarse", "coarse_normals", "coarse_rgb_values", "coarse_depth_map"):
        assert torch.equal(getattr(a, f), getattr(b, f)), f
    from vf_nerf_amd.render_output import LazyColours
    lazy = object.__getattribute__(b, "coarse_colors")            # (reading the field would complete it: render_output.LazyColours)
    assert isinstance(lazy, LazyColours)
    evaluated = (lazy.sparse != 0).any(dim=1)
    assert torch.equal(lazy.sparse[evaluated], a.coarse_colors[evaluated])
    # a READER of the field gets every sample's colour, as the reference returns it (vector_field_nerf.py:338): the dense render's, to the bit
    assert torch.equal(b.coarse_colors, a.coarse_colors) and type(b.coarse_colors) is torch.Tensor
    frac = float(evaluated.float().mean())
    print(f"{name}: colours evaluated for {frac:.3f} of the {evaluated.numel()} samples; rgb / depth / normals / depths bit-identical to the dense render")
    assert 0.0 < frac < 0.6
    assert torch.equal(b.z_vals.cpu(), d["z_vals"]) and rel_err(b.coarse_rgb_values, d["rgb"]) < 1e-4
    # without supplied draws (the device Philox stream generates them inside the kernels): the same stream position, the same image
    model.rng_seed = 9
    for sparse in (False, True):
        model.sparse_colours, model._rng_offset = sparse, 0
        with torch.no_grad():
            outs[sparse] = model.render(g["pose"], g["uv"], g["intrinsics"], epoch=0)
    assert torch.equal(outs[False].coarse_rgb_values, outs[True].coarse_rgb_values) and torch.equal(outs[False].z_vals, outs[True].z_vals)
    assert model._rng_offset > 0
