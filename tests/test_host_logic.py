"""Host-side logic that needs no GPU: C-ABI exports, layer planning, facade plumbing, checkpoint keys,
ray sharding and the gloo world_size-2 gradient all-reduce."""
import os
import re
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import vf_nerf_amd
from vf_nerf_amd import distributed as vdist
from vf_nerf_amd import lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cpu_model(**kw):
    torch.manual_seed(0)
    m = vf_nerf_amd.VectorFieldNerf(vf_nerf_amd.shipped_config(torch.device("cpu"), **kw))
    return m


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "vfn.h")).read()
    declared = set(re.findall(r"\b(vfn_[a-z0-9_]+)\s*\(", header))
    declared -= {"vfn_status"}
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name
    assert handle.vfn_abi_version() == lib.ABI_VERSION == 5


def test_binding_signatures_come_from_the_header():
    """Every export carries restype + argtypes generated from include/vfn.h; a few known prototypes are spelled out here, and the
    binding's struct mirrors have the sizes the library was compiled with."""
    import ctypes as C
    handle = lib.load()
    protos = lib.header_prototypes()
    assert set(protos) == set(lib.EXPORTS)
    for name in lib.EXPORTS:
        fn = getattr(handle, name)
        assert fn.argtypes is not None and len(fn.argtypes) == len(protos[name][1]), name
    assert protos["vfn_fill_uniform"] == ("int", ["float*", "int64_t", "uint64_t", "uint64_t", "void*"])
    assert protos["vfn_uniform_sample"][1][:4] == ["int32_t", "int32_t", "float", "float"]
    assert protos["vfn_render_fwd"][1][0] == "const vfn_render_params*" and len(protos["vfn_render_fwd"][1]) == 25
    assert handle.vfn_last_error.restype is C.c_char_p and handle.vfn_packed_size.restype is C.c_int64
    for i, mirror in enumerate(lib.struct_mirrors()):
        assert handle.vfn_abi_struct_bytes(i) == C.sizeof(mirror), mirror.__name__
    assert handle.vfn_abi_struct_bytes(99) == -1
    assert C.sizeof(lib.RenderParams) % 8 == 0 and lib.RenderParams.timing_events.size == 4 * C.sizeof(C.c_void_p)


def test_reordered_arguments_are_refused_before_the_call():
    """A pointer where a size belongs, a float where an integer belongs, an integer where a pointer belongs: the generated
    argtypes raise at conversion time, nothing is launched."""
    import ctypes as C
    handle = lib.load()
    with pytest.raises(C.ArgumentError):
        handle.vfn_fill_uniform(C.c_void_p(0), C.c_void_p(0), 1, 2, None)            # pointer as the element count
    with pytest.raises(C.ArgumentError):
        handle.vfn_fill_uniform(None, 1.5, 1, 2, None)                               # float as the element count
    with pytest.raises(C.ArgumentError):
        handle.vfn_fill_uniform(C.c_int32(4), 4, 1, 2, None)                         # integer as the output pointer
    with pytest.raises(C.ArgumentError):
        handle.vfn_uniform_sample(1, 1, None, 1.0, None, None, None, None, None, None, None, None)   # pointer as `near`
    assert handle.vfn_fill_uniform(None, C.c_int32(0), 1, 2, None) == 0              # n = 0: accepted, nothing to do (width converted)


def test_packed_size_and_plan_errors():
    m = _cpu_model(n_samples=8, n_importance=8)
    vf, rn = m.vector_field_network, m.rendering_network
    vf_hidden = 8 * 256 * (5 + 32 + 32) + 7 * 256 * 32 + 8 * 256 * 33 + 8 * 256 * 32 * 3 + 8 * 256 * 32
    vf_bias = 8 * 32 * 8 + 7 * 32
    assert lib.packed_size(lib.NET_VF, vf.geometry()) == vf_hidden + vf_bias + 16 * 256 + 16
    rn_hidden = 8 * 256 * 37 + 3 * 8 * 256 * 32
    assert lib.packed_size(lib.NET_RENDER, rn.geometry()) == rn_hidden + 4 * 256 + 16 * 256 + 16
    bad = lib.make_geom(3, 6, -1, 256, [39, 128, 128], [128, 128, 259], [1, 1, 0])
    with pytest.raises(lib.VfnError, match="out_features=128"):
        lib.packed_size(lib.NET_VF, bad)


def test_cpu_tensor_is_refused_without_fallback():
    m = _cpu_model(n_samples=8, n_importance=8)
    m.eval()
    with torch.no_grad(), pytest.raises(lib.VfnError, match="no CPU fallback"):
        m.vector_field_network(torch.zeros(4, 3))


def test_facade_surface_and_checkpoint_roundtrip():
    m = _cpu_model(n_samples=16, n_importance=8)
    assert m.fine_vector_field_network is m.vector_field_network          # alias (Q4)
    assert len(m.parameters()) == 89 and sum(p.numel() for p in m.unique_parameters()) == 805780
    assert sum(p.numel() for p in m.parameters()) == 1337122
    assert m.ray_sampler.N_samples == 16 and m.fine_sampler.N_samples == 8 and m.fine_sampler.max_samples == 100
    m.fine_sampler.N_samples += 5
    assert m.fine_sampler.N_samples == 13
    assert float(m.density.get_beta()) == 0.5 and abs(float(m.density.get_mean()) - 0.7) < 1e-7
    assert float(m.density.get_scale()) == 100.0
    assert set(m.config.cos_sim_weights_dict()) == {f"w_{i}" for i in range(11)}
    keys = list(m.vector_field_network.state_dict())
    assert keys[0] == "layers.0.0.weight" and "layers.3.1.running_mean" in keys
    assert keys[-2:] == ["layers.8.weight", "layers.8.bias"]
    with tempfile.TemporaryDirectory() as d:
        m.save(7, d)
        ck = torch.load(os.path.join(d, "latest.pth"))
        assert set(ck) == {"vf_net", "rendering_net", "density", "epoch", "optimizer", "scheduler", "fine_vf_net"}
        m2 = _cpu_model(n_samples=16, n_importance=8)
        assert m2.load(os.path.join(d, "7.pth")) == 8
    with pytest.raises(ValueError):
        vf_nerf_amd.shipped_config(torch.device("cpu"), anneal="bogus")


def test_render_requires_fine_sampling_and_a_device():
    m = _cpu_model(n_samples=8, n_importance=0)
    z = torch.zeros(2, 4, 4)
    with pytest.raises(ValueError):
        m.render(z, torch.zeros(2, 2), z, 0)
    from vf_nerf_amd.lib import VfnError
    for mode in ("eval", "train"):          # either way the arithmetic is on the device: CPU tensors are refused, loudly
        m = _cpu_model(n_samples=8, n_importance=8)
        getattr(m, mode)()
        assert m.vector_field_network.training == (mode == "train") == m.rendering_network.training
        with pytest.raises(VfnError):
            m.render(z, torch.zeros(2, 2), z, 0)


def test_window_schedule_is_a_normalised_tent():
    m = _cpu_model(n_samples=8, n_importance=8)
    w = m.annealing.get_weights(350, "cpu")
    assert abs(float(w.sum()) - 1) < 1e-6 and int(torch.argmax(w)) == 5 and float(w[0]) < float(w[4])
    assert torch.allclose(m.annealing.get_weights(-1, "cpu"), torch.full((11,), 1 / 11))


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 4096, 8191):
        for world in (1, 2, 3, 8):
            spans = [vdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _ddp_worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    r, w, _ = vdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(rank)                      # different initial weights per rank on purpose
    model = vf_nerf_amd.VectorFieldNerf(vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=8, n_importance=8))
    vdist.broadcast_parameters(model, src=0)
    bucket = vdist.GradientBucket(model)
    assert bucket.numel() == 805780              # the alias must not double the bucket (Q4)
    model.optimizer.zero_grad(set_to_none=True)  # a trainer that drops the views...
    for i, p in enumerate(model.unique_parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    bucket.all_reduce_mean()                     # ...is re-bound, then ONE collective
    expect = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, expect * (i + 1)))
             for i, p in enumerate(model.unique_parameters()))
    first = model.vector_field_network.layers[0][0].weight.detach().clone()
    lo, hi = vdist.shard_bounds(7, rank, world)
    gathered = vdist.gather_rows(torch.full((hi - lo, 2), float(rank)), 7)
    torch.save({"ok": ok, "w": first, "rows": gathered}, os.path.join(result_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_gloo_world2_gradient_allreduce_and_broadcast():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_ddp_worker, args=(world, port, d), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f"r{r}.pt")) for r in range(world)]
    assert all(r["ok"] for r in res)
    assert torch.equal(res[0]["w"], res[1]["w"])          # broadcast made the replicas identical
    rows = res[0]["rows"]
    assert rows.shape[0] == 7 and torch.equal(rows[:, 0], torch.tensor([0., 0, 0, 0, 1, 1, 1]))


def test_dropin_aliases_reference_import_paths():
    import importlib
    import sys
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("models", "evaluation")}
    try:
        import vf_nerf_amd.dropin  # noqa: F401
        mod = importlib.import_module("models.nerf.vector_field_nerf")
        assert mod.VectorFieldNerf is vf_nerf_amd.VectorFieldNerf
        from models.vector_field.vector_field_network import VectorFieldNetwork
        from models.samplers.ray_sampler import RangeFineSampler, UniformSampler  # noqa: F401
        from models.helpers.density_functions import LaplaceDensity  # noqa: F401
        assert VectorFieldNetwork is vf_nerf_amd.networks.VectorFieldNetwork
    finally:
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "evaluation")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_sequential_adam_and_clip_match_torch_sequential_loops():
    """optim.SequentialAdam / optim.clip_grad_norm_ against torch.optim.Adam(foreach=False) /
    torch.nn.utils.clip_grad_norm_(foreach=False) on a parameter list that names some tensors twice (Q4): identical
    parameters, moments and step counters after several steps, identical state_dict layout."""
    from vf_nerf_amd import optim
    torch.manual_seed(3)
    shapes = [(7, 5), (5,), (3, 3), (4,)]

    def make():
        torch.manual_seed(4)
        ps = [torch.nn.Parameter(torch.randn(*s)) for s in shapes]
        return ps, ps + ps[:2]                 # the first two are listed twice

    pa, la = make()
    pb, lb = make()
    oa = torch.optim.Adam(la, lr=3e-3, foreach=False)
    ob = optim.SequentialAdam(lb, lr=3e-3)
    for it in range(5):
        g = [torch.randn(*s) * (10.0 if it == 2 else 0.1) for s in shapes]
        for plist in (pa, pb):
            for p, gg in zip(plist, g):
                p.grad = gg.clone()
        na = torch.nn.utils.clip_grad_norm_(la, 0.5, foreach=False)
        nb = optim.clip_grad_norm_(lb, 0.5)
        assert abs(float(na) - float(nb)) <= 1e-6 * float(na)
        for a, b in zip(pa, pb):
            assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=0)
            b.grad.copy_(a.grad)               # remove the norm's summation-order rounding from the Adam comparison
        oa.step(); ob.step()
        for a, b in zip(pa, pb):
            assert torch.equal(a, b), it
    for a, b in zip(pa, pb):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
    assert float(oa.state[pa[0]]["step"]) == 10.0 and float(oa.state[pa[2]]["step"]) == 5.0
    sda, sdb = oa.state_dict(), ob.state_dict()
    assert sda["param_groups"][0]["params"] == sdb["param_groups"][0]["params"] and sda["state"].keys() == sdb["state"].keys()
    ob.load_state_dict(sda)                    # reference checkpoints load


def test_vfloss_matches_oracle_restatement():
    """vf_nerf_amd.loss.VFLoss (reference interface, one read-back) against the oracle's restatement of
    models/losses/vf_loss.py:34-87 — before and after the norm<1 term switches on, with and without depth / supervision."""
    from types import SimpleNamespace
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd.loss import VFLoss
    gen = torch.Generator().manual_seed(2)
    n, s = 16, 8
    cfg = SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=5, directional_derivatives_start=0)
    wts = SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0)
    crit = VFLoss(cfg, wts)
    ow = O.LossWeights(norm_smaller_than_one_start=5)
    for epoch, with_depth, with_sup in ((0, True, True), (7, True, True), (0, False, False)):
        pred = dict(rgb=torch.rand(n, 3, generator=gen), depth=torch.rand(n, 1, generator=gen),
                    normals=torch.randn(n * s, 3, generator=gen), directional_derivatives=None,
                    supervised_normals=torch.randn(10, 3, generator=gen) if with_sup else torch.empty(0, 3))
        gt = dict(rgb=torch.rand(n, 3, generator=gen), depth=torch.rand(n, 1, generator=gen) * 2 if with_depth else torch.empty(0),
                  supervised_normals=torch.randn(10, 3, generator=gen) if with_sup else torch.empty(0))
        loss, logs = crit(pred, gt, epoch)
        want = O.vf_loss(pred["rgb"], pred["depth"], pred["normals"], pred["supervised_normals"], gt["rgb"], gt["depth"],
                         gt["supervised_normals"], ow, epoch=epoch)
        assert abs(float(loss) - float(want)) < 1e-6 * max(1.0, abs(float(want)))
        assert set(logs) == {"rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss",
                             "directional_derivatives_loss"} and all(isinstance(v, float) for v in logs.values())
        assert (logs["norm_smaller_than_one_loss"] > 0) == (epoch >= 5)


def test_product_vfloss_matches_reference_vfloss_golden():
    """vf_nerf_amd.loss.VFLoss against the outputs of the reference's own VFLoss.forward (tests/golden/trainer_steps.npz,
    models/losses/vf_loss.py:34-87), every branch; and the shape of its log dictionary."""
    from types import SimpleNamespace
    from helpers import load_trainer_fixture
    from vf_nerf_amd.loss import VFLoss
    _, d = load_trainer_fixture()
    base = {k[len("loss.in."):]: v for k, v in d.items() if k.startswith("loss.in.")}
    crit = VFLoss(SimpleNamespace(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                  SimpleNamespace(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1,
                                  directional_derivatives=0.3))
    for name in ("early", "late", "dd_before_start", "no_depth_no_sup"):
        epoch, dd, depth, sup = [int(v) for v in d[f"loss.{name}.case"]]
        pred = {"rgb": base["rgb"], "depth": base["depth"], "normals": base["normals"],
                "supervised_normals": base["sup"] if sup else torch.empty(0, 3), "directional_derivatives": base["dd"] if dd else None}
        gt = {"rgb": base["rgb_gt"], "depth": base["depth_gt"] if depth else torch.empty(0),
              "supervised_normals": base["sup_gt"] if sup else torch.empty(0)}
        loss, logs = crit(pred, gt, epoch)
        assert abs(float(loss) - float(d[f"loss.{name}.total"])) <= 1e-6 * max(1.0, float(d[f"loss.{name}.total"])), name
        got = torch.tensor(list(logs.values()), dtype=torch.float64)
        assert list(logs) == ["rgb_loss", "depth_loss", "unit_norm_loss", "supervision_loss", "norm_smaller_than_one_loss",
                              "directional_derivatives_loss"]
        assert float((got - d[f"loss.{name}.terms"]).abs().max()) <= 1e-6, (name, got)


def test_product_supervision_selection_matches_reference_golden():
    """supervision.get_center_indices_and_gt / get_border_indices_and_gt (the tensor-op halves of the trainer's supervision,
    models/helpers/functions.py:75-98,137-157) against the reference's own outputs."""
    from helpers import load_trainer_fixture
    from vf_nerf_amd import supervision
    fx, d = load_trainer_fixture()
    c = torch.tensor(fx["centroid"])
    for t in range(fx["steps"]):
        n_, g_ = supervision.get_center_indices_and_gt(d[f"s{t}.out.points"], d[f"s{t}.out.normals"], c, fx["border_radius"])
        assert torch.equal(n_, d[f"s{t}.ray_center_normals"]) and torch.equal(g_, d[f"s{t}.ray_center_gt"])
    far, radius = d["border_idx.args"].tolist()
    a, b = supervision.get_border_indices_and_gt(d["border_idx.points"], d["border_idx.normals"], far, radius, d["border_idx.centroid"])
    assert torch.equal(a, d["border_idx.out_normals"]) and torch.equal(b, d["border_idx.out_gt"])


def test_reference_written_checkpoint_loads():
    """tests/golden/ref_checkpoint_latest.pth was written by the REFERENCE's save() (models/nerf/vector_field_nerf.py:196-214)
    after two optimizer steps of a narrow model.  VectorFieldNerf.load must take it as is: same keys, every tensor restored, the
    optimizer's moments and step counters (the aliased VF parameters took two updates per step, Q4), the scheduler, and the
    reference's return value epoch + 1 (:194).  Saving again gives a file with the same layout."""
    from helpers import REF_CHECKPOINT, load_trainer_fixture, narrow_checkpoint_model
    _, d = load_trainer_fixture()
    ck = torch.load(REF_CHECKPOINT, map_location="cpu")
    assert set(ck) == {"vf_net", "rendering_net", "density", "epoch", "optimizer", "scheduler", "fine_vf_net"}
    m, c = narrow_checkpoint_model(d)
    for mine, theirs in ((m.vector_field_network.state_dict(), ck["vf_net"]), (m.rendering_network.state_dict(), ck["rendering_net"]),
                         (m.density.state_dict(), ck["density"])):
        assert list(mine) == list(theirs) and all(mine[k].shape == theirs[k].shape for k in mine)
    assert m.load(REF_CHECKPOINT) == int(d["ckpt.epoch"]) + 1 == c["saved_epoch"] + 1
    for k, v in ck["vf_net"].items():
        assert torch.equal(m.vector_field_network.state_dict()[k], v), k
    for k, v in ck["rendering_net"].items():
        assert torch.equal(m.rendering_network.state_dict()[k], v), k
    assert float(m.density.beta) == float(d["ckpt.beta"]) and float(m.density.scale) == float(d["ckpt.scale"])
    assert abs(m.optimizer.param_groups[0]["lr"] - float(d["ckpt.lr"])) < 1e-15
    assert m.scheduler.state_dict()["last_epoch"] == int(d["ckpt.scheduler_last_epoch"]) == 2
    assert len(m.optimizer.state_dict()["state"]) == int(d["ckpt.n_optimizer_states"])
    st = m.optimizer.state[m.vector_field_network.layers[2][0].weight]
    assert float(st["step"]) == float(d["ckpt.vf_step"]) == 4.0            # two steps x two updates (Q4)
    assert float(m.optimizer.state[m.rendering_network.layers[1][0].weight]["step"]) == 2.0
    assert float(st["exp_avg"].abs().max()) > 0
    with tempfile.TemporaryDirectory() as tmp:
        m.save(c["saved_epoch"], tmp)
        again = torch.load(os.path.join(tmp, "latest.pth"), map_location="cpu")
    assert set(again) == set(ck) and again["epoch"] == ck["epoch"]
    for part in ("vf_net", "rendering_net", "density", "fine_vf_net"):
        assert list(again[part]) == list(ck[part]) and all(torch.equal(again[part][k], ck[part][k]) for k in ck[part])
    assert again["optimizer"]["param_groups"][0]["params"] == ck["optimizer"]["param_groups"][0]["params"]
    assert again["optimizer"]["state"].keys() == ck["optimizer"]["state"].keys()
    assert again["scheduler"]["last_epoch"] == ck["scheduler"]["last_epoch"]


def test_scaled_f16_gradient_slot_round_trip():
    """The host-side encoder / decoder of the scaled f16 gradient form (lib.rows_to_frag_f16s / frag_f16s_to_rows: what the GPU
    tests compare the chain's output with, and feed the weight-gradient kernel): small integers survive exactly, random values to
    11 significant bits per lane-tile at any magnitude, all-zero lanes carry the byte 255, padding points decode to nothing."""
    import torch
    from vf_nerf_amd import lib
    g = torch.Generator().manual_seed(0)
    ints = torch.randint(-3, 4, (77, 256), generator=g).float()
    assert torch.equal(lib.frag_f16s_to_rows(lib.rows_to_frag_f16s(ints), 77), ints)
    for scale in (1e-30, 1e-20, 1.0, 1e20, 1e37):          # (fp32 denormals count as zero: the smallest here is 1e-36)
        dy = torch.randn(1000, 256, generator=g) * torch.logspace(-6, 0, 256)[None, :] * scale
        dy[17] = 0
        slot = lib.rows_to_frag_f16s(dy)
        back = lib.frag_f16s_to_rows(slot, 1000)
        lane_max = dy.reshape(1000, 8, 4, 2, 4).abs().amax(dim=(2, 4), keepdim=True).expand(1000, 8, 4, 2, 4).reshape(1000, 256)
        assert float(((back - dy).abs() / lane_max.clamp_min(1e-45)).max()) < 2 ** -11, scale
        groups = lib.frag_groups(1000)
        exps = slot.view(groups, lib.GROUP_FLOATS).view(torch.uint8)[:, lib.F16S_EXP_OFF:lib.F16S_EXP_OFF + 512].reshape(groups, 8, 2, 32)
        assert bool((exps[17 // 32, :, :, 17 % 32] == 255).all())            # the all-zero point
        assert bool((exps[-1, :, :, 1000 % 32:] == 255).all())               # padding points of the last group
        assert int(exps[exps != 255].max()) <= 239


def test_bench_launches_its_own_ranks_dry_run_gloo_world2():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks as child processes (torch.distributed.run), and rank
    0's single JSON line comes back through the parent.  --backend gloo --dry-run: the plumbing only (process group, parameter
    broadcast, flat gradient bucket all-reduce, barrier / max-over-ranks timing, per-rank rates), no GPU anywhere."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "4",
                           "--warmup", "1"], cwd=tempfile.gettempdir(), env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [l for l in proc.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, proc.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["per_rank_rays_per_s"]["dist_world_size"] == 2 and rec["dry_run"] is True
    assert rec["bucket_elements"] == 805780 and rec["bucket_allreduce_ok"] and rec["replicas_identical_after_broadcast"]
    assert rec["steps"] == 4 and rec["scaling"] == "weak" and rec["per_rank_rays_per_s"]["min"] <= rec["per_rank_rays_per_s"]["max"]
    # rank 1 sleeps twice as long per step as rank 0: the whole-job figure follows the slowest rank
    assert rec["value"] <= 2 * rec["per_rank_rays_per_s"]["min"] * 1.05


def test_bench_refuses_a_world_size_that_is_not_gpus():
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run"], env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0 and "started 1 rank" in (proc.stderr + proc.stdout)


def test_dropin_puts_the_chunked_view_renderer_behind_the_evaluator():
    """evaluation.methods imports open3d / pyrender / trimesh at the top, so it only exists where the reference's environment
    does; with a module of that name present, install() replaces its render_images and nothing else in it."""
    import importlib
    import sys
    import types
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("models", "evaluation")}
    fake_pkg, fake = types.ModuleType("evaluation"), types.ModuleType("evaluation.methods")
    fake_pkg.__path__ = []
    fake.render_images, fake.metrics = (lambda *a, **k: "reference"), (lambda *a, **k: "reference metrics")
    try:
        sys.modules["evaluation"], sys.modules["evaluation.methods"] = fake_pkg, fake
        dropin = importlib.import_module("vf_nerf_amd.dropin")
        dropin.install()
        from vf_nerf_amd import evaluator
        assert fake.render_images is evaluator.render_images and fake.metrics() == "reference metrics"
    finally:
        import vf_nerf_amd.dropin as dropin
        dropin.uninstall_clip_grad_norm()
        for k in [k for k in sys.modules if k.split(".")[0] in ("models", "evaluation")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_lazy_loss_terms_behave_like_the_dict_the_trainer_expects():
    """VFLoss returns {name: float}; the trainer aliases it as its running sums, adds a key, adds step values with += and divides
    in place (train/vector_field_nerf_train.py:262-279)."""
    from types import SimpleNamespace as NS
    from vf_nerf_amd.loss import VFLoss
    crit = VFLoss(NS(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                  NS(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0))
    g = torch.Generator().manual_seed(0)
    pred = {"rgb": torch.rand(4, 3, generator=g), "depth": torch.rand(4, 1, generator=g), "normals": torch.randn(9, 3, generator=g),
            "supervised_normals": torch.randn(3, 3, generator=g), "directional_derivatives": None}
    gt = {"rgb": torch.rand(4, 3, generator=g), "depth": torch.rand(4, 1, generator=g), "supervised_normals": torch.randn(3, 3, generator=g)}
    loss, terms = crit(pred, gt, 0)
    assert isinstance(terms, dict) and list(terms.keys())[0] == "rgb_loss" and len(terms) == 6
    average = terms
    average["loss"] = loss.item()
    _, more = crit(pred, gt, 0)
    for key in more.keys():
        average[key] += more[key]
    for key in average.keys():
        average[key] /= 2
    assert abs(average["rgb_loss"] - float((pred["rgb"] - gt["rgb"]).abs().mean())) < 1e-7
    assert abs(2.0 * average["rgb_loss"] + 0.5 * average["depth_loss"] + 0.1 * average["unit_norm_loss"] + average["supervision_loss"]
               - 2 * average["loss"]) < 1e-5


def test_modules_and_parameters_stay_picklable_next_to_the_flat_optimizer(tmp_path):
    """The optimizer finds its parameters through a registry outside the tensors (optim.owner_of): nothing lands in a
    Parameter's __dict__, so torch.save(module) / torch.save({'p': param}) / pickle keep working after model.optimizer exists."""
    import pickle
    from vf_nerf_amd import optim
    m = _cpu_model(n_samples=8, n_importance=8)
    p = m.vector_field_network.layers[0][0].weight
    assert optim.owner_of(p) is m.optimizer and "_flat_adam" not in p.__dict__
    torch.save(m.rendering_network, tmp_path / "module.pt")
    torch.save({"p": p}, tmp_path / "param.pt")
    again = pickle.loads(pickle.dumps(p))
    assert torch.equal(again, p) and optim.owner_of(again) is None
    back = torch.load(tmp_path / "module.pt", weights_only=False)
    assert torch.equal(back.layers[0][0].weight, m.rendering_network.layers[0][0].weight)


def test_adam_skips_frozen_parameters_like_torch():
    """requires_grad_(False) on a parameter: torch.optim.Adam neither updates it nor counts a step for it (grad is None), with
    weight decay as without.  SequentialAdam / FlatAdam (CPU path here; the device path is tests/test_hip_trainer.py) agree with
    torch.optim.Adam(foreach=False) on a list with a duplicated and a frozen parameter."""
    from vf_nerf_amd import optim
    torch.manual_seed(0)
    a = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2, 2))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    for ps in (a, b):
        ps[1].requires_grad_(False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        oa = optim.FlatAdam(a + a[:1], lr=1e-2, weight_decay=0.1)
        ob = torch.optim.Adam(b + b[:1], lr=1e-2, weight_decay=0.1, foreach=False)
    for _ in range(3):
        for ps, o in ((a, oa), (b, ob)):
            o.zero_grad()
            loss = (ps[0] ** 2).sum() + (ps[2] * 3).sum() + (ps[1] * 0).sum()
            loss.backward()
            o.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, atol=1e-7)
    assert a[1].grad is None and len(oa.state.get(a[1], {})) == 0
    assert float(oa.state[a[0]]["step"]) == 6 and float(oa.state[a[2]]["step"]) == 3


def test_dense_centre_selection_gives_the_compacted_loss_and_gradient():
    """trainer.TrainStep's default: the ray samples inside the centre ball without boolean-mask compaction (no host
    synchronisation): rows outside enter as zeros and the mean's denominator is the selected-row count (VFLoss
    ``supervised_rows``).  Same supervision loss and the same gradient wrt the normals as functions.py:137-157 + VFLoss's mean."""
    from types import SimpleNamespace as NS
    from vf_nerf_amd import supervision
    from vf_nerf_amd.loss import VFLoss
    crit = VFLoss(NS(depth_loss_clamp=0.5, norm_smaller_than_one_start=11000, directional_derivatives_start=100),
                  NS(rgb=2.0, depth=0.5, unit_norm=0.1, supervision=1.0, norm_smaller_than_one=0.1, directional_derivatives=0.0))
    g = torch.Generator().manual_seed(3)
    pts = torch.rand(7, 20, 3, generator=g) * 0.6 + torch.tensor([-0.3, -0.3, 0.25])
    centroid, radius = torch.tensor([0.0, 0.0, 0.55]), 0.15
    extra, extra_gt = torch.randn(11, 3, generator=g), torch.randn(11, 3, generator=g)
    common = {"rgb": torch.rand(7, 3, generator=g), "depth": torch.rand(7, 1, generator=g), "directional_derivatives": None}
    gt_common = {"rgb": torch.rand(7, 3, generator=g), "depth": torch.rand(7, 1, generator=g)}
    results = []
    for dense in (False, True):
        normals = torch.randn(7, 20, 3, generator=torch.Generator().manual_seed(9)).requires_grad_(True)
        if dense:
            rc_n, rc_gt, n_sel = supervision.center_rows_dense(pts, normals, centroid, radius)
            pred = dict(common, normals=normals.reshape(-1, 3), supervised_normals=torch.cat([extra, rc_n]), supervised_rows=n_sel + 11.0)
        else:
            rc_n, rc_gt = supervision.get_center_indices_and_gt(pts, normals, centroid, radius)
            assert 0 < rc_n.shape[0] < 140
            pred = dict(common, normals=normals.reshape(-1, 3), supervised_normals=torch.cat([extra, rc_n]))
        loss, terms = crit(pred, dict(gt_common, supervised_normals=torch.cat([extra_gt, rc_gt])), 0)
        loss.backward()
        results.append((float(loss), terms["supervision_loss"], normals.grad.clone()))
    (la, sa, ga), (lb, sb, gb) = results
    assert abs(la - lb) < 1e-6 and abs(sa - sb) < 1e-6 and float((ga - gb).abs().max()) < 1e-7


def _dp_step_worker(rank, world, port, result_dir, sequence="explicit"):
    """One data-parallel optimizer step on the CPU: this rank's shard of the fixture's rays through the ORACLE's render + autograd
    (there is no CPU backend of the product's kernels; the oracle is the checker's compute here), the gradients written into the
    flat bucket, ONE all-reduce, the clip over the duplicated parameter list, the optimizer's step — everything after the gradients
    is product code (distributed.shard_bounds / GradientBucket / optim.clip_grad_norm_ / FlatAdam's CPU path)."""
    from helpers import build_model, load_fixture, oracle_settings
    from oracle import vfnerf_oracle as O
    from vf_nerf_amd import optim
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        vdist.init_from_env("gloo")
    torch.set_num_threads(2)
    fx, d = load_fixture("c1_perturb")
    model = build_model(fx, d)
    if world > 1:
        vdist.broadcast_parameters(model, src=0)
    bucket = vdist.GradientBucket(model)
    lo, hi = vdist.shard_bounds(d["uv"].shape[0], rank, world)
    pose, uv, K = vdist.shard_rays(d["pose"], d["uv"], d["intrinsics"], rank, world)
    assert uv.shape[0] == hi - lo
    vf_sd = {k: v.detach().clone() for k, v in model.vector_field_network.state_dict().items()}
    rn_sd = {k: v.detach().clone() for k, v in model.rendering_network.state_dict().items()}
    leaves = {}
    for net, sd in ((model.vector_field_network, vf_sd), (model.rendering_network, rn_sd)):
        for name, p in net.named_parameters():
            sd[name].requires_grad_(True)
            leaves[id(p)] = sd[name]
    out = O.render(uv, pose, K, vf_sd, rn_sd, oracle_settings(fx), u_coarse=d["u_coarse"][lo:hi], u_fine=d["u_fine"][lo:hi], u_add=d["u_add"][lo:hi])
    loss = 2.0 * out["rgb"].abs().mean() + 0.5 * out["depth"].mean() + 0.1 * ((out["normals"].reshape(-1, 3).norm(dim=1) - 1) ** 2).mean()
    loss.backward()
    bucket.zero()
    for p in model.unique_parameters():
        if id(p) in leaves and leaves[id(p)].grad is not None:
            p.grad.copy_(leaves[id(p)].grad)
    if sequence == "drop_in":
        # the reference trainer's own lines (train/vector_field_nerf_train.py:254-258): the all-reduce sits inside the wrapped
        # clip_grad_norm_ (vf_nerf_amd.dropin), i.e. still between backward() and the clip
        import vf_nerf_amd.dropin  # noqa: F401
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
    else:
        bucket.all_reduce_mean()
        norm = optim.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
    model.optimizer.step()
    watched = {k: dict(net.named_parameters())[k].detach().clone()
               for net, k in ((model.vector_field_network, "layers.5.0.weight"), (model.vector_field_network, "layers.8.weight"),
                              (model.rendering_network, "layers.2.0.weight"), (model.rendering_network, "layers.4.bias"))}
    torch.save({"loss": float(loss), "norm": float(norm), "w": watched, "rays": hi - lo}, os.path.join(result_dir, f"dp_{sequence}_w{world}_r{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


@pytest.mark.parametrize("sequence", ["explicit", "drop_in"])
def test_two_rank_data_parallel_step_equals_the_single_process_step(sequence):
    """("drop_in": the all-reduce where the reference's unchanged trainer reaches it — inside the wrapped clip_grad_norm_.)
    BASELINE.json configs[3] in small, on the CPU: the rays of a batch sharded over two ranks, one all-reduce (mean) of the flat
    gradient bucket before the clip, then the optimizer's step — against ONE process taking the same step on the whole batch.  With
    equal shards the mean of the shard losses is the batch loss, so the clip norm and the weights after the step must agree (both
    ranks with each other exactly; with the single process up to the order of the sums)."""
    port = 29700 + (os.getpid() % 2000) + (5000 if sequence == "drop_in" else 0)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_dp_step_worker, args=(1, port, tmp, sequence), nprocs=1, join=True)      # (own process: the worker pins torch's thread count)
        mp.spawn(_dp_step_worker, args=(2, port, tmp, sequence), nprocs=2, join=True)
        one = torch.load(os.path.join(tmp, f"dp_{sequence}_w1_r0.pt"))
        two = [torch.load(os.path.join(tmp, f"dp_{sequence}_w2_r{r}.pt")) for r in range(2)]
    assert one["rays"] == 48 and [t["rays"] for t in two] == [24, 24]
    assert abs(0.5 * (two[0]["loss"] + two[1]["loss"]) - one["loss"]) < 1e-6 * max(1.0, abs(one["loss"]))
    assert abs(two[0]["norm"] - two[1]["norm"]) == 0.0 and abs(two[0]["norm"] - one["norm"]) < 1e-4 * one["norm"]
    for k in one["w"]:
        assert torch.equal(two[0]["w"][k], two[1]["w"][k]), k                      # replicas stay identical
        moved = float((one["w"][k] - two[0]["w"][k]).abs().max())
        assert moved < 0.05 * 5e-4, (k, moved)                                     # a twentieth of one Adam update (lr 5e-4)


def test_training_products_knob_and_what_it_selects():
    """``model.training_products``: 3 by default, 1 = the opt-in 16-bit-native mode — shared with the vector-field net (its standalone
    forwards follow it), refused for anything else, and selected by ``backward._storage`` only on the default storages (f16
    activations, fragment order, scaled-f16 gradients); a workspace of that kind asks the kernels for single products."""
    import vf_nerf_amd
    from vf_nerf_amd import backward, lib
    model = vf_nerf_amd.VectorFieldNerf(vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=8, n_importance=8))
    assert model.training_products == 3 and backward._storage(model, True) == (True, True, "f16")
    model.training_products = 1
    assert model.vector_field_network.training_products == 1
    assert backward._storage(model, True) == (True, True, "f16p1") and backward._storage(model.vector_field_network, True) == (True, True, "f16p1")
    assert backward._storage(model, False) == (False, False, None)
    for attr, other in (("activation_storage", "fp32"), ("gradient_storage", "bf16"), ("workspace_layout", "rows")):
        keep = getattr(model, attr)
        setattr(model, attr, other)
        assert backward._storage(model, True)[2] != "f16p1", attr      # any other storage: the mode does not apply, three products run
        setattr(model, attr, keep)
    with pytest.raises(ValueError):
        model.training_products = 2
    ws = backward._Workspace(64, 13, torch.device("cpu"), f16=True, frag=True, dy16="f16p1")
    assert ws.single and ws.dy16 == "f16" and ws.fwd_flags() == (lib.WS_F16 | lib.WS_FRAG | lib.WS_P1) and ws.dy_flags() == (lib.DY_FRAG | lib.DY_F16S | lib.DY_P1)
    ws3 = backward._Workspace(64, 13, torch.device("cpu"), f16=True, frag=True, dy16="f16")
    assert not ws3.single and ws3.fwd_flags() == (lib.WS_F16 | lib.WS_FRAG) and ws3.dy_flags() == (lib.DY_FRAG | lib.DY_F16S)


def test_layer_product_arithmetic_follows_precision_and_the_net_knob():
    """batchstat._arith: split-operand products on the 16-bit matrix cores by default (forward f16x3, the first layer on bf16 in three parts,
    backward three bf16 products — bf16 in three parts with ``gemm_arithmetic = "split24"``); the exact fp32 matrix instruction when the net
    asks for it (``gemm_arithmetic = "fp32"``) or the facade does (``precision = "fp32"`` — what the range
    guard switches a model to); anything else is refused."""
    import types
    from vf_nerf_amd import batchstat, lib
    net = types.SimpleNamespace()
    assert batchstat._arith(net, False) == lib.GEMM_SPLIT_F16 and batchstat._arith(net, True) == lib.GEMM_SPLIT_BF16
    assert batchstat._arith(net, False, first_layer=True) == lib.GEMM_BF16X6
    net.gemm_arithmetic = "split24"
    assert batchstat._arith(net, False) == lib.GEMM_SPLIT_F16 and batchstat._arith(net, True) == lib.GEMM_BF16X6
    del net.gemm_arithmetic
    net.precision = "fp32"
    assert batchstat._arith(net, False) == batchstat._arith(net, True) == lib.GEMM_EXACT
    net.precision, net.gemm_arithmetic = "f16x3", "fp32"
    assert batchstat._arith(net, False) == batchstat._arith(net, True) == lib.GEMM_EXACT
    net.gemm_arithmetic = "tf32"
    with pytest.raises(ValueError):
        batchstat._arith(net, False)
    model = vf_nerf_amd.VectorFieldNerf(vf_nerf_amd.shipped_config(torch.device("cpu"), n_samples=8, n_importance=8))
    model.precision = "fp32"
    assert model.vector_field_network.precision == model.rendering_network.precision == "fp32"


def test_live_traffic_pass_reports_why_it_could_not_run(monkeypatch):
    """bench.live_hbm_traffic: without rocprofv3 on PATH (this container's case on a CPU-only run is simulated here) it returns no figure and the
    reason; bench.hbm_traffic then falls back to the committed PMC file and carries that reason beside the file's provenance."""
    import shutil
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    monkeypatch.setattr(shutil, "which", lambda name: None)
    args = type("A", (), dict(colour_products=None, rays=4096, coarse=64, fine=64, weights="trained"))()
    got = bench.live_hbm_traffic(args)
    assert got[0] is None and got[1]["live"] is False and "rocprofv3" in got[1]["why"]
    monkeypatch.setattr(bench, "_LIVE_TRAFFIC", got)
    total, prov = bench.hbm_traffic(True, "fused16", 3)
    assert total is not None and total > 10485760 and prov["file"].startswith("profiles/") and prov["live_pass"]["live"] is False
    assert bench.under_profiler() in (False, True)


def test_reference_sequence_is_the_reference_trainers_call_sequence():
    """tools/reference_sequence.py (what the GPU tests, tools/host_profile.py and bench.py's ``train.drop_in_sequence`` run as "the
    reference trainer's own call sequence") against the REFERENCE'S OWN ``VectorFieldNerfRunner.train_epoch``: both are run on the same
    recording stand-ins (tests/trace_reference_loop.py, in a subprocess: it imports /root/reference) and must make the same calls with the
    same argument shapes and scalars in the same order — render, sample_border_points, vector_field_network, get_center_indices_and_gt,
    sample_center_points, vector_field_network, the loss, zero_grad, parameters, clip_grad_norm_, optimizer.step, scheduler.step.
    Skipped where the reference is not present (the GPU box)."""
    import json
    import subprocess
    import sys
    if not os.path.isdir("/root/reference"):
        pytest.skip("no /root/reference here")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "trace_reference_loop.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    assert rec["identical"] and rec["calls_reference"] == rec["calls_restatement"] == 24, rec
    assert rec["sequence"][:7] == ["model.render", "functions.sample_border_points", "model.vector_field_network", "functions.get_center_indices_and_gt",
                                   "functions.sample_center_points", "model.vector_field_network", "loss"]


def test_deferred_scalars_carry_the_reference_loops_running_sums():
    """vf_nerf_amd/deferred.py: ``loss.item()`` and ``losses_dict[key]`` as numbers that stay on the device while they are only ADDED (the
    reference trainer's per-step accumulation, train/vector_field_nerf_train.py:262-275) and are plain floats on any other use.  Here on
    CPU tensors (the mechanics are device-independent): the reference's own accumulation code, then its division at the end of the epoch."""
    import json
    import math

    from vf_nerf_amd import loss as vloss
    from vf_nerf_amd.deferred import DeferredScalar, DeviceScalars, as_loss

    g = torch.Generator().manual_seed(3)
    steps = [torch.rand(8, generator=g) for _ in range(5)]
    average_losses = None
    reads = []
    for vec in steps:
        holder = DeviceScalars(vec.clone())
        loss = as_loss(vec[6].clone().requires_grad_(True) * 1.0, holder, 6)
        losses_dict = vloss._LazyTerms(vloss._NAMES, None, holder=holder)
        assert list(losses_dict.keys()) == list(vloss._NAMES) and len(losses_dict) == 6
        loss.backward()                                   # a tensor like any other
        if average_losses is None:                        # (the reference's code, verbatim in structure)
            average_losses = losses_dict
            average_losses["loss"] = loss.item()
        else:
            average_losses["loss"] += loss.item()
            for key in losses_dict.keys():
                average_losses[key] += losses_dict[key]
        reads.append(holder)
    # nothing has been read back so far: every per-step vector still sits on its "device"
    assert all(h._host is None for h in reads[1:]) and isinstance(dict.__getitem__(average_losses, "loss"), DeferredScalar)
    for key in average_losses.keys():
        average_losses[key] /= len(steps)
    want = torch.stack(steps).double().mean(dim=0)
    assert all(type(v) is float for v in dict.values(average_losses))
    for j, name in enumerate(vloss._NAMES):
        assert abs(average_losses[name] - float(want[j])) < 1e-12
    assert abs(average_losses["loss"] - float(want[6])) < 1e-12
    # any other use is the float: arithmetic with numbers, comparison, formatting, math functions, json through items()
    h = DeviceScalars(torch.tensor([0.25, 2.0]))
    a, b = DeferredScalar(h, 0), DeferredScalar(h, 1)
    assert a * 4 == 1.0 and 1 - a == 0.75 and b / a == 8.0 and a < b and f"{b:.1f}" == "2.0" and math.isfinite(a) and float(a) == 0.25
    assert a + 1 == 1.25 and 1 + a == 1.25 and type(a + 1) is float and repr(b) == "2.0" and round(b) == 2 and a + b == 2.25
    import numbers
    assert isinstance(a, numbers.Real)
    keep = vloss.DEFERRED_SCALARS
    try:
        vloss.DEFERRED_SCALARS = False
        d = vloss._LazyTerms(vloss._NAMES, None, holder=DeviceScalars(steps[0]))
        assert type(d["rgb_loss"]) is float
    finally:
        vloss.DEFERRED_SCALARS = keep
    d = vloss._LazyTerms(vloss._NAMES, None, holder=DeviceScalars(steps[0]))
    assert isinstance(d["rgb_loss"], DeferredScalar) and type(d.get("rgb_loss")) is float and type(d["rgb_loss"]) is float   # get() read it back
    assert json.loads(json.dumps(d)) == {k: float(steps[0][j]) for j, k in enumerate(vloss._NAMES)}


def test_deferred_scalars_in_a_per_step_logger():
    """What a user's per-step logger does with ``loss.item()`` / ``losses_dict[key]``: a wandb-style payload
    (dict -> json), pickling (a multiprocessing queue, ``torch.save`` of a metrics dict), ``copy.deepcopy``, numpy arrays of a history,
    ``isinstance(x, float)`` after ``deferred.resolve``.  The value is always the float; the foreign type never leaves the process."""
    import copy
    import json
    import pickle

    import numpy as np

    from vf_nerf_amd import deferred, dropin, loss as vloss
    from vf_nerf_amd.deferred import DeferredScalar, DeviceScalars

    vec = torch.tensor([0.5, 0.25, 0.125, 1.5, 0.0, 0.0, 2.375, 7.0])
    terms = lambda: vloss._LazyTerms(vloss._NAMES, None, holder=DeviceScalars(vec.clone()))
    d = terms()
    item = DeferredScalar(DeviceScalars(vec.clone()), 6)
    payload = {"train/loss": item, "train/terms": {k: d[k] for k in d.keys()}, "step": 3, "lr": 5e-4}
    assert isinstance(payload["train/terms"]["rgb_loss"], DeferredScalar)
    # json: with the encoder hook (either one), or resolved first for an encoder that takes no hook
    want = {"train/loss": 2.375, "train/terms": {k: float(vec[j]) for j, k in enumerate(vloss._NAMES)}, "step": 3, "lr": 5e-4}
    assert json.loads(json.dumps(payload, default=float)) == want
    assert json.loads(json.dumps(payload, default=deferred.json_default)) == want
    with pytest.raises(TypeError):
        json.dumps({"x": object()}, default=deferred.json_default)
    resolved = deferred.resolve(payload)
    assert json.loads(json.dumps(resolved)) == want and type(resolved["train/loss"]) is float and isinstance(resolved["train/loss"], float)
    assert all(type(v) is float for v in resolved["train/terms"].values()) and deferred.resolve([item, (item, 1)]) == [2.375, (2.375, 1)]
    # pickle / deepcopy: the float travels, the device vector does not
    back = pickle.loads(pickle.dumps({"loss": item, "history": [item, item]}))
    assert back == {"loss": 2.375, "history": [2.375, 2.375]} and type(back["loss"]) is float
    assert type(copy.deepcopy(item)) is float and type(copy.copy(item)) is float and copy.deepcopy({"a": [item]}) == {"a": [2.375]}
    import io
    buf = io.BytesIO()
    torch.save({"loss": item}, buf)
    buf.seek(0)
    assert torch.load(buf, weights_only=False) == {"loss": 2.375}
    # numpy: a history of per-step losses is a float64 array, not an object array
    hist = [DeferredScalar(DeviceScalars(vec.clone() + k), 6) for k in range(4)]
    arr = np.asarray(hist)
    assert arr.dtype == np.float64 and arr.shape == (4,) and np.allclose(arr, [2.375, 3.375, 4.375, 5.375])
    assert abs(float(np.mean(hist)) - 3.875) < 1e-12 and np.asarray(item).shape == () and np.float32(item) == np.float32(2.375)
    assert np.array(hist, dtype=np.float32).dtype == np.float32 and torch.tensor(hist, dtype=torch.float32).tolist() == [2.375, 3.375, 4.375, 5.375]
    # the switch for a logger that cannot be touched: plain floats at once (a synchronisation per step), through the drop-in's install()
    keep = vloss.DEFERRED_SCALARS
    try:
        dropin.install(deferred_scalars=False)
        assert vloss.DEFERRED_SCALARS is False and type(terms()["rgb_loss"]) is float
        dropin.install(deferred_scalars=True)
        assert isinstance(terms()["rgb_loss"], DeferredScalar)
    finally:
        vloss.DEFERRED_SCALARS = keep


def test_install_can_leave_torchs_clip_grad_norm_alone():
    """``dropin.install(patch_clip=False)``: ``torch.nn.utils.clip_grad_norm_`` is PyTorch's own function again and the flat optimizer clips
    inside ``step()`` when a step session parked its gradient (optim.CLIP_INSIDE_STEP); ``install()`` puts the wrapper back."""
    from vf_nerf_amd import dropin, optim
    wrapped = torch.nn.utils.clip_grad_norm_
    assert wrapped.__module__ == "vf_nerf_amd.dropin" and optim.CLIP_INSIDE_STEP is False
    try:
        dropin.install(patch_clip=False)
        assert torch.nn.utils.clip_grad_norm_ is wrapped.__wrapped__ and optim.CLIP_INSIDE_STEP is True
        assert torch.nn.utils.clip_grad_norm_.__module__.startswith("torch.")
        # PyTorch's own function on parameters without gradients (what a parked step looks like to it): nothing to scale
        p = torch.nn.Parameter(torch.ones(3))
        assert float(torch.nn.utils.clip_grad_norm_([p, p], 0.5)) == 0.0
    finally:
        dropin.install()
    assert torch.nn.utils.clip_grad_norm_.__module__ == "vf_nerf_amd.dropin" and optim.CLIP_INSIDE_STEP is False


def test_separable_lattice_is_recognised_row_by_row():
    """grid.lattice_axes / lattice_rows_match (the host side of get_set_predictions' lattice path): the grid evaluation/methods.py:190-208
    builds — here with its own operations, a translation and a centroid — is recognised from its axis tables and verified bit for bit;
    one perturbed coordinate, a NaN, another row order or another shape is refused; a rank verifies exactly the rows it evaluates."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from vf_nerf_amd import grid
    res = 24
    s = bench.reference_lattice(res, scale=1.3, translation=(0.1, -0.2, 0.05), centroid=(0.01, 0.3, 0.55))
    lat = grid.lattice_axes(s)
    assert lat is not None and lat[0] == res and all(a.shape == (res,) for a in lat[1:])
    # the tables ARE the grid: cell (i, j, k) at row (i res + j) res + k
    i, j, k = 5, 17, 3
    assert torch.equal(s[(i * res + j) * res + k], torch.stack([lat[1][i], lat[2][j], lat[3][k]]))
    all_rows = grid._rank_runs(res ** 3, 1000, 0, 1)
    assert grid.lattice_rows_match(s, res, lat[1:], all_rows) and grid.lattice_rows_match(s, res, lat[1:], all_rows, workers=3)
    covered = sorted(r for w in range(4) for r in grid._rank_runs(res ** 3, 1000, w, 4))
    assert covered[0][0] == 0 and covered[-1][1] == res ** 3 and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    bad = s.clone()
    bad[res ** 3 // 3, 1] += 1e-6                                   # one coordinate of one row, off by an ulp or so
    assert grid.lattice_axes(bad) is not None and not grid.lattice_rows_match(bad, res, lat[1:], all_rows)
    owner = (res ** 3 // 3) // 1000 % 4                             # only the rank that evaluates that row needs to refuse it
    for w in range(4):
        assert grid.lattice_rows_match(bad, res, lat[1:], grid._rank_runs(res ** 3, 1000, w, 4)) == (w != owner)
    nan = s.clone()
    nan[77, 2] = float("nan")
    assert not grid.lattice_rows_match(nan, res, lat[1:], all_rows)
    assert grid.lattice_axes(s[:-1]) is None and grid.lattice_axes(s.double()) is None and grid.lattice_axes(torch.rand(res ** 3, 3)) is None
    assert grid.lattice_axes(s.reshape(res, res, res, 3).permute(2, 1, 0, 3).reshape(-1, 3).contiguous()) is None      # k-major order
    assert grid.lattice_axes(torch.cat([s, torch.ones(res ** 3, 1)], 1)) is None
