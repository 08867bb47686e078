"""Data-parallel training step on the HIP path with TWO ranks (BASELINE.json configs[3] in small).

RCCL needs one device per rank and the GPU box has one, so the two ranks share ``cuda:0`` and talk over gloo (which carries device
tensors through the host): everything else is the code a multi-GPU run executes — ``distributed.shard_rays``, the step session's render /
backward on each rank's shard, ONE all-reduce (mean) of the optimizer's flat gradient buffer where the reference's unchanged trainer
passes between ``backward()`` and ``optimizer.step()`` (inside the wrapped ``clip_grad_norm_``, or inside ``optimizer.step()`` with
``dropin.install(patch_clip=False)``), the clip over the duplicated parameter list, the flat Adam.  Against ONE process taking
the same step on the whole batch.  (The CPU twin of this test, tests/test_host_logic.py::test_two_rank_data_parallel_step_equals_the_
single_process_step, runs the oracle's autograd under the same collectives.)"""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out_dir, patched):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from helpers import build_model, load_fixture
    from vf_nerf_amd import distributed as vdist, dropin, stepengine
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dropin.install(patch_clip=patched)
        fx, d = load_fixture("bench_sizes")
        model = build_model(fx, d, device="cuda:0")
        if world > 1:
            vdist.broadcast_parameters(model, src=0)
        n = d["uv"].shape[0]
        lo, hi = vdist.shard_bounds(n, rank, world)
        g = {k: v.to(dev) for k, v in d.items() if isinstance(v, torch.Tensor)}
        pose, uv, K = vdist.shard_rays(g["pose"], g["uv"], g["intrinsics"], rank, world)
        uni = {k: g[k][lo:hi].contiguous() for k in ("u_coarse", "u_fine", "u_add")}
        gen = torch.Generator().manual_seed(4242)
        s_t = d["z_vals"].shape[1]
        a, b, c = torch.randn(n, 3, generator=gen), torch.randn(n, 1, generator=gen), 0.05 * torch.randn(n, s_t, 3, generator=gen)
        a, b, c = (t[lo:hi].to(dev) for t in (a, b, c))
        model.optimizer.zero_grad()
        out = model.render(pose, uv, K, epoch=0, uniforms=uni)
        eng = stepengine.StepEngine.of(model)
        assert eng.why_not is None and eng.session is not None, eng.why_not
        # means over rays (and samples): with equal shards the average of the ranks' gradients is the whole batch's gradient
        loss = (out.coarse_rgb_values * a).mean() + (out.coarse_depth_map * b).mean() + (out.coarse_normals * c).mean()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), model.config.scheduler_config.clip_norm)
        f = model.optimizer.flat()
        clipped = None
        if patched:                       # the wrapped clip has all-reduced and clipped the flat gradient: every rank holds the same buffer now
            clipped = f["grad"].detach().cpu().clone()
        else:
            assert float(norm) == 0.0 and "parked_max_norm" in f
        model.optimizer.step()
        model.scheduler.step()
        torch.cuda.synchronize()
        params = f["param"].detach().cpu().clone()
        torch.save({"loss": float(loss.detach()), "norm": float(norm), "clipped": clipped, "params": params, "rays": hi - lo,
                    "entries": [(p.numel(), m) for p, _, _, m in f["entries"]]}, os.path.join(out_dir, f"dp_{int(patched)}_w{world}_r{rank}.pt"))
    finally:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


@pytest.mark.parametrize("patched", [True, False], ids=["wrapped_clip", "clip_inside_optimizer_step"])
def test_two_rank_step_on_the_hip_path_equals_the_single_process_step(patched):
    port = 29800 + (os.getpid() % 1500) + (2000 if patched else 0)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(1, port, tmp, patched), nprocs=1, join=True)
        mp.spawn(_worker, args=(2, port, tmp, patched), nprocs=2, join=True)
        one = torch.load(os.path.join(tmp, f"dp_{int(patched)}_w1_r0.pt"))
        two = [torch.load(os.path.join(tmp, f"dp_{int(patched)}_w2_r{r}.pt")) for r in range(2)]
    assert one["rays"] == 96 and [t["rays"] for t in two] == [48, 48]
    assert abs(0.5 * (two[0]["loss"] + two[1]["loss"]) - one["loss"]) < 2e-6 * max(1.0, abs(one["loss"]))
    # replicas stay identical: same clipped gradient, same parameters after the step, bit for bit
    assert torch.equal(two[0]["params"], two[1]["params"])
    if patched:
        assert torch.equal(two[0]["clipped"], two[1]["clipped"]) and two[0]["norm"] == two[1]["norm"]
        # ... and equal to the single process's up to the order of the sums (per tensor, against its largest entry; the 16-bit storages'
        # own bound against exact gradients is 1e-3) — the norm, which counts the aliased parameters twice (Q4), to 1e-3
        assert abs(two[0]["norm"] - one["norm"]) < 1e-3 * one["norm"]
        off, worst = 0, 0.0
        for numel, _ in one["entries"]:
            ref, got = one["clipped"][off:off + numel], two[0]["clipped"][off:off + numel]
            worst = max(worst, float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30))
            off += numel
        print(f"two ranks vs one process: worst clipped-gradient difference {worst:.2e} of a tensor's largest entry; clip norm {two[0]['norm']:.6f} / {one['norm']:.6f}")
        assert worst < 2e-3
    # the parameters moved by an Adam step on both sides, and to the same place except where an update is a coin flip (|g| ~ eps):
    # at most 2 % of the entries further apart than a twentieth of one update (lr 5e-4; the aliased parameters take two)
    moved = (two[0]["params"] - one["params"]).abs()
    frac = float((moved > 0.05 * 5e-4).float().mean())
    print(f"parameters after the step: {frac:.4f} of the entries differ by more than a twentieth of an update")
    assert frac < 0.02


def _loop_worker(rank, world, port, out_dir):
    """Ten steps of the reference trainer's loop body (tools/reference_sequence.ReferenceLoop: render, both supervision batches with THIS
    rank's own random points, VFLoss with its data-dependent centre-ball rows, zero_grad, backward, clip_grad_norm_ — where the drop-in
    all-reduces —, optimizer.step, scheduler.step) on this rank's 64-ray shard of the recorded run's first batches."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    from types import SimpleNamespace
    import torch.distributed as dist
    import reference_sequence
    import replay_reference_run as rr
    from vf_nerf_amd import distributed as vdist, loss as vloss, stepengine, trainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        raw, recipe = rr.load_task("256")
        model = rr.build_student(raw, recipe, dev)
        vdist.broadcast_parameters(model, src=0)
        vdist.seed_rank_streams(model, rank, base_seed=3)                   # every rank its own jitter and supervision points
        batches = rr.batches_on(raw, dev)
        crit = vloss.VFLoss(SimpleNamespace(**trainer.SHIPPED_LOSS_CONFIG), SimpleNamespace(**trainer.SHIPPED_LOSS_WEIGHTS))
        loop = reference_sequence.ReferenceLoop(model, crit, reference_sequence.StandInDataset(recipe["centroid"], recipe["far"], recipe["near"]),
                                                recipe["border_radius"], clip_norm=recipe["clip_norm"], sync_each_step=True)
        losses = []
        for t in range(10):
            # a global batch of 128 rays = two recorded 64-ray batches; this rank takes one of them
            b = batches[2 * t + rank]
            loss, _ = loop(b, 0)
            eng = stepengine.StepEngine.of(model)
            assert eng.why_not is None and eng.session is not None, eng.why_not
            losses.append(float(loss))
        torch.cuda.synchronize()
        f = model.optimizer.flat()
        torch.save({"params": f["param"].detach().cpu().clone(), "exp_avg": f["exp_avg"].detach().cpu().clone(), "losses": losses,
                    "running_loss": float(loop.average_losses["loss"])}, os.path.join(out_dir, f"loop_r{rank}.pt"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_through_the_reference_loop_stay_identical_replicas():
    """The data-parallel invariant over several steps of the UNCHANGED loop body: ranks see different rays, draw different jitter and
    supervision points, select different numbers of centre-ball rows — and after every step hold the SAME parameters and Adam moments,
    because the one all-reduce (inside the wrapped clip_grad_norm_) hands every rank the same gradient before the clip and the update."""
    port = 31800 + (os.getpid() % 1500)
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_loop_worker, args=(2, port, tmp), nprocs=2, join=True)
        a, b = (torch.load(os.path.join(tmp, f"loop_r{r}.pt")) for r in range(2))
    assert torch.equal(a["params"], b["params"]) and torch.equal(a["exp_avg"], b["exp_avg"])
    assert a["losses"] != b["losses"] and all(l == l and l < 10.0 for l in a["losses"] + b["losses"])      # different shards, finite
    assert abs(a["running_loss"] - sum(a["losses"])) < 1e-4 * abs(sum(a["losses"]))                        # the loop's deferred running sum
    print(f"two ranks, ten steps of the reference loop: losses rank 0 {a['losses'][0]:.4f} -> {a['losses'][-1]:.4f}, rank 1 {b['losses'][0]:.4f} -> {b['losses'][-1]:.4f}; "
          "parameters and first moments bit-identical")
