"""Interleaved A/B of the launch plans of vfn_render_fwd in ONE process on ONE GPU: the batch in two halves on two streams inside the
call (the default), the five merged launches on one stream
(draws generated inside the ray / fine kernels, proposal argmax + fine sampler in one launch, the proposal results moved by the
composite launch) against the same pipeline through the stand-alone entry points (eight launches).  Same values either way.

    python tools/ab_render_plan.py > profiles/r02/ab_render_plan.txt"""
import sys, statistics, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
PLANS = {"2 ranges": (False, 2), "3 ranges": (False, 3), "4 ranges": (False, 4), "one stream": (False, 1), "eight launches, one stream": (True, 1)}
for rays, s_c, n_f in ((4096, 64, 64), (1024, 64, 64), (256, 64, 64), (4096, 100, 35), (1024, 100, 35)):
    model, uv, pose, K = bench.build_scene(dev, rays, s_c, n_f, 0)
    times = {k: [] for k in PLANS}
    with torch.no_grad():
        for k, (sep, streams) in PLANS.items():
            model.render_separate_launches, model.render_streams = sep, streams
            for _ in range(20): model.render(pose, uv, K, epoch=0)
        torch.cuda.synchronize()
        reps = max(20, 40960 // rays)
        for rnd in range(12):
            for k, (sep, streams) in PLANS.items():
                model.render_separate_launches, model.render_streams = sep, streams
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps): model.render(pose, uv, K, epoch=0)
                torch.cuda.synchronize()
                times[k].append((time.perf_counter() - t0) / reps * 1e3)
    med = {k: statistics.median(t) for k, t in times.items()}
    print(f"{rays:5d} rays x ({s_c} + {n_f}) samples per call: " + "   ".join(f"{k}: {v:.4f} ms ({rays / v / 1e3:.3f} M rays/s)" for k, v in med.items()))
