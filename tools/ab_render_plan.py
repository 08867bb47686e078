"""Interleaved A/B of the two launch plans of vfn_render_fwd in ONE process on ONE GPU, one stream: the five merged launches
(draws generated inside the ray / fine kernels, proposal argmax + fine sampler in one launch, the proposal results moved by the
composite launch) against the same pipeline through the stand-alone entry points (eight launches).  Same values either way.

    python tools/ab_render_plan.py > profiles/r02/ab_render_plan.txt"""
import sys, statistics, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for rays in (4096, 1024, 256):
    model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, 0)
    times = {False: [], True: []}
    with torch.no_grad():
        for sep in (False, True):
            model.render_separate_launches = sep
            for _ in range(20): model.render(pose, uv, K, epoch=0)
        torch.cuda.synchronize()
        reps = max(20, 40960 // rays)
        for rnd in range(12):
            for sep in (False, True):
                model.render_separate_launches = sep
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps): model.render(pose, uv, K, epoch=0)
                torch.cuda.synchronize()
                times[sep].append((time.perf_counter() - t0) / reps * 1e3)
    a, b = statistics.median(times[False]), statistics.median(times[True])
    print(f"{rays:5d} rays x 128 samples per call: five launches {a:.4f} ms ({rays / a / 1e3:.3f} M rays/s)   eight launches {b:.4f} ms "
          f"({rays / b / 1e3:.3f} M rays/s)   {b / a:.3f}x")
