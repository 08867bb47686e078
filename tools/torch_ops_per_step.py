"""Which torch (ATen) operators a training step still runs beside the package's own kernels, with counts and device time:
    python tools/torch_ops_per_step.py [rays]"""
import sys, torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import trainer
from torch.profiler import profile, ProfilerActivity
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
model, uv, pose, K = bench.build_scene(dev, rays, 64, 64, 0)
teacher, _, _, _ = bench.build_scene(dev, 16, 64, 64, seed=0, perturb=False, weight_seed=1)
teacher.precision = "fp32"
with torch.no_grad():
    t = teacher.render(pose, uv, K, 0)
rgb_gt, depth_gt = t.coarse_rgb_values.clone(), t.coarse_depth_map.clone()
step = trainer.TrainStep(model, (0.0, 0.0, 0.6), border_radius=0.05, far=1.0)
for _ in range(10):
    step(pose, uv, K, rgb_gt, depth_gt)
torch.cuda.synchronize()
N = 20
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step(pose, uv, K, rgb_gt, depth_gt)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.count >= N]
rows.sort(key=lambda e: -(getattr(e, "device_time_total", 0) or 0))
for e in rows[:30]:
    print(f"{e.count / N:6.1f} per step  {e.key:28s} device {(getattr(e, 'device_time_total', 0) or 0) / N:7.1f} us per step")
