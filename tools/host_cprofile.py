"""cProfile of 100 training steps (trainer.TrainStep, 1 024 rays x 128): where the HOST time of a step goes, function by function
(tools/host_profile.py has the per-phase view).  `run_backward` carrying ~1 ms of its own means the host is waiting for the device
there, i.e. the step is device-bound:    python tools/host_cprofile.py"""
import sys, cProfile, pstats, torch
sys.path.insert(0, '.')
import bench
from vf_nerf_amd import trainer, supervision
dev = torch.device("cuda:0")
model, uv, pose, K = bench.build_scene(dev, 1024, 64, 64, 0)
teacher, _, _, _ = bench.build_scene(dev, 16, 64, 64, seed=0, perturb=False, weight_seed=1)
teacher.precision = "fp32"
with torch.no_grad():
    t = teacher.render(pose, uv, K, 0)
rgb_gt, depth_gt = t.coarse_rgb_values.clone(), t.coarse_depth_map.clone()
step = trainer.TrainStep(model, (0.0, 0.0, 0.6), border_radius=0.05, far=1.0)
for _ in range(10): step(pose, uv, K, rgb_gt, depth_gt)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step(pose, uv, K, rgb_gt, depth_gt)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
