"""How representative is `cpu_baseline.kind = "port"`?  The reference is Python and cannot travel to the GPU box, so bench.py times
the oracle's restatement there.  HERE (the build container, which has /root/reference) the reference's OWN
VectorFieldNerf.render() on the CPU (its --gpu cpu path) and the oracle's render() are timed on the same weights, rays and thread
count: if the two agree, the port's rays/s on the GPU box's host cores is a fair stand-in for the reference's.

    python tools/cpu_reference_vs_port.py [rays] [threads] > profiles/r03/cpu_reference_vs_port.json      (build container only)"""
import importlib.util, json, os, sys, time
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(REPO, "tests", "golden", "make_golden.py"))
mg = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mg)                      # (imports /root/reference: fails loudly anywhere else)
from oracle import vfnerf_oracle as O            # noqa: E402
from vf_nerf_amd import synthetic                 # noqa: E402

rays = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
fx = dict(mg.FIXTURES["bench_sizes"], n_rays=rays)          # 64 + 64 samples, stratified, shipped 8 x 256 / 4 x 256 geometry
model = mg.build_reference_model(fx)
uv, pose, K = synthetic.pinhole_batch(rays, fx["width"], fx["height"], fx["focal"], fx["cam_seed"], pose=synthetic.orbit_pose(25.0, 10.0, 0.9))
vf_sd = {k: v.detach() for k, v in model.vector_field_network.state_dict().items()}
rn_sd = {k: v.detach() for k, v in model.rendering_network.state_dict().items()}
settings = O.RenderSettings(n_samples=fx["n_samples"], n_fine=fx["n_importance"], perturb=True, dir_to_normal_th=fx["th"], fine_range=fx["fine_range"],
                            density=O.DensityParams(scale_min=1.0))


g = torch.Generator().manual_seed(5)
uni = dict(u_coarse=torch.rand(rays, fx["n_samples"], generator=g), u_fine=torch.rand(rays, fx["n_importance"], generator=g),
           u_add=torch.rand(rays, fx["n_importance"], generator=g))      # (the oracle takes its draws as inputs; the reference draws its own)


def timed(fn, budget=20.0):
    with torch.no_grad():
        fn()
        reps, t0 = 0, time.perf_counter()
        while True:
            fn()
            reps += 1
            el = time.perf_counter() - t0
            if el >= budget or reps >= 40:
                return reps, el


out = {}
for tag, fn in (("reference render() (models/nerf/vector_field_nerf.py:216-338, --gpu cpu)", lambda: model.render(pose, uv, K, 0, False)),
                ("oracle render() (oracle/vfnerf_oracle.py)", lambda: O.render(uv, pose, K, vf_sd, rn_sd, settings, **uni))):
    reps, el = timed(fn)
    out[tag] = {"rays_per_s": round(rays * reps / el, 1), "repetitions": reps, "seconds": round(el, 2)}
vals = [v["rays_per_s"] for v in out.values()]
print(json.dumps({"workload": f"{rays} rays x {fx['n_samples']} + {fx['n_importance']} samples, shipped geometry, torch fp32 CPU, {threads} threads "
                              f"({os.cpu_count()} CPUs in this container)", "results": out, "oracle_over_reference": round(vals[1] / vals[0], 3)}, indent=1))
