"""Why do two-product colours drift with training?  Per checkpoint of a GPU training run: the measured two- vs three-product
colour difference (bench.two_product_check), a float64 simulation of the weights' f16 rounding alone on uniform points AND on the
ray samples of the check, and the magnitude statistics of the colour branch (folded weights, BatchNorm gains, activations).

    python tools/debug_gap.py [rays] [steps] [every]"""
import sys

import torch

sys.path.insert(0, '.')
import bench  # noqa: E402
from oracle import vfnerf_oracle as O  # noqa: E402
from vf_nerf_amd import supervision, trainer  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
every = int(sys.argv[3]) if len(sys.argv) > 3 else 500
dev = torch.device("cuda:0")
s_c = n_f = 64
centroid = (0.0, 0.0, 0.55)


def fold(sd, i):
    if f"layers.{i}.0.weight" in sd:
        W, b = sd[f"layers.{i}.0.weight"].double(), sd[f"layers.{i}.0.bias"].double()
        g, be, mu, var = (sd[f"layers.{i}.1.{k}"].double() for k in ("weight", "bias", "running_mean", "running_var"))
        s_ = g / torch.sqrt(var + 1e-5)
        return W * s_[:, None], (b - mu) * s_ + be
    return sd[f"layers.{i}.weight"].double(), sd[f"layers.{i}.bias"].double()


def sim(model, pts, dirs, which=("feat", 0, 1, 2, 3, 4)):
    """max |colour(exact weights) - colour(f16-rounded colour-branch weights)| in float64, and per-layer activation magnitudes."""
    vsd = {k: v.detach().cpu() for k, v in model.vector_field_network.state_dict().items()}
    rsd = {k: v.detach().cpu() for k, v in model.rendering_network.state_dict().items()}
    pts, dirs = pts.cpu(), dirs.cpu()
    r16 = lambda w: w.float().half().double()      # noqa: E731
    pe = O.positional_encoding(pts, 6).double()
    x = pe
    for i in range(8):
        W, b = fold(vsd, i)
        if i == 4:
            x = torch.cat([x, pe], 1)
            W = W / (2 ** 0.5)
        x = torch.relu(x @ W.T + b)
    W8, b8 = fold(vsd, 8)
    nrm = torch.tanh(x @ W8[:3].T + b8[:3])
    out, mags = [], {}
    for rounded in (False, True):
        Wf = r16(W8[3:]) if (rounded and "feat" in which) else W8[3:]
        pre = x @ Wf.T + b8[3:]
        feats = torch.tanh(pre)
        if not rounded:
            mags["trunk out max"] = float(x.abs().max()); mags["feat pre max"] = float(pre.abs().max()); mags["W8 feat max"] = float(W8[3:].abs().max())
        h = torch.cat([pts.double(), O.positional_encoding(dirs, 4).double(), nrm, feats], 1)
        for i in range(5):
            W, b = fold(rsd, i)
            if not rounded:
                mags[f"rn{i} |W| max"] = float(W.abs().max())
            if rounded and i in which:
                W = torch.cat([W[:, :33], r16(W[:, 33:])], 1) if i == 0 else r16(W)
            h = h @ W.T + b
            if not rounded:
                mags[f"rn{i} pre max"] = float(h.abs().max())
            h = torch.relu(h) if i < 4 else torch.sigmoid(h)
        out.append(h)
    return float((out[0] - out[1]).abs().max()), mags


teacher, _, _, _ = bench.build_scene(dev, 16, s_c, n_f, seed=0, weight_seed=1)
pool = trainer.TeacherTargets(teacher, views=8, width=64, height=64, focal=60.0, seed=5)
model, _, _, _ = bench.build_scene(dev, 16, s_c, n_f, seed=0, weight_seed=0)
model.rng_seed, model._rng_offset = 11, 0
supervision.manual_seed(3)
step = trainer.TrainStep(model, centroid, border_radius=0.15, far=1.0)
g = torch.Generator().manual_seed(977)
upts = (torch.rand(4096, 3, generator=g) - 0.5) + torch.tensor([0.0, 0.0, 0.55])
udirs = torch.nn.functional.normalize(torch.randn(4096, 3, generator=g), dim=-1)
for t in range(steps + 1):
    if t % every == 0:
        model.eval()
        pose, uv, K, _, _ = pool.batch(777_000, 1024)
        rec = bench.two_product_check(model, uv, pose, K)
        with torch.no_grad():
            o = model.render(pose[:64], uv[:64], K[:64], epoch=0)
        rp, rd = o.points_coarse.reshape(-1, 3), o.ray_dirs
        s_u, mags = sim(model, upts, udirs)
        s_r, mags_r = sim(model, rp, rd)
        parts = {str(w): sim(model, rp, rd, which=(w,))[0] for w in ("feat", 0, 1, 2, 3, 4)}
        bn = {f"rn{i} gamma max": float(model.rendering_network.layers[i][1].weight.abs().max()) for i in range(4)}
        print(f"step {t}: measured {rec['max_abs_colour_difference']:.2e} | sim uniform pts {s_u:.2e} | sim ray samples {s_r:.2e} | by part "
              + " ".join(f"{k}:{v:.1e}" for k, v in parts.items()), flush=True)
        print("    ", {k: round(v, 2) for k, v in mags_r.items()}, {k: round(v, 2) for k, v in bn.items()}, flush=True)
    if t < steps:
        b = pool.batch(t, n_rays)
        step(b[0], b[1], b[2], b[3], b[4], epoch=0)
