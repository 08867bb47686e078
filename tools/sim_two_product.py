#!/usr/bin/env python3
"""CPU simulation of candidate 2-product variants of the f16x3 kernels against the 1e-4 contract.

f16x3 evaluates every fp32-equivalent product as a_hi b_hi + a_hi b_lo + a_lo b_hi (hi = f16(v), lo = f16(v - hi)).  Dropping
one product moves the matrix-pipe ceiling from 2500/3 to 2500/2 TFLOP/s, at the price of one operand carried with 11
significant bits:
    "act_hi"   weights hi + lo, activations hi only      (w_hi x_hi + w_lo x_hi)
    "w_hi"     weights hi only, activations hi + lo      (w_hi x_hi + w_hi x_lo)
Products of two f16 values are exact in fp32 and the accumulation is fp32; emulated here with float64 matmuls of the split
operands rounded to fp32 per layer (an optimistic model of the accumulation).  Both MLPs, all golden fixtures, errors of the
normals / colours / composited rgb against the reference's golden outputs (fp32 CPU) and against a float64 evaluation.

    python tools/sim_two_product.py > profiles/r02/two_product_simulation.txt
"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, REPO)
from helpers import FIXTURE_NAMES, build_model, load_fixture, oracle_settings  # noqa: E402
from oracle import vfnerf_oracle as O  # noqa: E402

torch.set_num_threads(8)


def split(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


def mm(x, w, mode):
    xh, xl = split(x)
    wh, wl = split(w)
    d = lambda t: t.double()
    acc = d(xh) @ d(wh).T
    if mode in ("f16x3", "act_hi"):
        acc = acc + d(xh) @ d(wl).T
    if mode in ("f16x3", "w_hi"):
        acc = acc + d(xl) @ d(wh).T
    return acc.float()


def fold(sd, i):
    if f"layers.{i}.0.weight" in sd:
        W, b = sd[f"layers.{i}.0.weight"], sd[f"layers.{i}.0.bias"]
        g, be, mu, var = (sd[f"layers.{i}.1.{k}"] for k in ("weight", "bias", "running_mean", "running_var"))
        s = g / torch.sqrt(var + 1e-5)
        return W * s[:, None], (b - mu) * s + be
    return sd[f"layers.{i}.weight"], sd[f"layers.{i}.bias"]


def vf(points, sd, mode):
    pe = O.positional_encoding(points, 6)
    x = pe
    for i in range(9):
        W, b = fold(sd, i)
        if i == 4:
            x = torch.cat([x, pe], 1)
            W = W / (2 ** 0.5)
        y = mm(x, W, mode) + b
        x = torch.relu(y) if i < 8 else torch.tanh(y)
    return x


def rn(points, normals, dirs, feats, sd, mode):
    x = torch.cat([points, O.positional_encoding(dirs, 4), normals, feats], 1)
    for i in range(5):
        W, b = fold(sd, i)
        y = mm(x, W, mode) + b
        x = torch.relu(y) if i < 4 else torch.sigmoid(y)
    return x


def main():
    print(__doc__.split("\n\n")[0])
    print(f"{'fixture':15s} {'mode':7s} | normals vs golden | colours vs golden | rgb vs golden | normals vs f64 | contract 1e-4")
    worst = {}
    for name in FIXTURE_NAMES:
        fx, d = load_fixture(name)
        m = build_model(fx, d)
        vsd = {k: v.detach() for k, v in m.vector_field_network.state_dict().items()}
        rsd = {k: v.detach() for k, v in m.rendering_network.state_dict().items()}
        pts = d["points"].reshape(-1, 3)
        n, s_t = d["z_vals"].shape
        dirs = d["ray_dirs"].unsqueeze(1).repeat(1, s_t, 1).reshape(-1, 3)
        ref64 = O.vf_mlp(pts.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in vsd.items()})[:, :3]
        for mode in ("f16x3", "act_hi", "w_hi"):
            out = vf(pts, vsd, mode)
            nrm, feats = out[:, :3], out[:, 3:]
            col = rn(pts, nrm, dirs, feats, rsd, mode)
            rgb = (d["weights"].unsqueeze(-1) * col.reshape(n, s_t, 3)).sum(1)
            e_n = float((nrm - d["normals"].reshape(-1, 3)).abs().max())
            e_c = float((col - d["colors"]).abs().max())
            e_rgb = float((rgb - d["rgb"]).abs().max())
            e_64 = float((nrm.double() - ref64).abs().max())
            ok = max(e_n, e_c, e_rgb) < 1e-4
            worst[mode] = max(worst.get(mode, 0.0), e_n, e_c, e_rgb)
            print(f"{name:15s} {mode:7s} | {e_n:17.2e} | {e_c:17.2e} | {e_rgb:13.2e} | {e_64:14.2e} | {'holds' if ok else 'BROKEN'}")
    print()
    for mode, w in worst.items():
        print(f"worst over fixtures, {mode:7s}: {w:.2e}  ->  {'within' if w < 1e-4 else 'outside'} the 1e-4 contract")


if __name__ == "__main__":
    main()
