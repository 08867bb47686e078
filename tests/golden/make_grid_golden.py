#!/usr/bin/env python3
"""Golden vectors for the dense-grid stages by running the REFERENCE's own functions
(evaluation/utils/mc_utils.py, evaluation/utils/guassian_smoothing.py) on small random fields.  Build container only
(needs /root/reference, read-only); the fixture holds inputs and expected outputs, nothing of the reference's source.

    python tests/golden/make_grid_golden.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from evaluation.utils import mc_utils  # noqa: E402
from evaluation.utils.guassian_smoothing import smooth_vf  # noqa: E402


def field(n, seed):
    """A smooth-ish field with a converging sheet so that some cells carry a surface (divergence == 1)."""
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, n)
    p = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), dim=-1).reshape(-1, 3)
    target = torch.tensor([0.15, -0.1, 0.05])
    plane_n = F.normalize(torch.tensor([0.6, -0.3, 0.74]), dim=0)
    dist = ((p - target) * plane_n).sum(-1, keepdim=True)
    v = -torch.sign(dist) * plane_n * (0.3 + dist.abs()) + 0.15 * torch.randn(n ** 3, 3, generator=g)
    return v


def main():
    out = {}
    for n, seed in ((10, 0), (13, 1)):
        pred = field(n, seed)
        div = mc_utils.extract_divergence(pred, n)
        sm3 = smooth_vf(pred.reshape(n, n, n, 3), k=3, sigma=1)
        sm9 = smooth_vf(pred.reshape(n, n, n, 3), k=9, sigma=2)
        norms = torch.norm(pred.clone(), dim=1)
        vt = F.normalize(pred, dim=1).reshape(n, n, n, 3)
        choice = mc_utils.unify_direction(div, vt.permute(3, 0, 1, 2), N=n)
        comb, pair_norms = mc_utils.make_comb_format(choice, norms, n)
        tag = f"n{n}"
        out.update({f"{tag}.pred": pred, f"{tag}.div": div, f"{tag}.smooth3": sm3, f"{tag}.smooth9": sm9, f"{tag}.choice": choice,
                    f"{tag}.comb": comb, f"{tag}.pair_norms": pair_norms})
        print(f"n={n}: surface cells {int(div.sum())} / {n ** 3}; corners siding with the second vector: {float(choice.float().mean()):.3f}")
    path = os.path.join(HERE, "grid_stages.npz")
    np.savez_compressed(path, **{k: v.numpy() for k, v in out.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
