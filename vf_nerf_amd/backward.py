"""Autograd wrappers of the training path (a13 in SURVEY.md §8).  Filled in after the forward path."""
from __future__ import annotations


def _todo(*_a, **_k):
    raise NotImplementedError(
        "the HIP backward kernels (dW/dX of the fused MLPs, scan/window-cosine backward) are not built yet: "
        "call under torch.no_grad() for rendering / evaluation")


vf_forward_autograd = _todo
render_forward_autograd = _todo
fine_pass_autograd = _todo
