#!/usr/bin/env python3
"""Is tools/reference_sequence.py the reference trainer's loop body CALL FOR CALL?  Checked mechanically, in the build container (needs
/root/reference, read-only): the reference's OWN ``VectorFieldNerfRunner.train_epoch`` (train/vector_field_nerf_train.py:161-292, imported
with the host-only modules it pulls in stubbed, as tests/golden/make_train_golden.py does) and ``reference_sequence.ReferenceLoop`` are
both run on the SAME recording stand-ins — a model whose ``render`` / ``vector_field_network`` / ``parameters`` / optimizer / scheduler
record their calls, a recording loss, recording ``functions.*`` helpers, a recording ``clip_grad_norm_`` — and the two call traces
(names, argument shapes and scalar arguments, in order) must be identical.  Prints one JSON line; run by
tests/test_host_logic.py::test_reference_sequence_is_the_reference_trainers_call_sequence in a subprocess (it changes the working
directory and stubs modules)."""
import json
import os
import sys
import types
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(REF):
    print(json.dumps({"skipped": "no /root/reference"}))
    sys.exit(0)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


for _name in ("cv2", "imageio", "open3d", "skimage", "skimage.metrics", "skimage.transform", "skimage.io", "lpips", "trimesh", "wandb", "GPUtil",
              "configargparse", "pyhocon", "torchvision", "torchvision.transforms"):
    if _name not in sys.modules:
        m = _StubModule(_name)
        m.__path__ = []
        sys.modules[_name] = m
sys.path.insert(0, REF)
os.chdir(REF)                                   # the trainer does sys.path.append('.')
_ds = types.ModuleType("datasets.normal_datasets")
_ds.__path__ = []
_ds.dataset_dict = {}
sys.modules["datasets.normal_datasets"] = _ds
import train.vector_field_nerf_train as ref_train  # noqa: E402  (the REFERENCE's module)

import reference_sequence  # noqa: E402  (installs the drop-in aliases; the restatement under test)

N, S = 40, 16
DEV = torch.device("cpu")


def sig(x):
    if isinstance(x, torch.Tensor):
        return ["tensor", list(x.shape)]
    if isinstance(x, (int, float, bool, str)) or x is None:
        return x if not isinstance(x, float) else round(x, 9)
    if isinstance(x, torch.device):
        return str(x)
    if isinstance(x, dict):
        return {k: sig(v) for k, v in sorted(x.items())}
    if isinstance(x, (list, tuple)):
        return [sig(v) for v in x]
    return type(x).__name__


def make_world(trace):
    def rec(name, *args):
        trace.append([name] + [sig(a) for a in args])

    w = torch.nn.Parameter(torch.ones(1, 259))
    scale = torch.nn.Parameter(torch.ones(()))

    def render(pose, pixels, intrinsics, epoch, white=False):
        rec("model.render", pose, pixels, intrinsics, epoch, white)
        n = pixels.shape[0]
        return SimpleNamespace(points_coarse=torch.linspace(0, 1, n * S * 3).reshape(n, S, 3), coarse_normals=torch.ones(n, S, 3) * scale,
                               coarse_rgb_values=torch.ones(n, 3) * scale, coarse_depth_map=torch.ones(n, 1) * scale, directional_derivtives=None,
                               fine_normals=None, fine_rgb_values=None, fine_depth_map=None)

    def vf(points):
        rec("model.vector_field_network", points)
        return points.sum(dim=1, keepdim=True) * w

    class Opt:
        param_groups = [{"lr": 5e-4}]

        def zero_grad(self):
            rec("optimizer.zero_grad")

        def step(self):
            rec("optimizer.step")

    class Sched:
        def step(self):
            rec("scheduler.step")

    def parameters():
        rec("model.parameters")
        return [w, scale, w]

    nerf_cfg = SimpleNamespace(cuda_config=SimpleNamespace(device=DEV), border_supervision=True, center_supervision=True,
                               ray_sampler_config=SimpleNamespace(fine_sampling=lambda: True), scheduler_config=SimpleNamespace(clip_norm=0.5))
    model = SimpleNamespace(render=render, vector_field_network=vf, optimizer=Opt(), scheduler=Sched(), parameters=parameters, config=nerf_cfg)

    def loss(pred, gt, epoch):
        rec("loss", pred, gt, epoch)
        total = pred["rgb"].mean() + pred["depth"].mean() + pred["normals"].mean() + pred["supervised_normals"].mean()
        return total, {"rgb_loss": 0.1, "depth_loss": 0.2}

    g = torch.Generator().manual_seed(0)

    def sample_border_points(r_min, r_max, num_samples, centroid, device="cpu"):
        rec("functions.sample_border_points", r_min, r_max, num_samples, centroid, device)
        return torch.rand(num_samples, 3, generator=g), torch.rand(num_samples, 3, generator=g)

    def sample_center_points(centroid, radius, num_samples, device="cpu"):
        rec("functions.sample_center_points", centroid, radius, num_samples, device)
        return torch.rand(num_samples, 3, generator=g), torch.rand(num_samples, 3, generator=g)

    def get_center_indices_and_gt(points, normals, centroid, radius):
        rec("functions.get_center_indices_and_gt", points, normals, centroid, radius)
        return normals.reshape(-1, 3)[:5], torch.zeros(5, 3)

    def get_border_indices_and_gt(*a):
        rec("functions.get_border_indices_and_gt", *a)
        raise AssertionError("the 'center' init branch is not the one the shipped scenes take")

    functions = SimpleNamespace(sample_border_points=sample_border_points, sample_center_points=sample_center_points,
                                get_center_indices_and_gt=get_center_indices_and_gt, get_border_indices_and_gt=get_border_indices_and_gt)

    def clip_grad_norm_(parameters, max_norm, *a, **k):
        rec("torch.nn.utils.clip_grad_norm_", len(list(parameters)), max_norm)
        return torch.tensor(1.0)

    class Dataset:
        white_bkgd = False
        gt_mesh_centroid = torch.tensor([0.0, 0.0, 0.55])

        def get_vf_init_method(self):
            return ("exterior", "")

        def get_bounds(self):
            return 0.0, 1.0

        def get_centroid(self, device):
            return self.gt_mesh_centroid.to(device)

    batch = {"uv": torch.rand(1, N, 2, generator=g), "intrinsics": torch.eye(4).repeat(1, N, 1, 1), "pose": torch.eye(4).repeat(1, N, 1, 1),
             "rgb": torch.rand(1, N, 3, generator=g), "depth": torch.rand(1, N, 1, generator=g)}
    return SimpleNamespace(model=model, loss=loss, functions=functions, clip=clip_grad_norm_, dataset=Dataset(), batch=batch, nerf_cfg=nerf_cfg)


def run_reference(steps=2):
    trace = []
    wd = make_world(trace)
    runner = SimpleNamespace(model=wd.model, loss=wd.loss, dataset=wd.dataset, dataloader=[wd.batch] * steps,
                             config=SimpleNamespace(vf_nerf_config=wd.nerf_cfg, offline=True,
                                                    dataset_config=SimpleNamespace(dataset_name="replica", border_radius=0.15)))
    keep_fn, keep_clip = ref_train.functions, torch.nn.utils.clip_grad_norm_
    ref_train.functions, torch.nn.utils.clip_grad_norm_ = wd.functions, wd.clip
    try:
        stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")          # train_epoch prints its averages
        ref_train.VectorFieldNerfRunner.train_epoch(runner, 7)
    finally:
        sys.stdout = stdout
        ref_train.functions, torch.nn.utils.clip_grad_norm_ = keep_fn, keep_clip
    return trace


def run_restatement(steps=2):
    trace = []
    wd = make_world(trace)
    keep_fn, keep_clip = reference_sequence.functions, torch.nn.utils.clip_grad_norm_
    reference_sequence.functions, torch.nn.utils.clip_grad_norm_ = wd.functions, wd.clip
    try:
        loop = reference_sequence.ReferenceLoop(wd.model, wd.loss, wd.dataset, 0.15)
        for _ in range(steps):
            loop(wd.batch, 7)
    finally:
        reference_sequence.functions, torch.nn.utils.clip_grad_norm_ = keep_fn, keep_clip
    return trace


a, b = run_reference(), run_restatement()
first = next((i for i, (x, y) in enumerate(zip(a, b)) if x != y), None)
print(json.dumps({"calls_reference": len(a), "calls_restatement": len(b), "identical": a == b,
                  "first_difference": None if a == b else {"index": first if first is not None else min(len(a), len(b)),
                                                           "reference": a[first] if first is not None and first < len(a) else None,
                                                           "restatement": b[first] if first is not None and first < len(b) else None},
                  "sequence": [c[0] for c in a[:len(a) // 2]]}))
