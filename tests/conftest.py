import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _reproducible_library_convolutions():
    """The hand-written kernels sum nothing with atomics: a training step is bit-reproducible run to run (loss and every
    gradient, in every mode; scripts/lab/determinism.py).  Under split products (`math="f32_split"`, the headline) no convolution
    runs on a library at all.  In the exact-fp32 and bf16 modes the stem's 3x3 / 7x7 / strided convolutions are still MIOpen's, whose
    default solver choice includes split-K kernels with atomic accumulation (igemm_*_gkgs), forward included; on the random recipe
    weights that last-bit scatter is amplified to ~0.5 % of the bf16 loss, enough to make loose comparisons flip.  The GPU tests
    therefore ask MIOpen for its deterministic solvers (a no-op for f32_split); the bench keeps the default (fastest) ones."""
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    yield
    torch.backends.cudnn.deterministic = old


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


_SD_CACHE = {}


def recipe_sd(kind="hybrid", seed=0):
    """Reference-layout state dict filled by tests/golden/recipe.py (cached; treat as read-only)."""
    from recipe import recipe_state_dict
    key = (kind, seed)
    if key not in _SD_CACHE:
        fn = {"tiny": "state_dict_layout_tiny.json", "distil": "state_dict_layout_distil.json"}.get(kind, "state_dict_layout.json")
        with open(os.path.join(GOLDEN, fn)) as f:
            layout = json.load(f)
        if kind == "coco":                                  # train_acr_coco.py:91: ACR(num_classes=80) -- only the head differs
            layout["cls_head.weight"], layout["cls_head.bias"] = [80, 768], [80]
        _SD_CACHE[key] = recipe_state_dict(layout, seed)
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def hybrid_sd():
    return recipe_sd("hybrid")


@pytest.fixture(scope="session")
def tiny_sd():
    return recipe_sd("tiny")


def seed_planes(cams, t, shape):
    """(1 + n, h, w) planes evaluation.py:27-33 takes the argmax over, restricted to the classes present."""
    return np.stack([np.full(shape, t, np.float32)] + [cams[c] for c in sorted(cams)])


SEED_TIE_MARGIN = 4e-4      # a CONSTANT: twice the 2e-4 the fp32 CAMs are held to on [0, 1] at real geometry


def assert_seeds_exact_or_tie(got_seed, ref_seed, ref_cams, t, cam_err, what=""):
    """north_star: argmax seeds bit-exact.  A pixel may differ only where the reference's own decision is an fp tie: its
    top-1 / top-2 margin there is at most SEED_TIE_MARGIN -- a fixed number, not one scaled by this run's own error (the
    measured max |CAM - reference CAM| ``cam_err`` is printed next to it and must itself stay below half the margin for
    the argument "both planes moved by <= cam_err" to hold).  Anything else fails; ties are printed."""
    if np.array_equal(got_seed, ref_seed):
        return 0
    diff = got_seed != ref_seed
    planes = np.sort(seed_planes(ref_cams, t, ref_seed.shape), axis=0)
    margin = (planes[-1] - planes[-2])[diff]
    print("seed pixels differing %s t=%.1f: %d of %d, reference margins max %.3e (measured CAM error %.3e)"
          % (what, t, int(diff.sum()), diff.size, float(margin.max()), cam_err))
    assert float(margin.max()) <= SEED_TIE_MARGIN, (what, t, float(margin.max()), cam_err)
    assert 2.0 * cam_err <= SEED_TIE_MARGIN or float(margin.max()) <= 2.0 * cam_err + 1e-7, (what, t, cam_err)
    return int(diff.sum())


def has_gpu():
    return torch.cuda.is_available()
