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


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


_SD_CACHE = {}


def recipe_sd(kind="hybrid", seed=0):
    """Reference-layout state dict filled by tests/golden/recipe.py (cached; treat as read-only)."""
    from recipe import recipe_state_dict
    key = (kind, seed)
    if key not in _SD_CACHE:
        fn = "state_dict_layout.json" if kind == "hybrid" else "state_dict_layout_tiny.json"
        with open(os.path.join(GOLDEN, fn)) as f:
            layout = json.load(f)
        _SD_CACHE[key] = recipe_state_dict(layout, seed)
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def hybrid_sd():
    return recipe_sd("hybrid")


@pytest.fixture(scope="session")
def tiny_sd():
    return recipe_sd("tiny")


def has_gpu():
    return torch.cuda.is_available()
