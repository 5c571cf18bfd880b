import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo():
    return REPO


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def weights():
    from phyloformer_amd.weights import load_weights
    cache = {}

    def get(name="pf"):
        if name not in cache:
            cache[name] = load_weights(os.path.join(REPO, "models", f"{name}.ckpt"))
        return cache[name]
    return get


@pytest.fixture(scope="session")
def engines(weights):
    """Engine factory for -m gpu tests; fails loudly if the native library or GPU is missing."""
    from phyloformer_amd.engine import Engine
    cache = {}

    def get(name="pf", precise=-1):
        """precise: -1 = the product's choice (float64 kernels for ill-conditioned shapes, csrc/pf_precise.hip.h);
        0 = the default split-fp16 kernels on every shape - what the tests of their tile / group / shard edge
        cases at small sizes want."""
        if (name, precise) not in cache:
            cache[name, precise] = Engine(weights(name), device=0)
            cache[name, precise].set_option("precise", precise)
        return cache[name, precise]
    yield get
    for e in cache.values():
        e.close()
