import json
import os
import sys

import numpy as np
import pytest
import torch

if torch.cuda.is_available():
    # torch's bundled HIP runtime must be initialised before libalbatross_amd.so
    # (system ROCm) makes its first HIP call — see albatross_amd/distributed.py
    torch.cuda.init()

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def ctx():
    """One agp_context on cuda:0 for the whole GPU session.  Fails loudly when
    the HIP library is missing or no device is visible: there is no fallback."""
    import albatross_amd as ab
    c = ab.Context(0)
    yield c
    c.close()


@pytest.fixture
def make_ctx():
    """Contexts created INSIDE a test, after it has set the switches the library reads once per context
    (include/albatross_amd.h, "switches": AGP_STEP_BELOW, AGP_GRAM_SOP, AGP_SHARD_BLOCK, ...)."""
    import albatross_amd as ab
    made = []

    def make():
        c = ab.Context(0)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()


def synthetic_3d(n, seed):
    """SURVEY.md section 8d configs 2/3, exactly as pinned there and as bench.py times them: X ~ U[0,10]^3 from libstdc++'s
    mt19937(seed) + uniform_real_distribution (bench.mt19937_uniform reproduces it bit for bit, checked against the
    compiled generator's tests/golden/bench512.json), y = sum_k sin x_k + 0.1 cos(10 x_0)."""
    from bench import make_dataset
    return make_dataset(n, seed)


def synthetic_stations(n, seed):
    """SURVEY.md section 8d config 4 (bench.synthetic_stations): (ecef km, elevation m, temperature)."""
    import bench
    return bench.synthetic_stations(n, seed)


def temperature_covariance(ab):
    """The tuned covariance of examples/temperature_example/temperature_example.cc:34-85 (bench.temperature_covariance)."""
    import bench
    return bench.temperature_covariance(ab)
