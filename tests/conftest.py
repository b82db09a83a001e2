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


def synthetic_3d(n, seed):
    """X ~ U[0,10]^3, y = sum_k sin x_k + 0.1 cos(10 x_0)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0., 10., size=(n, 3))
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    return x, y


def synthetic_stations(n, seed):
    """stations lat ~ U[25, 50] deg, lon ~ U[-125, -65] deg, h ~ U[0, 3000] m -> ECEF (WGS-84,
    km); temperature = 60 - 0.0065 h 1.8 + a smooth field + N(0, 1.75).  Returns (ecef km, elevation m, temperature)."""
    rng = np.random.default_rng(seed)
    lat = np.deg2rad(rng.uniform(25., 50., n))
    lon = np.deg2rad(rng.uniform(-125., -65., n))
    h = rng.uniform(0., 3000., n)
    a, f = 6378137.0, 1. / 298.257223563
    e2 = f * (2. - f)
    nu = a / np.sqrt(1. - e2 * np.sin(lat) ** 2)
    ecef = np.stack([(nu + h) * np.cos(lat) * np.cos(lon), (nu + h) * np.cos(lat) * np.sin(lon),
                     (nu * (1. - e2) + h) * np.sin(lat)], axis=1) / 1000.
    smooth = 8. * np.sin(3. * lat) * np.cos(2. * lon) + 5. * np.cos(5. * lon)
    temp = 60. - 0.0065 * h * 1.8 + smooth + rng.normal(0., 1.75, n)
    return ecef, h, temp


def temperature_covariance(ab):
    """The tuned covariance of examples/temperature_example/temperature_example.cc:34-85; the elevation scaling
    1 + factor * max(0, center - h) (temperature_example_utils.h:78-84) is supplied as an explicit scale column."""
    class Elevation(ab.ScalingFunction):
        def _call_impl(self, c):
            raise AssertionError("scale columns are supplied explicitly")

    cov = ab.ScalingTerm(Elevation()) * ab.Constant(5.07288) + ab.IndependentNoise(1.75027) \
        + ab.Exponential(1.10298, 1.0, ab.AngularDistance()) * ab.SquaredExponential(5835.56, 13.913, ab.RadialDistance())
    scale = lambda h: 1. + 0.000153439 * np.maximum(0., 4446.5 - h)
    return cov, scale
