import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Grouped view of one tests/golden/*.npz: g['case'] -> dict of arrays for keys 'case.<field>'."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        self.cases = {}
        for k in self.z.files:
            if "." in k:
                c, f = k.split(".", 1)
                self.cases.setdefault(c, {})[f] = self.z[k]

    def __getitem__(self, case):
        return self.cases[case]

    def names(self):
        return sorted(self.cases)


@pytest.fixture(scope="session")
def golden_msda():
    return Golden("msda.npz")


@pytest.fixture(scope="session")
def golden_cost():
    return Golden("cost.npz")


@pytest.fixture(scope="session")
def golden_lsap():
    return Golden("lsap.npz")


@pytest.fixture(scope="session")
def golden_ema():
    return Golden("ema.npz")


@pytest.fixture(scope="session")
def golden_pseudo():
    return Golden("pseudo.npz")


@pytest.fixture(scope="session")
def golden_module():
    return Golden("msda_module.npz")


def skipped_sample_mask(loc, shapes):
    """True where the sample sits exactly on h == -1 or w == -1 (or h == H / w == W): there the reference's
    CUDA kernel skips the sample (strict inequalities, ms_deform_im2col_cuda.cuh:288) while its
    grid_sample-based debug path keeps a one-sided derivative -> grad_sampling_loc is not comparable."""
    H = shapes[:, 0].reshape(1, 1, 1, -1, 1).astype(loc.dtype)
    W = shapes[:, 1].reshape(1, 1, 1, -1, 1).astype(loc.dtype)
    h = loc[..., 1] * H - 0.5
    w = loc[..., 0] * W - 0.5
    m = (h == -1) | (w == -1) | (h == H) | (w == W)
    return np.broadcast_to(m[..., None], loc.shape)


def kink_mask(loc, shapes):
    """True where a sample lies (numerically) on a pixel centre line, i.e. h or w is an integer up to
    rounding.  The bilinear interpolant has a kink there: d/dloc is one-sided, and which side a
    implementation takes depends on how it rounds the coordinate (the reference's grid_sample path maps
    loc -> 2*loc-1 -> pixel, the CUDA kernel maps loc*W-0.5 directly).  Values and the other gradients are
    continuous, so only grad_sampling_loc is excluded on this measure-zero set."""
    H = shapes[:, 0].reshape(1, 1, 1, -1, 1).astype(np.float64)
    W = shapes[:, 1].reshape(1, 1, 1, -1, 1).astype(np.float64)
    h = loc[..., 1].astype(np.float64) * H - 0.5
    w = loc[..., 0].astype(np.float64) * W - 0.5
    tol = 1e-9 if loc.dtype == np.float64 else 1e-4
    m = (np.abs(h - np.round(h)) < tol) | (np.abs(w - np.round(w)) < tol)
    return np.broadcast_to(m[..., None], loc.shape)
