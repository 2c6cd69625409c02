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


def force_variant(fwd, bwd):
    """Force an MSDA kernel variant for the following calls.  Variants exist only in the experiments build of the library
    (SEMIDETR_EXPERIMENTS=1 -> libsemidetr_hip_exp.so): on the product library every non-default request is skipped --
    tests/test_gpu_experiments.py re-runs those tests in a child process on the experiments build."""
    import semi_detr_amd
    if (fwd, bwd) != (0, 0) and not semi_detr_amd._lib.EXPERIMENTS:
        pytest.skip("kernel variants need the experiments build (SEMIDETR_EXPERIMENTS=1)")
    semi_detr_amd._lib.set_variant(fwd, bwd)


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


FULL_LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]


def full_shape_inputs(seed=3):
    """Exactly oracle/gen_golden.py:full_inputs -- the BASELINE.json micro-benchmark shape (N=2, Lq=300, L=4, M=8,
    P=4, D=32, S=22223) regenerated from the seed (torch's CPU generator is deterministic); tests/golden/msda_full.npz
    holds the reference's outputs for these inputs and a checksum of the inputs themselves."""
    import torch
    g = torch.Generator().manual_seed(seed)
    N, M, D, Lq, L, P = 2, 8, 32, 300, 4, 4
    S = sum(h * w for h, w in FULL_LEVELS)
    value = torch.rand(N, S, M, D, generator=g) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.rand(N, Lq, M * D, generator=g)
    return value, loc, attn, gout


def check_full_shape(out, gv, gl, ga, loc, golden):
    """out / gv / gl / ga (numpy) of the full micro-benchmark shape against the reference's CPU path."""
    shapes = np.asarray(FULL_LEVELS, np.int64)
    np.testing.assert_allclose(out, golden["out"], rtol=0, atol=2e-7)          # north-star bar: 1e-4
    np.testing.assert_allclose(ga, golden["gattn"], rtol=0, atol=2e-7)
    ok = ~(kink_mask(loc, shapes) | skipped_sample_mask(loc, shapes))
    scale = np.abs(golden["gloc"]).max()
    np.testing.assert_allclose(gl[ok], golden["gloc"][ok], rtol=0, atol=2e-5 * scale)
    gv = gv.reshape(2, -1, 256)
    np.testing.assert_allclose(gv[:, ::61], golden["gvalue_rows"], rtol=0, atol=2e-6)
    starts = np.concatenate([[0], np.cumsum([h * w for h, w in FULL_LEVELS])])
    sums = np.asarray([[gv[n, starts[l]:starts[l + 1]].astype(np.float64).sum() for l in range(4)] for n in range(2)])
    np.testing.assert_allclose(sums, golden["gvalue_level_sums"], rtol=2e-6)


def drive_mean_teacher_sequence(z, name, device, update_fn=None):
    """Replays one `seq.<name>` hook sequence of tests/golden/ema.npz -- produced by the REFERENCE's own MeanTeacher driven
    through before_run / before_train_iter / after_train_iter (oracle/gen_golden.py:gen_ema) -- on semi_detr_amd.MeanTeacher and
    yields (iteration, logged momentum or nan, hook.momentum, [teacher tensors]) after every iteration.  `update_fn` replaces
    the device kernel for the CPU test of the host logic (schedule, interval, decay, unwrapping, parameter pairing)."""
    import types
    import torch
    from semi_detr_amd import MeanTeacher
    cfgv = z[f"seq.{name}.cfg"]
    cfg = dict(momentum=float(cfgv[0]), interval=int(cfgv[1]), warm_up=int(cfgv[2]))
    if len(cfgv) > 4:
        cfg.update(decay_factor=float(cfgv[3]), decay_intervals=[int(v) for v in cfgv[4:]])
    n_par = len([k for k in z.files if k.startswith(f"seq.{name}.teacher0.")])

    def net(which):
        m = torch.nn.Module()
        for i in range(n_par):
            m.register_parameter(f"p{i}", torch.nn.Parameter(torch.from_numpy(z[f"seq.{name}.{which}0.{i}"].copy()).to(device)))
        m.p0.requires_grad_(False)
        m.register_buffer("buf", torch.from_numpy(z[f"seq.{name}.buf0"].copy()).to(device))
        return m
    model = torch.nn.Module()
    model.teacher, model.student = net("teacher"), net("student")
    hook = MeanTeacher(**cfg)
    if update_fn is not None:
        hook.momentum_update = lambda mdl, mom: update_fn(mdl, mom)
    wrapped = model
    if name == "wrapped":                      # the reference unwraps `runner.model.module` (mean_teacher.py:27-28)
        wrapped = torch.nn.DataParallel(model) if device != "cpu" else _Wrap(model)
    runner = types.SimpleNamespace(model=wrapped, iter=0, log_buffer=types.SimpleNamespace(output={}))
    hook.before_run(runner)
    for it in range(int(z["seq.iters"])):
        runner.iter = it
        runner.log_buffer.output.pop("ema_momentum", None)
        hook.before_train_iter(runner)
        logged = runner.log_buffer.output.get("ema_momentum", float("nan"))
        with torch.no_grad():
            for i in range(n_par):
                getattr(model.student, f"p{i}").copy_(torch.from_numpy(z[f"seq.{name}.student{it + 1}.{i}"]).to(device))
        hook.after_train_iter(runner)
        yield it, logged, hook.momentum, [getattr(model.teacher, f"p{i}").detach().cpu().numpy() for i in range(n_par)], model


class _Wrap(__import__("torch").nn.parallel.DistributedDataParallel.__mro__[1]):      # an nn.Module that looks like a wrapper
    def __init__(self, module):
        super().__init__()
        self.module = module
