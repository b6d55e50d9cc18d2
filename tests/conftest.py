"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))

    return load


LAST_LAYERS = ("sam_w1", "clipseg_w1", "head_w2", "conv1_w", "conv1_b", "conv0_w", "conv0_b")


@pytest.fixture(scope="session")
def grad_parity():
    """Gradient criterion of the composed-step parity tests: `check(got, ref)` with dicts name -> array.

    A field's first-layer ReLU mask flips between the reference's CPU fp32 evaluation and the HIP path for a (sample, hidden
    unit) pair whose pre-activation lies within the ~2e-6 rounding of positions / interpolation of zero.  One flipped pair moves
    one row of that layer's weight gradient, and the table entries that sample touches, by ~1/sqrt(samples) of their size:
    0.5-4 % of the tensor's largest entry in a few elements, next to ~1e-5 everywhere else (tools/debug_ministep.py traces
    each outlier to its one sample).  So every tensor is held to a relative L1 error of 5e-3 -- a wrong term, level or scale
    is O(1) there -- and the tensors behind their network's last ReLU to 2e-4 of their largest entry.  The oracle evaluated in
    fp32 and in fp64 differs from itself in exactly this pattern (tests/test_oracle_golden.py::test_gradient_conditioning:
    field_table 6e-3 / base_w0 5e-3 / sam_w0 8e-3 of the largest entry, 1e-3 relative L1).  The proposal network's gradient
    comes from the interlevel loss alone, which is ~5e-11 in these untrained configurations -- a difference of nearly equal
    histograms (its fp32 and fp64 evaluations are 1e-3 apart): held to 3e-2; the loss kernels' own tests (test_ops_gpu.py)
    check that backward on well-conditioned inputs."""
    import numpy as np

    def check(got, ref, l1_tol=5e-3, max_tol=2e-4):
        l1, mx = {}, {}
        for k, r in ref.items():
            r = np.asarray(r, dtype=np.float64)
            a = got[k]
            a = a.detach().cpu().double().numpy() if hasattr(a, "detach") else np.asarray(a, dtype=np.float64)
            err = np.abs(a.reshape(r.shape) - r)
            l1[k], mx[k] = float(err.sum() / np.abs(r).sum()), float(err.max() / np.abs(r).max())
        report = {k: f"{l1[k]:.1e}/{mx[k]:.1e}" for k in l1}
        assert max(v for k, v in l1.items() if not k.startswith("prop_")) <= l1_tol, report
        assert max([v for k, v in l1.items() if k.startswith("prop_")] or [0.0]) <= 3e-2, report
        strict = [k for k in LAST_LAYERS if k in mx]
        assert strict and max(mx[k] for k in strict) <= max_tol, report
        return report

    return check
