import json
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


def load_golden(name):
    """Returns (meta dict, arrays dict) of tests/golden/<name>.npz (written by make_golden.py
    from the real reference).  Plain numpy arrays only — no pickles."""
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"golden fixture {name}.npz not generated")
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    return meta, {k: z[k] for k in z.files if k != "meta"}


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(params=["det", "atomic"])
def reduction_mode(request):
    """Tests of kernels that combine partial sums across workgroups request this fixture to run in BOTH modes: the deterministic
    forms (partials through a workspace, fixed-order second pass) and the fp32-atomic forms (the default outside the tests)."""
    from diverse_channel_vit_amd import hip
    old = hip.set_deterministic(request.param == "det")
    yield request.param
    hip.set_deterministic(old)


@pytest.fixture(autouse=True)
def _deterministic_by_default(request):
    """GPU tests run in deterministic mode (the reference's trainer sets cudnn.deterministic = True, utils.py:394-401) unless they ask
    for `reduction_mode` themselves: results are then reproducible run to run and bounds can sit close to the measured value."""
    if "gpu" not in request.keywords or "reduction_mode" in request.fixturenames:
        yield
        return
    from diverse_channel_vit_amd import hip
    old = hip.set_deterministic(True)
    yield
    hip.set_deterministic(old)
