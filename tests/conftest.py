import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-fusion_amd"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        path = os.path.join(ROOT, "tests", "golden", name + ".npz")
        return dict(np.load(path, allow_pickle=False))
    return load


@pytest.fixture(params=[1, 0], ids=["split-bf16", "fp32-mfma"])
def mlp_precision(request):
    """Runs a test under both arithmetic modes of the attentive pools' dense chains
    (mvf_set_mlp_precision: 1 = wide and three-layer chains on the bf16 matrix cores with
    three-way split float32 operands, the default; 0 = every chain on the float32 cores)."""
    from mvfusion import _lib
    lib = _lib.load()
    before = lib.mvf_get_mlp_precision()
    assert lib.mvf_set_mlp_precision(request.param) == 0
    yield request.param
    lib.mvf_set_mlp_precision(before)
