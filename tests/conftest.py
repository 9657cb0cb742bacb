import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zedo-release_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Every GPU test runs in BOTH arithmetic modes of the hidden layers with UNCHANGED tolerances: "f32" (exact fp32 MFMA, the
# default and the headline) and "f16x3" (split-fp16 operands on the fp16 matrix pipe, include/zedo_hip.h ZEDO_MATH_F16X3).
# The mode travels as the ZEDO_MATH environment variable, which zedo_hip.Weights reads when a handle is created and child
# processes inherit.  ZEDO_TEST_MATH=f32 (or f16x3) restricts a run to one mode.
MATH_MODES = [m for m in ("f32", "f16x3") if os.environ.get("ZEDO_TEST_MATH", m) == m]


def pytest_generate_tests(metafunc):
    if "math_mode" in metafunc.fixturenames and metafunc.definition.get_closest_marker("gpu"):
        metafunc.parametrize("math_mode", MATH_MODES, indirect=True, scope="session")


@pytest.fixture(scope="session", autouse=True)
def math_mode(request):
    mode = getattr(request, "param", None)
    if mode is None:                         # CPU tests: nothing to switch
        yield "f32"
        return
    old = os.environ.get("ZEDO_MATH")
    os.environ["ZEDO_MATH"] = mode
    yield mode
    if old is None:
        os.environ.pop("ZEDO_MATH", None)
    else:
        os.environ["ZEDO_MATH"] = old


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture(scope="session")
def weights0():
    from lib.dataset import synthetic as syn
    return syn.make_weights(seed=0)
