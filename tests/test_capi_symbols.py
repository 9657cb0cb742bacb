"""CPU: the C-ABI library loads and exports exactly the symbols include/zedo_hip.h declares
(no compute calls without a GPU), and the product path refuses to run without a GPU."""
import ctypes
import os
import re

import pytest
import torch  # noqa: F401  (must precede dlopen of libzedo_hip.so: torch bundles its own HIP runtime)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zedo-release_amd", "zedo_hip", "libzedo_hip.so")
HDR = os.path.join(ROOT, "include", "zedo_hip.h")


def declared_functions():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zedo_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    return ctypes.CDLL(LIB)


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for n in ("zedo_weights_create", "zedo_schedule_create", "zedo_reproj_prepare", "zedo_reproj_grad",
              "zedo_score_eps", "zedo_sde_step", "zedo_oil_run", "zedo_ipo_fit", "zedo_rotate_init",
              "zedo_min_mpjpe", "zedo_workspace_bytes"):
        assert n in names


def test_library_exports_every_declared_symbol(lib):
    for n in declared_functions():
        assert hasattr(lib, n), f"{n} declared in include/zedo_hip.h but not exported"
    assert lib.zedo_abi_version() == 5
    lib.zedo_error_string.restype = ctypes.c_char_p
    assert b"workspace" in lib.zedo_error_string(-3)


def test_binding_matches_header(lib):
    import zedo_hip
    assert sorted(zedo_hip.SIGNATURES) == declared_functions()
    lib.zedo_workspace_bytes.restype = ctypes.c_size_t
    assert zedo_hip.workspace_bytes(1000) == 1024 * (64 + 2048) * 4
    assert zedo_hip.param_names() == [n for n, _ in __import__("lib.dataset.synthetic", fromlist=["x"]).state_dict_layout()]


def test_no_cpu_fallback():
    import torch
    import zedo_hip
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zedo_hip.ZedoError):
        zedo_hip.reproj_prepare(torch.zeros(2, 17, 2), torch.zeros(2, 3, 3))
    with pytest.raises(zedo_hip.ZedoError):
        zedo_hip.Weights({})
