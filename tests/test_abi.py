"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol that
include/dfine_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "dfine_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dfine_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from custom_d_fine_amd.csrc import build
    lib_path = build.build(verbose=False)
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dfine_hip.h but not exported"


def test_binding_covers_header():
    from custom_d_fine_amd import hip
    assert sorted(hip.EXPORTED) == _declared()
    assert hip._lib.dfine_abi_version() == hip.ABI_VERSION


def test_library_targets_gfx950():
    from custom_d_fine_amd.csrc import build
    blob = open(build.build(verbose=False), "rb").read()
    assert b"gfx950" in blob


def test_product_ops_refuse_cpu_tensors():
    """No CPU fallback in the product: HIP-backed operators raise on CPU tensors."""
    import pytest
    import torch
    from custom_d_fine_amd import kernels
    assert kernels._TEST_BACKEND is None
    v = torch.zeros(1, 4, 1, 16)
    with pytest.raises(RuntimeError, match="HIP kernel"):
        kernels.msda(v, [(2, 2)], torch.zeros(1, 1, 1, 1, 2), torch.zeros(1, 1, 1, 1), [1])
    with pytest.raises(RuntimeError, match="HIP kernel"):
        kernels.hungarian_assign(torch.zeros(1, 1, 2, 3), torch.zeros(1, 1, 2, 4), torch.zeros(1, dtype=torch.long),
                                 torch.zeros(1, 4), [1], 2., 5., 2., .25, 2.)
