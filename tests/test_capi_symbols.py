"""The C-ABI shared library loads and exports every symbol include/laplace_hip.h declares
(no compute calls: there is no GPU in the CPU test tier)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "laplace_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from laplace_amd._lib import LIB_PATH, SIGNATURES, load_library

    if not os.path.exists(LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    lib = load_library()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in laplace_hip.h but not exported"
        assert name in SIGNATURES, f"{name} has no ctypes signature in laplace_amd/_lib.py"
    for name in SIGNATURES:
        assert name in names, f"{name} bound in _lib.py but not declared in laplace_hip.h"
    assert lib.lk_version() >= 100
    assert lib.lk_gram_workspace_bytes(576, 131072) > 0
    assert lib.lk_syevj_workspace_bytes(4608) >= 4 * 4608 * 4608 * 4


def test_product_path_fails_loudly_without_gpu_tensors():
    """No CPU fallback: handing CPU tensors to the HIP kernels raises."""
    import torch

    from laplace_amd._lib import HipKernels, LaplaceHipError

    K = HipKernels()
    with pytest.raises(LaplaceHipError):
        K.gram_tn(torch.zeros(4, 4), 1.0, torch.zeros(4, 4))


def test_missing_library_raises(tmp_path):
    from laplace_amd._lib import LaplaceHipError, load_library

    with pytest.raises(LaplaceHipError):
        load_library(str(tmp_path / "nope.so"))
