"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re
import pytest

import vicalib_amd.lib as lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def test_library_is_built_and_loads():
    assert os.path.exists(lib.LIB_PATH), "run python __graft_entry__.py (build()) first"
    lib.load()


def test_header_symbols_all_exported():
    hdr = open(os.path.join(ROOT, "include", "vicalib_amd.h")).read()
    declared = set(re.findall(r"\b(vc_[a-z0-9_]+)\s*\(", hdr)) - {"vc_allreduce_fn"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    L = lib.load()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/vicalib_amd.h but not exported"


def test_no_cpu_fallback_without_device():
    if _have_gpu():
        pytest.skip("GPU present")
    L = lib.load()
    h = C.c_void_p()
    assert L.vc_create(C.byref(h), 0) == -1      # VC_ERR_NO_DEVICE
    assert not h.value
    with pytest.raises(lib.VicalibError):
        lib.ViCalibrator(0)


def test_product_never_references_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "vicalib_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", ".sh")):
                txt = open(os.path.join(base, f)).read()
                assert "vco_" not in txt and "oracle_lib" not in txt and "libvco" not in txt, f
