"""The C-ABI library loads on a CPU-only box and exports every symbol include/*.h declares
(no compute calls: those need a GPU and live in the -m gpu tests)."""
import ctypes as C
import os
import re

import pytest

from prosper_amd import capi, structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for header in ("prosper_pt.h", "prosper_host.h"):
        text = open(os.path.join(ROOT, "include", "prosper_pt", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(prosper_(?:pt|host)_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    syms = declared_symbols()
    assert len(syms) >= 30
    for name in syms:
        assert hasattr(lib, name), name
    assert lib.prosper_pt_abi_version() == 3


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.ProsperPtError) as e:
        capi.Context(device=0)
    assert e.value.code == -2  # PROSPER_PT_ERR_NO_DEVICE: no CPU fallback exists


def test_bad_arguments_are_rejected_before_touching_the_gpu():
    lib = capi.lib()
    assert lib.prosper_pt_create(None, None) == -1
    bad = S.DeviceDesc(4, 0, 0, 0)  # wrong struct_size
    h = C.c_void_p()
    assert lib.prosper_pt_create(C.byref(bad), C.byref(h)) == -1
    assert b"descriptor" in lib.prosper_pt_last_error()
    assert lib.prosper_pt_render(None, None, None, 1, 1, None, 0, None) == -1


def test_product_does_not_reference_the_oracle():
    """The product package must never import, link or call anything under oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "prosper_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("the oracle", "").replace("CPU oracle", ""), os.path.join(dirpath, f)
