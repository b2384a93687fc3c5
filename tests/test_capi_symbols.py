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
    assert lib.prosper_pt_abi_version() == 4


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


def test_debug_options_struct_matches_the_binding():
    """prosper_pt_debug_options <-> structs.DebugOptions: same size, and the defaults the header documents."""
    o = S.DebugOptions()
    capi.lib().prosper_pt_debug_options_default(C.byref(o))
    assert o.struct_size == C.sizeof(S.DebugOptions)
    signed = ("batchedTextures", "widePacks", "alphaCellShift", "nodeOrder", "childOrder", "bandedBatches")
    assert all(getattr(o, n) == -1 for n in signed)
    others = [n for n, _ in S.DebugOptions._fields_ if n not in ("struct_size",) + signed]
    assert all(getattr(o, n) == 0 for n in others)
    # every field of the header's struct, in order
    text = open(os.path.join(ROOT, "include", "prosper_pt", "prosper_pt.h")).read()
    body = text[text.index("typedef struct prosper_pt_debug_options"):text.index("} prosper_pt_debug_options;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = re.findall(r"\b(?:uint32_t|int32_t|float)\s+([A-Za-z_]+);", body)
    assert names == [n for n, _ in S.DebugOptions._fields_]


def test_the_library_does_not_read_the_environment_outside_create():
    """A plugin in someone else's process: no ambient variable may change what it does.  The one opt-in gate
    (PROSPER_PT_DEBUG=1 -> PROSPER_PT_DEBUG_OPTIONS, read by prosper_pt_create) is all the binary knows of."""
    import subprocess
    names = subprocess.run(["strings", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout.splitlines()
    hits = [n for n in names if "PROSPER_PT_DEBUG" in n or "PROSPER_PT_REBUILD" in n]
    assert sorted(hits) == ["PROSPER_PT_DEBUG", "PROSPER_PT_DEBUG_OPTIONS"], hits
    calls = {}
    base = os.path.join(ROOT, "prosper_amd", "csrc")
    for dirpath, _, files in os.walk(base):
        for f in files:
            if f.endswith((".cpp", ".hpp", ".hip")):
                n = open(os.path.join(dirpath, f)).read().count("getenv(")
                if n:
                    calls[os.path.relpath(os.path.join(dirpath, f), base)] = n
    assert calls == {"prosper_pt.cpp": 2}, calls
    src = open(os.path.join(base, "prosper_pt.cpp")).read()
    gate = src[src.index("bool debug_options_from_environment"):]
    assert gate[:gate.index("\n}\n")].count("getenv(") == 2  # both inside the gate that only prosper_pt_create calls
    assert src.count("debug_options_from_environment(") == 2  # its definition and that one call


@pytest.mark.gpu
def test_the_environment_gate_is_read_once_at_create_and_only_when_asked():
    """PROSPER_PT_DEBUG=1 + PROSPER_PT_DEBUG_OPTIONS="name=value,...": parsed into a new context's options by prosper_pt_create;
    without the gate variable the options string is ignored; an unknown name or a bad value fails the create; changing the
    environment afterwards changes nothing (no other call looks at it)."""
    import subprocess
    import sys
    script = r"""
import os, sys
sys.path.insert(0, %r)
from prosper_amd import capi
try:
    ctx = capi.Context(device=0)
except capi.ProsperPtError as e:
    print("create failed:", e)
    sys.exit(0)
o = ctx.debug_options()
print("options", o.ldsStackEntries, "%%.1e" %% o.boxPad, o.widePacks, o.noLdsTables)
os.environ["PROSPER_PT_DEBUG_OPTIONS"] = "ldsStackEntries=32"
from prosper_amd import scenes
ctx.upload_scene(scenes.cornell())
print("after upload", ctx.debug_options().ldsStackEntries)
""" % ROOT

    def run(env_extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("PROSPER_PT_DEBUG")}
        env.update(env_extra)
        out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stderr[-1500:]
        return out.stdout
    text = run({"PROSPER_PT_DEBUG": "1", "PROSPER_PT_DEBUG_OPTIONS": "ldsStackEntries=16,boxPad=3e-5,widePacks=0,noLdsTables"})
    assert "options 16 3.0e-05 0 1" in text and "after upload 16" in text, text
    text = run({"PROSPER_PT_DEBUG_OPTIONS": "ldsStackEntries=16"})  # no gate: ignored
    assert "options 0 0.0e+00 -1 0" in text, text
    assert "unknown option 'stack'" in run({"PROSPER_PT_DEBUG": "1", "PROSPER_PT_DEBUG_OPTIONS": "stack=16"})
    assert "is not a value" in run({"PROSPER_PT_DEBUG": "1", "PROSPER_PT_DEBUG_OPTIONS": "ldsStackEntries=sixteen"})
    assert "ldsStackEntries is 0, 16, 24 or 32" in run({"PROSPER_PT_DEBUG": "1", "PROSPER_PT_DEBUG_OPTIONS": "ldsStackEntries=20"})
