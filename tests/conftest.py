import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def _experiments_only():
    """Tests of the measured-slower variants: they exist only in libprosper_pt_experiments.so (make EXPERIMENTS=1; run the
    suite with PROSPER_PT_LIB=prosper_amd/libprosper_pt_experiments.so).  Decided when the test runs, not at import: the
    library must not be loaded before the test modules are collected (one of them imports torch, which brings its own
    copy of the HIP runtime; whichever loads first is the one the process uses)."""
    from prosper_amd import capi
    if not capi.has_experiments():
        pytest.skip("library built without -DPPT_EXPERIMENTS")


needs_experiments = pytest.mark.usefixtures("_experiments_only")


@pytest.fixture(autouse=True)
def _no_debug_options_leak():
    """Process-wide debug options (capi.debug) never outlive the test that set them."""
    yield
    from prosper_amd import capi
    capi.debug()


@pytest.fixture(scope="session", autouse=True)
def _extra_streams():
    """PROSPER_TEST_EXTRA_STREAMS=n: n more HIP streams alive (and used once) for the whole session.  Which streams share a
    hardware queue - and with it the order in which the frames' streams get to run - depends on the streams alive in the
    process: a dependency that is missing between two of the library's streams can hide behind one sharing and show behind
    another (it did: tests/test_adoption.py "fifth_stream").  Run the GPU suite with 1 and 2 now and then."""
    n = int(os.environ.get("PROSPER_TEST_EXTRA_STREAMS", "0"))
    streams = []
    if n:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for _ in range(n):
            st, ev = ctypes.c_void_p(), ctypes.c_void_p()
            assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0 and hip.hipEventCreate(ctypes.byref(ev)) == 0
            assert hip.hipEventRecord(ev, st) == 0 and hip.hipStreamSynchronize(st) == 0
            hip.hipEventDestroy(ev)
            streams.append(st)
    yield
    # (left alive: contexts of the session fixture are closed after this one)


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def gpu_ctx():
    """One prosper_pt context on cuda:0 for the whole session (loads libprosper_pt.so; no fallback)."""
    from prosper_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()  # the test harness may build the product; the product itself never falls back
    ctx = capi.Context(device=0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def cornell_world():
    from prosper_amd import scenes
    return scenes.cornell(with_skybox=True)


def default_pc(structs, focal_length, frame_index=1, max_bounces=4, draw_type=0, ibl=False, dof=False,
               skip_history=True, accumulate=True, clamp=True, roulette=3):
    flags = 0
    flags |= structs.PC_FLAG_SKIP_HISTORY if skip_history else 0
    flags |= structs.PC_FLAG_ACCUMULATE if accumulate else 0
    flags |= structs.PC_FLAG_IBL if ibl else 0
    flags |= structs.PC_FLAG_DEPTH_OF_FIELD if dof else 0
    flags |= structs.PC_FLAG_CLAMP_INDIRECT if clamp else 0
    return structs.ReferencePC(draw_type, flags, frame_index, 1e-5, 1.0, focal_length, roulette, max_bounces)


def same_bits(a, b):
    """Bitwise equality of float32 arrays, with NaN == NaN regardless of payload."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    eq = a.view(np.uint32) == b.view(np.uint32)
    return eq | (np.isnan(a) & np.isnan(b))
