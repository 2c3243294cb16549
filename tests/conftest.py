import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def _use_sim():
    from comat_amd import ops
    from sim_backend import SimKernels
    ops.set_kernel_backend(SimKernels())
    return torch.device("cpu")


def _use_hip():
    from comat_amd import _hip, ops
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test run without a GPU")
    ops.set_kernel_backend(_hip.HipKernels())  # raises if the .so is missing: no silent fallback
    return torch.device("cuda:0")


def _reset_stream_state():
    """a test owns its kernel backend instance (and with it every per-stream workspace): drop the package-level stream
    state that refers to the previous test's - the capture stream (a test that failed inside a capture leaves it in capture
    mode), queued weight gradients, side-stream bookkeeping"""
    from comat_amd import ops
    if torch.cuda.is_available():
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001 - a pending error of the test that just failed
            pass
        ops.reset_capture_stream(torch.device("cuda:0"))
    ops.drop_side_stream_state()


@pytest.fixture(params=["sim", pytest.param("hip", marks=pytest.mark.gpu)])
def dev(request):
    """Device + kernel backend: 'sim' = CPU simulator of the C ABI (host-logic check, runs anywhere),
    'hip' = the real libcomat_hip.so on cuda:0 (the parity tests proper)."""
    d = _use_sim() if request.param == "sim" else _use_hip()
    yield d
    from comat_amd import ops
    _reset_stream_state()
    ops.set_kernel_backend(None)


@pytest.fixture
def sim():
    d = _use_sim()
    yield d
    from comat_amd import ops
    ops.set_kernel_backend(None)


@pytest.fixture
def hip():
    d = _use_hip()
    yield d
    from comat_amd import ops
    _reset_stream_state()
    ops.set_kernel_backend(None)
