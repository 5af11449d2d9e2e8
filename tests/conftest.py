import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The parity tests feed the CPU oracle torch's own normalize / exp / sigmoid of the model's parameters and compare integer buffers and
# projection floats BIT FOR BIT, so the Tracer they build takes the activated tensors like the reference's (tracer.py:323-327).  The
# product default — a model with the reference's activations hands its raw tensors over and the library activates them in-kernel,
# equal up to the last bits of expf — is tested where it is switched on explicitly (tests/test_gpu_parity.py::test_raw_parameter_*).
os.environ.setdefault("GUT_TRACER_RAW_PARAMETERS", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gut():
    import importlib
    return importlib.import_module("3dgrut_amd")
