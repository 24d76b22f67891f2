import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def engine():
    """The product library on cuda:0.  No fallback: a missing library or device is a test ERROR."""
    # torch first: it brings its own copy of the HIP runtime, and whichever copy is loaded first serves the whole process.  With
    # librustybam_amd.so (linked against /opt/rocm) loaded first, a later torch.cuda initialisation in the same process finds no
    # device (tests/test_gpu_fullsize.py uses torch for device memory and segment sums).
    import torch  # noqa: F401
    import rustybam_amd
    eng = rustybam_amd.Engine(0)
    yield eng
    eng.close()
