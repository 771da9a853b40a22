import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(autouse=True)
def _give_gpu_memory_back():
    """tests at the BASELINE sizes leave hundreds of GB in torch's caching allocator; the multi-process tests that follow
    start child processes on the same GPU, which would find it full"""
    yield
    import gc
    if "torch" in sys.modules:
        import torch
        if torch.cuda.is_available():
            gc.collect()
            torch.cuda.empty_cache()
