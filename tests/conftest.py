import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["f32", "bf16x6", "f16x3"])
def precision(request):
    """Runs a GPU test under every contraction precision: exact fp32 MFMA, split-bf16 (bf16x6), scaled split-fp16 (f16x3)."""
    from glfusion_amd import ops
    ops.set_precision(request.param)
    yield request.param
    ops.set_precision("f32")
