import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def host_threads() -> int:
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (os.cpu_count() reports the
    whole host on a GPU box whose container owns 16 cores: 8x oversubscription of the oracle's CPU runs)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 32))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    import torch
    torch.set_num_threads(host_threads())


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


@pytest.fixture(autouse=True)
def _fresh_engine_state():
    """Process-wide heuristics of the engine must not leak from one test into the next: the 'device memory is short, stop
    retaining pre-split images' switch (ops.retain_ok) stays on for the rest of a PROCESS once a large test tripped it."""
    yield
    import sys
    ops = sys.modules.get("glfusion_amd.ops")
    if ops is not None:
        ops._retain_off.clear()
        ops._retain_step.clear()
