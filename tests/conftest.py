import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver via gpurun)")


@pytest.fixture(scope="session")
def golden_ops():
    from safetensors.torch import load_file

    return load_file(os.path.join(GOLDEN, "ops.safetensors"))


@pytest.fixture(scope="session")
def golden_model():
    from safetensors.torch import load_file

    return load_file(os.path.join(GOLDEN, "wan-tiny_model.safetensors"))


@pytest.fixture(scope="session")
def golden_sched():
    from safetensors.torch import load_file

    return load_file(os.path.join(GOLDEN, "scheduler.safetensors"))
