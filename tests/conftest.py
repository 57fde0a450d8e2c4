import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _quiescent_device(request):
    """GPU tests start and end on an idle device: a fault or a use-after-free of one test's asynchronous work
    surfaces in THAT test, not in whichever test happens to run next."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from yolo2_light_amd._lib import lib
    yield
    if lib.yl_device_count() > 0:
        assert lib.yl_device_synchronize(-1) == 0, "device work of this test failed asynchronously"


@pytest.fixture(scope="session")
def olib():
    import common
    return common.oracle_lib()
