import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _gpu_objects_die_at_test_boundaries(request):
    """GPU tests: cyclic garbage (a dropped engine with its hipGraphs, launch plans, private memory pools, streams and
    events) is collected HERE — after the test, with the device idle — instead of whenever the allocation counter of the
    interpreter says so.  Twice in ~12 full runs of the suite (both times the first process on a fresh box) the
    interpreter aborted inside a garbage collection that hit while the NEXT engine was running
    (tests/test_engine_e2e.py::test_hip_engine_graph_decode_equals_eager: faulthandler shows the main thread
    "Garbage-collecting" under an encode step); it never reproduced in 50 reruns of those tests, so the destructor at
    fault is not known — what is known is that the product's own serving loop runs with the collector off
    (engine/serve.py quiet_gc), and so now do the GPU tests."""
    if "gpu" not in request.keywords or not torch.cuda.is_available():
        yield
        return
    import gc
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.synchronize()
        if was:
            gc.enable()
