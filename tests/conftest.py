import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


CANARY = os.environ.get("VIDI_CANARY", "0") in ("1", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "hipgraph: captures a hipGraph (skipped under VIDI_CANARY: a capture cannot hipMalloc, and the guard-zone allocator does for every tensor)")
    if CANARY:
        import torch
        if torch.cuda.is_available():
            # guard-zone allocator (tests/canary/): must become torch's device allocator before the first device tensor of the session
            import canary
            canary.install()
            if os.environ["VIDI_CANARY"] == "2":        # ... and checked after EVERY C-ABI call (slow; names the call)
                from vidi_amd import hip

                def after_call(name, _c=canary):
                    n, msg, _, _ = _c.check()
                    if n:
                        _c.reset()
                        raise AssertionError(f"canary: {name} wrote outside a tensor: {msg}")
                hip.AFTER_CALL = after_call


@pytest.fixture(autouse=True)
def _canary_guard(request):
    """under VIDI_CANARY: the guard zones of every live tensor are compared with their pattern after each test"""
    yield
    if CANARY:
        import canary
        if canary.installed() and "canary_negative_control" not in request.node.name:
            n, msg, live, total = canary.check()
            if n:
                canary.reset()
                pytest.fail(f"canary: {n} allocation(s) with overwritten guard zones during {request.node.nodeid}: {msg}")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        if CANARY:
            skip_g = pytest.mark.skip(reason="VIDI_CANARY: hipGraph capture cannot allocate through the guard-zone allocator")
            for it in items:
                if "hipgraph" in it.keywords:
                    it.add_marker(skip_g)
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
