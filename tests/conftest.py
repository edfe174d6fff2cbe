import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle's thread pool: the GPU box grants a 16-core share of a much larger host, and torch's default (one thread per
    # host core) oversubscribes it — the F = 4 oracle passes of the gradient gates took 20 s each there against 2 s in the 8-core
    # build container
    try:
        import torch
        n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        torch.set_num_threads(max(1, min(16, n)))
    except Exception:
        pass


@pytest.fixture(scope='session', autouse=True)
def _native_stack_on_fatal_signals():
    """GPU runs: a SIGABRT / SIGSEGV inside the HIP runtime, RCCL or libaddk leaves its NATIVE frames on stderr (faulthandler, which pytest
    installs, shows Python frames only — two aborts of earlier rounds could not be explained for lack of them: DESIGN.md §7)."""
    try:
        import torch
        if torch.cuda.is_available():
            import addk
            addk.load().addk_debug_trace_fatal_signals()
    except Exception:
        pass
    yield


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz'))
        return cache[name]
    return load
