import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a box without a GPU skips the gpu-marked tests instead of failing them one by one
    (they need the HIP library on a device; `-m "not gpu"` is the CPU suite).  On a GPU box nothing is skipped: a missing
    or stale HIP extension there makes every gpu test FAIL in _capi.hip_api()."""
    gpu_items = [it for it in items if it.get_closest_marker('gpu') is not None]
    if not gpu_items:
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible (torch.cuda.is_available() is False)')
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope='session')
def oracle():
    from tests._helpers import oracle_lib
    return oracle_lib()


def pytest_terminal_summary(terminalreporter):
    """observed parity margins: for every tolerance-based fixture check, the largest excess over the rtol term that
    was actually seen next to the atol it is held to"""
    from tests._helpers import PARITY_LOG
    if not PARITY_LOG:
        return
    terminalreporter.write_sep('-', 'parity margins (max |got - want| - rtol |want|   vs   atol)')
    for k in sorted(PARITY_LOG):
        ex, atol = PARITY_LOG[k]
        terminalreporter.write_line('%-58s %10.3e   %8.1e' % (k, max(ex, 0.0), atol))
