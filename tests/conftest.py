import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    if os.environ.get('CTR_SEGV_TRACE'):      # debug aid: native stack on SIGSEGV (run with -p no:faulthandler)
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, 'tools', 'repro', 'libsegvtrace.so')).segv_trace_install()
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def sess():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    from recommendation_amd import Session
    return Session('cuda:0', seed=7)
