import os, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_reference.npz'))


def free_port():
    """A TCP port that is free right now (rendezvous of the multi-process tests: a fixed port collides with the TIME_WAIT
    socket of a run that ended seconds ago)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


@pytest.fixture(scope='session')
def golden_data():
    """G8-G11: byte / index / flag arithmetic of the reference's prepare.py, data.py, main.py, tcr.py (make_golden_data.py)."""
    import numpy as np
    return np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_data.npz'))
