import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'oracle: compares the HIP path with oracle/ or the reference fixtures; the kernel template '
                                       'instances such a test launches are recorded (tests/test_step_coverage_gpu.py)')
    config._dep_instances = {}          # {test nodeid: set of launch instances} of the oracle-marked tests of this session


@pytest.fixture(autouse=True)
def _record_launch_instances(request):
    """While an `oracle`-marked GPU test runs, the library's launch-instance log (dep_instance_log_*) is on; what it launched is
    kept per test for tests/test_step_coverage_gpu.py (in-process launches only: subprocess drivers are bit-identity / stress tests)."""
    if request.node.get_closest_marker('oracle') is None or request.node.get_closest_marker('gpu') is None or not has_gpu():
        yield
        return
    from icassp2022_depression_amd import _lib as L
    L.instance_log_enable(True)
    yield
    inst = L.instance_log_read(reset=True)
    L.instance_log_enable(False)
    request.config._dep_instances.setdefault(request.node.nodeid, set()).update(inst)


def load_golden(name):
    """npz fixture -> nested dict ('sd/ln.weight' -> d['sd']['ln.weight'])."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    out = {}
    for k in z.files:
        if '/' in k:
            a, b = k.split('/', 1)
            out.setdefault(a, {})[b] = z[k]
        else:
            out[k] = z[k]
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
