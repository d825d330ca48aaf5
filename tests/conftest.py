"""pytest configuration: the ``gpu`` marker, import paths, and the guard that keeps the GPU parity suite on the
hand-written kernels.

Every test marked ``gpu`` runs with ``RAYEN_STRICT_HIP=1`` and with the module's detour announcements
(``RuntimeWarning("rayen_amd: no HIP kernel serves ...")``) turned into errors: a kernel that starts refusing a shape
(``RAYEN_E_UNSUPPORTED``) fails the test instead of being answered by the packed torch evaluator on the device
libraries (rayen_amd/eager.py).  The few tests whose SUBJECT is that detour carry ``@pytest.mark.eager_detour``
and manage the environment themselves."""
import os
import sys
import warnings

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")
    config.addinivalue_line("markers", "eager_detour: the test exercises the loud device-library detour on purpose "
                                       "(no RAYEN_STRICT_HIP, warnings recorded by the test itself)")


@pytest.fixture(autouse=True)
def _gpu_tests_stay_on_the_kernels(request, monkeypatch):
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("eager_detour") is not None:
        yield
        return
    monkeypatch.setenv("RAYEN_STRICT_HIP", "1")
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=r"rayen_amd:", category=RuntimeWarning)
        yield


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")
