import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "icm-slam_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long CPU oracle runs")


def _native_build():
    try:
        from icmslam_hip import _lib
        lib = _lib.load()
        return "libicmslam_hip: %s  build %s  (%s)" % (lib.icm_version().decode(), lib.icm_build_id().decode(), _lib.LIB_PATH)
    except Exception as e:  # a missing library is the tests' own business to report
        return "libicmslam_hip: not loadable (%s)" % e


def pytest_report_header(config):
    """Every test log names the native build it ran on (icm_version / icm_build_id of the library the tests load), so a
    red log can be tied to a build."""
    return _native_build()


def pytest_terminal_summary(terminalreporter):
    terminalreporter.write_line(_native_build())   # (-q suppresses the header: the summary carries it too)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
