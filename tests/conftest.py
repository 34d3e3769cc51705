import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions leave gpurun_out/parity_report.json: per fixture group, how many cases met north_star's 1e-5 and the
    worst error seen (pytest -q hides prints; the builder copies the file to profiles/)."""
    try:
        import parity_util

        parity_util.write_report(ROOT)
    except Exception:
        pass
