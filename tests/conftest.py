import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the oracle (fp64 torch on the host) is the slow half of the -m gpu suite: a GPU box reports every core of the machine
    # while the job owns a 16-core share, and torch's default of one thread per reported core made those tests 10x slower
    import torch
    torch.set_num_threads(max(1, min(int(os.environ.get('VKAS_TEST_THREADS', '16')), os.cpu_count() or 1)))


def pytest_collection_modifyitems(config, items):
    import torch
    # device_count() does not initialise HIP on this image (is_available() does): the multi-process GPU test starts its
    # ranks from a parent that has not touched the GPU
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason='no GPU in this environment')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    from tests import parity_log
    path = parity_log.write_report()
    if path:
        print('\nparity report:', path)


def pytest_runtest_logreport(report):
    """VKAS_TEST_TIMES=<file>: append "<seconds> <phase> <test id>" as each test phase finishes (suite-time budgeting: the
    driver runs the whole -m gpu suite in one go)."""
    path = os.environ.get('VKAS_TEST_TIMES')
    if path and report.duration > 0.5:
        with open(path, 'a') as f:
            f.write('%8.1f %-8s %s %s\n' % (report.duration, report.when, report.outcome, report.nodeid))
