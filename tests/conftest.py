import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--runslow", action="store_true", default=False,
                     help="also run the cases marked slow (minutes of float64 CPU oracle each); DSEE_RUN_SLOW=1 does the same")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size float64 oracle cases kept out of the default run (--runslow / DSEE_RUN_SLOW=1: "
                                       "the nightly form of the suite)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--runslow") or os.environ.get("DSEE_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow case: run with --runslow or DSEE_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _release_device_state(request):
    """GPU tests build dozens of TrainerManagers, each with captured hipGraphs, a private graph memory pool and static input
    buffers: release them when the test that made them ends (DSEE_TEST_KEEP_MANAGERS=1 keeps the old behaviour: whenever the
    garbage collector gets to them), so that the suite's device state does not grow with the number of tests run before."""
    yield
    if request.node.get_closest_marker("gpu") is None or os.environ.get("DSEE_TEST_KEEP_MANAGERS") == "1":
        return
    import gc
    from deepsee_amd.managers import TrainerManager
    TrainerManager.close_all()
    gc.collect()


def pytest_runtest_logreport(report):
    """DSEE_TEST_DURATIONS=<file>: one line per finished test phase (a crash of the interpreter loses --durations)."""
    path = os.environ.get("DSEE_TEST_DURATIONS")
    if path and report.when == "call":
        with open(path, "a") as f:
            f.write("%8.2f  %s  %s\n" % (report.duration, report.outcome, report.nodeid))
