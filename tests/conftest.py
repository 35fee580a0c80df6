import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "job_form(mode): form of the jobs a test creates (bbs_ctx_set_latency_mode): False = "
                            "throughput (the default of the suite), True = latency, None = the library's AUTO")


import pytest  # noqa: E402


@pytest.fixture(autouse=True)
def _job_form(request):
    """Every case runs with one known job form: the throughput form unless the test is marked job_form(True / None)."""
    tests_dir = os.path.dirname(os.path.abspath(__file__))
    if tests_dir not in sys.path:
        sys.path.insert(0, tests_dir)
    try:
        import parity_cases as pc
    except Exception:                      # a test module that does not use the engine (missing library: its own error)
        yield
        return
    m = request.node.get_closest_marker("job_form")
    pc.LATENCY_MODE = m.args[0] if m else False
    yield
    pc.LATENCY_MODE = None
