import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


import pytest

# The seed sorts have two forms (job_internal.hpp, seed_lds_finish): four global passes, or -- from 2**30 records on -- three and the
# last code byte in the candidate kernels' LDS.  The small inputs of these modules would only ever see the first: they run the second
# (CORA_SEED_LDS=2 = always) unless a test chooses itself; the tests named in BOTH_SEED_SORTS run both.  test_gpu_large.py and
# test_gpu_bench.py keep the library's own choice.
_FORCE_LDS_FINISH = {"test_gpu_parity", "test_gpu_edges", "test_gpu_sharded", "test_gpu_cora_driver", "test_gpu_dist"}
BOTH_SEED_SORTS = pytest.mark.parametrize("seed_sort", ["four passes", "three passes + LDS"])


@pytest.fixture(autouse=True)
def _seed_sort_form(request, monkeypatch):
    mod = request.module.__name__.rsplit(".", 1)[-1]
    if "seed_sort" in getattr(request, "fixturenames", ()) and hasattr(request.node, "callspec") and "seed_sort" in request.node.callspec.params:
        monkeypatch.setenv("CORA_SEED_LDS", "0" if request.node.callspec.params["seed_sort"] == "four passes" else "2")
    elif mod in _FORCE_LDS_FINISH and "CORA_SEED_LDS" not in os.environ:
        monkeypatch.setenv("CORA_SEED_LDS", "2")
    yield
