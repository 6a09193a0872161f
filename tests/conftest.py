import os
import sys
import pytest

# The CPU oracle (and be=cpu) are OpenMP code checking SMALL cases: on the GPU box's 128+ hardware threads every parallel region of a batch-2 layer costs
# ~100 ms of fork / join across the whole machine (measured: oracle conv_fwd 183 ms per call there against 20 ms on 8 cores -- 250 s of a 670 s suite).
# A bounded team is faster for everything the tests run; set before any OpenMP runtime loads.  (bench.py subprocesses drop these again: tests/test_gpu_zz_bench.py.)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: longer CPU test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Order of the -m gpu files (the driver runs `pytest -m gpu -x`): hot-path parity first, the reference's own kernels (F4) next, then the callers
# either side of the path, and every non-parity test (timing attribution, bench.py subprocesses) last -- so that nothing which is not a parity
# test can stop the run before a parity test has executed.  Files not named here keep their alphabetical place between the two groups.
_GPU_ORDER = ["test_gpu_parity.py", "test_gpu_ref_cucl.py", "test_gpu_nhwc.py", "test_gpu_fullnet.py", "test_gpu_multi.py", "test_gpu_adapter.py",
              "test_gpu_cnn_op_info.py"]
_GPU_LAST = ["test_gpu_zz_properties.py", "test_gpu_zz_bench.py"]


def pytest_collection_modifyitems(session, config, items):
    def key(it):
        fn = os.path.basename(str(it.fspath))
        if fn in _GPU_ORDER:
            return (1, _GPU_ORDER.index(fn))
        if fn in _GPU_LAST:
            return (3, _GPU_LAST.index(fn))
        return (2 if fn.startswith("test_gpu_") else 0, 0)
    items.sort(key=key)   # stable: the order inside a file is kept
