import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def _cpu_quota_cores():
    """Cores the cgroup lets this process use (bench.py's rule); None if uncapped."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else int(q) / int(per)
    except (OSError, ValueError):
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


# The oracle is host PyTorch.  On the GPU box torch sees 128 cores behind a 16-core cgroup quota: at the default thread count the oracle
# runs ~10 x slower than at the quota (bench.py's thread sweep: 568 vs 58 ms per step), which was most of the GPU suite's wall time.
_q = _cpu_quota_cores()
if _q and _q >= 1 and "OMP_NUM_THREADS" not in os.environ and (os.cpu_count() or 1) > _q:
    os.environ["OMP_NUM_THREADS"] = str(int(_q))       # before torch is imported; child processes of the tests inherit it


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")
    config.addinivalue_line("markers", "stress_gate: hand-over protocols under skew / foreign load (-m 'gpu and stress_gate')")


@pytest.fixture(scope="session")
def gold_dir():
    return GOLD


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no ROCm device")
    return torch.device("cuda:0")
