import os
import sys

# OpenMP runtimes (torch's and the oracle's C loops) read this when they are loaded: idle threads sleep instead of spinning
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _host_threads():
    """The oracle's float64 work (NumPy / OpenBLAS, torch's CPU kernels in the audits' comparisons, the C loops) is sized to this
    process's CPU share: the GPU boxes show 256 logical CPUs to a process that may use 16 of them, and one thread per visible
    CPU costs more in contention than it computes."""
    from oracle.ops import cpu_share
    n = cpu_share()
    limits = None
    try:
        import torch
        torch.set_num_threads(n)
    except Exception:
        pass
    try:
        from threadpoolctl import threadpool_limits
        limits = threadpool_limits(limits=n)          # held for the session
    except Exception:
        pass
    yield n
    del limits


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def ws(device):
    from adunet_amd import ops
    return ops.Workspace(device)


def free_port() -> int:
    """A TCP port that is free on 127.0.0.1 right now (rendezvous of the multi-process tests: fixed port numbers collide
    with a neighbour's run or with a socket of the previous test still in TIME_WAIT)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]
