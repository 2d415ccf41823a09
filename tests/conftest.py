import os
import sys

import pytest

# the run-time compiler's disk cache stays out of the test session: what the JIT tests exercise is the compile itself (test_gpu_jit.py has the cache's own test)
os.environ.setdefault("DFDB_JIT_CACHE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dataframedbs.jl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (test infrastructure)."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def dfdb_mod():
    # (see `ctx` below: torch is imported — and its HIP runtime loaded — before libdfdb_hip.so, whichever fixture a test asks for first)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    import dfdb
    return dfdb


@pytest.fixture(scope="session")
def ctx(dfdb_mod):
    """One engine context on cuda:0 for the whole GPU session.  Fails loudly without the HIP library / a GPU."""
    import torch  # torch bundles its own libamdhip64.so, with the soname libamdhip64.so.7 that libdfdb_hip.so asks for: loaded FIRST, the engine binds to that
    # copy and the process holds ONE HIP runtime (streams, events and RCCL handles are shared: bench.py relies on it).  Loaded second, torch's libraries (which
    # ask for "libamdhip64.so") bring a second copy that finds "No HIP GPUs" once the first holds /dev/kfd.  dfdb_mod above imports torch first for the same reason.
    torch.cuda.init()
    c = dfdb_mod.default_context(0)
    if os.environ.get("DFDB_TEST_JIT") == "1":       # soak: every interpreter program of every test runs as its hipRTC-compiled kernel (0.2-0.4 s per new shape)
        c.set_option("jit", 2)
        c.set_option("jit_min_rows", 0)
    return c
