import os
import sys

import pytest

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
    # (see `ctx` below: where there is a GPU, torch's copy of the HIP runtime opens it before the engine's does — whichever fixture a test asks for first)
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
    import torch  # torch brings its own copy of the HIP / HSA runtime (another soname than the system one libdfdb_hip.so links): both live in this process
    # ... and the copy that opens the GPU SECOND must be the system one: torch's runtime finds "No HIP GPUs" once the system runtime holds /dev/kfd, the other
    # order works.  Tests that use torch tensors (full-size buffers, bitmaps on the device) would otherwise depend on which test touched the GPU first.
    torch.cuda.init()
    c = dfdb_mod.default_context(0)
    if os.environ.get("DFDB_TEST_JIT") == "1":       # soak: every interpreter program of every test runs as its hipRTC-compiled kernel (0.2-0.4 s per new shape)
        c.set_option("jit", 2)
        c.set_option("jit_min_rows", 0)
    return c
