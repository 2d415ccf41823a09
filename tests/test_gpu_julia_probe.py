"""GPU: executes the Julia shim (dataframedbs.jl_amd/julia/probe.jl) against the reference's own Julia path — if and only if a `julia`
binary with DataFrameDBs.jl installed exists on the box.  The build image has neither, so this is skipped there; the FFI is pinned
statically instead (tests/test_host_cpu.py::test_julia_shim_ccalls_match_the_header, julia/STATIC_REVIEW.md)."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_julia_shim_runtime_probe(ctx, tmp_path):
    julia = shutil.which("julia")
    if not julia:
        pytest.skip("no julia binary on this box (shim verified statically: julia/STATIC_REVIEW.md)")
    have = subprocess.run([julia, "-e", "using DataFrameDBs, DataFrames"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    if have.returncode != 0:
        pytest.skip("julia is here but DataFrameDBs.jl / DataFrames.jl are not installed")
    p = subprocess.run([julia, os.path.join(ROOT, "dataframedbs.jl_amd", "julia", "probe.jl"), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1800)
    assert p.returncode == 0 and b"PROBE OK" in p.stdout, p.stdout.decode(errors="replace")[-4000:]
