"""debug helper: one seed of test_random_risky_queue_streamed_and_sharded with the engine's error text"""
import sys, os, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "dataframedbs.jl_amd"), os.path.join(ROOT, "tests")]
import pytest
seed = sys.argv[1] if len(sys.argv) > 1 else "33"
import test_gpu_fuzz as F
_orig = F.outcome
def outcome(fn):
    try:
        return ("ok", fn())
    except Exception as e:
        print("OUTCOME EXC:", type(e).__name__, e, file=sys.stderr)
        return ("err", type(e).__name__)
F.outcome = outcome
sys.exit(pytest.main(["-x", "-q", "-m", "gpu", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"), "-k", f"risky_queue_streamed and [{seed}]", "-s"]))
