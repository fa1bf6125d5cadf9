"""One-off soak of the sort path variants (forced run-merge / radix) with many seeds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T
from tiebrush_amd import api
class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k): os.environ.pop(k, None)
ctx = api.Context(0)
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        T.test_fuzz_collapse_forced_run_sort(ctx, seed, MP())
    except AssertionError as e:
        bad += 1; print("FAIL seed", seed, str(e)[:200], flush=True)
    os.environ.pop("TBK_DEBUG", None)
print("sort soak done, failures:", bad)
