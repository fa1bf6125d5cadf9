"""One-off soak: the parity fuzz tests of tests/test_gpu_fuzz.py with many more seeds than the suite runs."""
import os, sys, time, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as T
from tiebrush_amd import api
ctx = api.Context(0)
lo, hi = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
names = [n for n, f in inspect.getmembers(T, inspect.isfunction) if n.startswith("test_fuzz") and list(inspect.signature(f).parameters) == ["ctx", "seed"]]
print("tests:", names, flush=True)
for seed in range(lo, hi):
    for n in names:
        try:
            getattr(T, n)(ctx, seed)
        except AssertionError as e:
            print("FAIL", n, "seed", seed, str(e)[:300], flush=True)
    if (seed - lo) % 10 == 9: print("seed", seed, "%.0f s" % (time.time() - t0), flush=True)
print("done")
