"""One-off soak of the device deflate (tbk_bgzf_deflate): payloads built from random segments of every kind the encoder treats differently —
noise, small alphabets, repeats at a fixed distance (near and at the 32 KiB reach), long runs, BAM-like records — in random order and length,
each run checked member by member against zlib (inflate, CRC32, ISIZE)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from test_gpu_deflate import check_run
from tiebrush_amd import api
ctx = api.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
def seg():
    k = int(rng.integers(0, 7))
    n = int(rng.choice([1, 2, 3, 7, 31, 258, 259, 1000, 5000, 40000, 70000]))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 1: return rng.integers(0, int(rng.choice([2, 3, 4, 5, 16])), n, dtype=np.uint8).tobytes()
    if k == 2: return bytes([int(rng.integers(0, 256))]) * n
    if k == 3:
        d = int(rng.choice([1, 2, 3, 4, 8, 255, 256, 4096, 32767, 32768, 32769]))
        base = rng.integers(0, 256, min(d, 70000), dtype=np.uint8).tobytes()
        return (base * (n // len(base) + 2))[:n + len(base)]
    if k == 4:
        rec = b"".join(int(rng.integers(0, 1 << 20)).to_bytes(4, "little") + b"\x01\x00\x64\x00" + rng.integers(0, 16, 50, dtype=np.uint8).tobytes() + b"I" * 100 for _ in range(max(1, n // 160)))
        return rec
    if k == 5: return bytes(range(256)) * (n // 256 + 1)
    return b""
total = nruns = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 200):
    p = b"".join(seg() for _ in range(int(rng.integers(1, 9))))
    if not p: continue
    run = ctx.bgzf_deflate(p)
    check_run(run, p)
    total += len(p); nruns += 1
    if it % 50 == 0:
        print(it, nruns, total, len(run)); sys.stdout.flush()
print("soak ok:", nruns, "runs,", total, "bytes")
