#!/usr/bin/env python3
"""Soak of the `tiebrush` command line after ABI 8 (kept results, two encode threads, chunks of any size, the hybrid decode and its
hand-overs): random inputs, random options, random routes through the tool (TBK_* switches) against oracle/_build/tb_cpu_e2e — the
reference's main loop on the repo's host codec around the oracle — record stream for record stream.
    python tools/soak_cli.py [runs] [seed]
Prints one line per mismatch and a summary; exit code 1 on any mismatch."""
import atexit, gzip, os, random, shutil, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tiebrush_amd import synth

TB = os.path.join(ROOT, "tiebrush_amd", "_build", "tiebrush")
CPU = os.path.join(ROOT, "oracle", "_build", "tb_cpu_e2e")


def records(path):
    """the BAM's record stream (everything behind the header), as bytes"""
    raw = gzip.decompress(open(path, "rb").read())
    l_text = int.from_bytes(raw[4:8], "little")
    o = 8 + l_text
    n_ref = int.from_bytes(raw[o:o + 4], "little")
    o += 4
    for _ in range(n_ref):
        l_name = int.from_bytes(raw[o:o + 4], "little")
        o += 4 + l_name + 4
    return raw[o:]


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20261005)
    d = tempfile.mkdtemp(prefix="tbk_soak_", dir="/tmp")
    atexit.register(shutil.rmtree, d, True)
    bad = 0
    for r in range(runs):
        k = rng.choice([2, 3, 5, 8, 12, 33, 70])
        reads = rng.choice([300, 2000, 9000, 40000])
        prof = rng.choice(["c2", "c3", "c5"])
        tile = synth.make_tile(k, reads, prof, n_loci=rng.choice([5, 60, 2000]), seed_base=rng.randrange(1 << 30))
        seq = rng.random() < 0.6
        for f in os.listdir(d):
            p = os.path.join(d, f)
            shutil.rmtree(p, True) if os.path.isdir(p) else os.remove(p)
        paths = synth.write_bams_fast(tile, os.path.join(d, "in"), seq=seq)
        flags = []
        s = rng.random()
        if s < 0.25:
            flags.append("-P")
        elif s < 0.5:
            flags.append("-E")
        if rng.random() < 0.3:
            flags.append("-S")
        if rng.random() < 0.3:
            flags.append("--keep-secondary")
        if rng.random() < 0.3:
            flags += ["-N", str(rng.choice([1, 2, 5]))]
        if rng.random() < 0.3:
            flags += ["-Q", str(rng.choice([1, 20, 40]))]
        env = dict(os.environ)
        route = []
        if rng.random() < 0.5:
            env["TBK_HYBRID"] = "1"
            env["TBK_HYBRID_SHARE"] = str(rng.choice([20, 50, 80]))
            route.append("hybrid%s" % env["TBK_HYBRID_SHARE"])
        elif rng.random() < 0.3:
            env["TBK_DEVICE_DECODE"] = "1"
            route.append("devdecode")
        elif rng.random() < 0.3:
            env["TBK_TILE_RECORDS"] = str(rng.choice([500, 5000, 50000]))
            route.append("tiles%s" % env["TBK_TILE_RECORDS"])
        if rng.random() < 0.7:
            env["TBK_DW_CHUNK_GROUPS"] = str(rng.choice([64, 200, 1000, 5000]))
            route.append("chunk%s" % env["TBK_DW_CHUNK_GROUPS"])
        if rng.random() < 0.3:
            env["TBK_DW_ONE_ENCODER"] = "1"
            route.append("1enc")
        if rng.random() < 0.25:
            env["TBK_NO_KEEP_RESULTS"] = "1"
            route.append("nokeep")
        if rng.random() < 0.2:
            env["TBK_NO_WARMUP"] = "1"
            route.append("nowarm")
        if rng.random() < 0.25:
            env["TBK_TEST_DW_REFUSE_CHUNK"] = str(rng.choice([0, 1, 3]))
            route.append("refuse%s" % env["TBK_TEST_DW_REFUSE_CHUNK"])
        if rng.random() < 0.15:
            flags += ["--writer", "host"]
        out, ref = os.path.join(d, "out.bam"), os.path.join(d, "ref.bam")
        a = subprocess.run([TB, "-o", out] + flags + paths, capture_output=True, text=True, env=env)
        cpu_flags = [f for f in flags if f not in ("--writer", "host")]
        b = subprocess.run([CPU, "-o", ref] + cpu_flags + paths, capture_output=True, text=True)
        desc = "run %d: %d files x %d reads %s seq=%d %s [%s]" % (r, k, reads, prof, seq, " ".join(flags), " ".join(route))
        if a.returncode != 0 or b.returncode != 0:
            bad += 1
            print("FAIL", desc, "rc", a.returncode, b.returncode, (a.stderr or b.stderr)[-300:].replace("\n", " | "), flush=True)
            continue
        ra, rb = records(out), records(ref)
        if ra != rb:
            bad += 1
            print("MISMATCH", desc, len(ra), len(rb), flush=True)
        elif r % 10 == 0:
            print("ok  ", desc, len(ra), flush=True)
    print("soak_cli: %d runs, %d bad" % (runs, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
