#!/usr/bin/env python3
"""builds tools/micro/exit_cost.hip and reports, per configuration, the child's own run time and how long after its _exit the parent's wait returned"""
import os, subprocess, sys, time
here = os.path.dirname(os.path.abspath(__file__))
exe = "/tmp/exit_cost"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, os.path.join(here, "exit_cost.hip")], check=True)
for dev, pin, mode in ((0, 0, 0), (0, 0, 0), (16384, 0, 0), (0, 512, 0), (0, 512, 1), (0, 512, 2), (0, 512, 3), (0, 512, 4), (0, 2048, 0), (0, 2048, 1), (0, 2048, 2), (0, 2048, 3), (0, 2048, 4)):
    t0 = time.time()
    r = subprocess.run([exe, str(dev), str(pin), "1", str(mode)], capture_output=True, text=True)
    t1 = time.time()
    if r.returncode != 0:
        print(dev, pin, "failed", r.returncode); continue
    run, x = (float(v) for v in r.stdout.split())
    print("mode %d %s | device %6d MB pinned %5d MB: child's own %.3f s, start-up before main %.3f s, AFTER _exit %.3f s" % (mode, r.stderr.strip(), dev, pin, run, (x - run) - t0, t1 - x), flush=True)
