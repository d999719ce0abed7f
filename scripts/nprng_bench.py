#!/usr/bin/env python
"""raw draw rate of edmp_amd.nprng on this host: thread counts x block sizes x CPU placement (each setting in a fresh
process, because OpenMP reads its environment once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
cpus = os.environ.get("PIN_CPUS")
if cpus:
    a, b = cpus.split("-"); os.sched_setaffinity(0, range(int(a), int(b) + 1))
import numpy as np
from edmp_amd import nprng
np.random.seed(0)
nt = int(os.environ["NT"])
n = 358400 * 16
nprng.standard_normal((n,), nthreads=nt)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); nprng.standard_normal((n,), nthreads=nt); ts.append(time.perf_counter() - t0)
print("%%6.2f ns/normal best, %%6.2f median" %% (1e9 * min(ts) / n, 1e9 * sorted(ts)[2] / n))
''' % ROOT
for env in [dict(), dict(OMP_PROC_BIND="close", OMP_PLACES="cores"), dict(PIN_CPUS="0-15"), dict(PIN_CPUS="0-7"), dict(PIN_CPUS="32-47"), dict(OMP_PROC_BIND="spread", OMP_PLACES="cores")]:
    for blk in ("18", "15", "13"):
        for nt in ("4", "8", "16"):
            e = dict(os.environ, NT=nt, EDMP_NPRNG_BLOCK=blk, **env)
            r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
            print(f"{str(env):60s} block 2^{blk} threads {nt:>2s}: {r.stdout.strip() or r.stderr.strip()[-200:]}", flush=True)
