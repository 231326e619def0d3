"""Kernel-trace input: the driver's form of bench.py on the headline LP -- one 5-pivot call, then 20-pivot calls.
Run under rocprofv3 --kernel-trace --memory-copy-trace, then scripts/timeline.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8192)
api = mvolps_amd.api()
A, b, c = synth.dense_lp(m, n, 12345)
P = api.create()
P.load_dense(A, b, c)
P.simplex(it_lim=0)
P.simplex(it_lim=5)
for k in range(3):
    api.sync()
    t = time.perf_counter()
    P.simplex(it_lim=20)
    api.sync()
    print("20-pivot call: %.1f us = %.2f us per pivot" % ((time.perf_counter() - t) * 1e6, (time.perf_counter() - t) * 1e6 / 20))
