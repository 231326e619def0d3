"""Kernel-trace input: window-64 B&B on the 512x1024 ILP (run under rocprofv3 --kernel-trace, then trace_busy.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
from tests import lpgen
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3)
bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=64)
t = time.perf_counter()
r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=2000, window=64)
print("wall ms", (time.perf_counter() - t) * 1e3, r["count"], r["total_pivots"])
