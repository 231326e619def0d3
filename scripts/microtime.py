"""Wall-clock cost of the individual C-ABI calls on a small ILP (host overheads, not kernel time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, capi, bnb
from tests import lpgen

api = mvolps_amd.api()
def T(label, fn, reps=1):
    t = time.perf_counter(); r = None
    for _ in range(reps): r = fn()
    dt = (time.perf_counter() - t) / reps
    print("%-40s %10.1f us" % (label, dt * 1e6), flush=True)
    return r

for (m, n) in [(16, 32), (512, 1024)]:
    print("---- %dx%d" % (m, n))
    A, b, c, U = synth.dense_ilp(m, n, 5, 2)
    P = T("load_ilp", lambda: lpgen.load_ilp(api, A, b, c, U))
    T("first simplex (build+solve)", lambda: P.simplex())
    T("re-solve (0 pivots)", lambda: P.simplex(), 5)
    Q = T("copy_prob", lambda: P.copy(), 1)
    T("copy_prob x5", lambda: P.copy(), 5)
    x = P.col_prim()
    frac = [j + 1 for j in range(n) if np.trunc(x[j]) != x[j]]
    pick = frac[0]
    T("set_col_bnds", lambda: api.set_col_bnds(Q.h, pick, capi.UP, 0.0, float(np.floor(x[pick - 1]))))
    T("child simplex", lambda: Q.simplex())
    print("child pivots", Q.it_cnt - P.it_cnt)
    T("get_col_prim x n", lambda: Q.col_prim())
    T("delete", lambda: Q.__del__())
    t = time.perf_counter()
    r = bnb.branch_and_bound(lpgen.load_ilp(api, A, b, c, U), quirks=0, max_nodes=40)
    dt = time.perf_counter() - t
    print("bnb 40 nodes: %.1f ms, %.1f us/node, pivots %d" % (dt * 1e3, dt * 1e6 / max(1, r["count"]), r["total_pivots"]))
