"""Warm-started child solves (bs.cpp:274-288) on a large dense LP: root to optimality, then for a few basic
columns the two branching children (UP floor / LO ceil on a clone), timed per dual pivot.
usage: childtime.py M N [CHILDREN]"""
import os, sys, time, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, capi
m, n = int(sys.argv[1]), int(sys.argv[2])
kids = int(sys.argv[3]) if len(sys.argv) > 3 else 6
api = mvolps_amd.api()
A, b, c = synth.dense_lp(m, n, 12345)
P = api.create(); P.load_dense(A, b, c)
t = time.perf_counter(); P.simplex(); api.sync(); el = time.perf_counter() - t
print(json.dumps({"root_pivots": P.it_cnt, "root_s": el, "us_per_pivot": el / max(1, P.it_cnt) * 1e6}), flush=True)
x = np.array(P.col_prim())
frac = [j for j in range(1, n + 1) if abs(x[j - 1] - round(x[j - 1])) > 1e-6][:kids]
tot_p = tot_t = 0
for j in frac:
    for typ, lo, hi in ((capi.UP, 0.0, math.floor(x[j - 1])), (capi.LO, math.ceil(x[j - 1]), 0.0)):
        Q = P.copy(); api.sync()
        it0 = Q.it_cnt
        api.set_col_bnds(Q.h, j, typ, lo, hi)
        t = time.perf_counter(); Q.simplex(); api.sync(); el = time.perf_counter() - t
        piv = Q.it_cnt - it0
        tot_p += piv; tot_t += el
        print(json.dumps({"col": j, "type": typ, "pivots": piv, "ms": el * 1e3, "us_per_pivot": el / max(1, piv) * 1e6, "status": Q.status}), flush=True)
        del Q
print(json.dumps({"children": 2 * len(frac), "pivots": tot_p, "us_per_pivot": tot_t / max(1, tot_p) * 1e6}))
