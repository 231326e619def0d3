"""Throughput of mvx_simplex_batch vs one-by-one solves on B&B-style children (512x1024 ILP)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, capi
from tests import lpgen

api = mvolps_amd.api()
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
A, b, c, U = synth.dense_ilp(m, n, 12345, 3)
P = lpgen.load_ilp(api, A, b, c, U)
P.simplex()
x = P.col_prim()
frac = [j + 1 for j in range(n) if np.trunc(x[j]) != x[j]]
print("root pivots", P.it_cnt, "fractional", len(frac))

def children(k):
    out = []
    for j in frac[:k]:
        ch = P.copy()
        api.set_col_bnds(ch.h, j, capi.UP, 0.0, float(np.floor(x[j - 1])))
        out.append(ch)
    return out

# warm up so that steady-state throughput is what is printed
for g in (1, 0):
    kids = children(16)
    api.simplex_batch((C.c_void_p * 16)(*[k.h for k in kids]), 16, None, None)
    for ch in children(2):
        ch.simplex()
for graphs, K in [(g, K) for g in (16, 32, 64) for K in (16, 32, 64, 80)]:
    api.set_batch_slots(graphs)
    K = min(K, len(frac))
    kids = children(K)
    api.sync()
    t = time.perf_counter()
    for ch in kids:
        ch.simplex()
    t_seq = time.perf_counter() - t
    piv = sum(ch.it_cnt - P.it_cnt for ch in kids)
    kids2 = children(K)
    arr = (C.c_void_p * K)(*[k.h for k in kids2])
    api.sync()
    t = time.perf_counter()
    api.simplex_batch(arr, K, None, None)
    t_b = time.perf_counter() - t
    assert all(a.obj == b_.obj for a, b_ in zip(kids, kids2))
    print("slots=%d K=%2d pivots=%5d  sequential %.2f ms (%.1f us/pivot)   batch %.2f ms (%.1f us/pivot)  speedup %.2fx" % (
        graphs, K, piv, t_seq * 1e3, t_seq * 1e6 / piv, t_b * 1e3, t_b * 1e6 / piv, t_seq / t_b))
