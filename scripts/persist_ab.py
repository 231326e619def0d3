"""Resident-tableau path (k_persist) against the two-kernel path (k_fa / k_fb) on cache-resident dense LPs:
us per pivot, same objective bits.  usage: persist_ab.py [pivots]"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
api = mvolps_amd.api()
mvolps_amd.require_device()
for (m, n) in ((256, 512), (512, 1024), (1024, 2048), (1024, 4096)):
    A, b, c = synth.dense_lp(m, n, 12345)
    row = {"m": m, "n": n}
    objs = []
    for mode in (0, 1, 0, 1):
        api.set_persist(mode)
        P = api.create()
        P.load_dense(A, b, c)
        P.simplex(it_lim=20)
        api.sync()
        t = time.perf_counter()
        P.simplex(it_lim=steps)
        api.sync()
        el = time.perf_counter() - t
        piv = P.it_cnt - 20
        key = "persist" if mode else "two_kernel"
        row[key + "_us_per_pivot"] = min(row.get(key + "_us_per_pivot", 1e9), el / max(1, piv) * 1e6)
        objs.append((P.obj, P.it_cnt))
        P.simplex()
        row[key + "_full"] = (P.obj.hex(), P.it_cnt, P.status)
    la, ab = C.c_longlong(0), C.c_longlong(0)
    api.persist_stats(C.byref(la), C.byref(ab))
    row["same_bits"] = len(set(objs)) == 1 and row["persist_full"] == row["two_kernel_full"]
    row["launches"], row["aborts"] = la.value, ab.value
    row["frac_hbm_roofline_persist"] = 16 * (m + 1) * (n + 1) / (row["persist_us_per_pivot"] * 1e-6) / 8e12
    print(json.dumps(row), flush=True)
