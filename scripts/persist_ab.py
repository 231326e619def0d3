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
for (m, n) in [tuple(int(v) for v in s.split("x")) for s in os.environ.get("AB_SIZES", "256x512,512x1024,1024x2048,1024x4096").split(",")]:
    A, b, c = synth.dense_lp(m, n, 12345)
    row = {"m": m, "n": n}
    objs = []
    for mode in (0, 2, 0, 2):  # 2 = resident-tableau path with its size cap lifted
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
    cyc = (C.c_ulonglong * 64)()
    api.persist_cycles(cyc)
    tot = list(cyc)
    row["cycles_per_pivot_cumulative"] = {k: tot[i] / max(1, tot[4]) for i, k in enumerate(("propose", "gather", "read", "apply"))}
    row["cycles_per_pivot_cumulative"]["sweeps"] = tot[5] / max(1, tot[4])
    row["cycles_per_pivot_cumulative"]["first_sweep"] = tot[6] / max(1, tot[4])
    M = (1 << 64) - 1
    row["spread_last_launches"] = {"work_max": tot[8], "work_min": M - tot[9], "gather_max": tot[10], "gather_min": M - tot[11]}
    row["frac_hbm_roofline_persist"] = 16 * (m + 1) * (n + 1) / (row["persist_us_per_pivot"] * 1e-6) / 8e12
    print(json.dumps(row), flush=True)
