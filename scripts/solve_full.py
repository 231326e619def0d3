"""Solve the synthetic dense LPs to optimality on the GPU; print pivots, time, objective."""
import json
import sys
import time

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth

api = mvolps_amd.api()
mvolps_amd.require_device()
for (m, n, seed) in [(1024, 2048, 12345), (4096, 8192, 12345)]:
    A, b, c = synth.dense_lp(m, n, seed)
    P = api.create()
    t = time.perf_counter()
    P.load_dense(A, b, c)
    P.simplex(it_lim=0)  # builds the tableau on the host and uploads it (PCIe)
    api.sync()
    t_up = time.perf_counter() - t
    t = time.perf_counter()
    rc = P.simplex()
    el = time.perf_counter() - t
    print(json.dumps({"m": m, "n": n, "seed": seed, "rc": rc, "status": P.status, "obj": P.obj, "pivots": P.it_cnt,
                      "secs": el, "pivots_per_s": P.it_cnt / el, "load_build_upload_secs": t_up,
                      "pivots_per_s_incl_upload": P.it_cnt / (el + t_up), "device_ms": api.last_solve_ms(P.h)}), flush=True)
