"""Sweep the streamed-update tuning knobs on one dense LP; prints ms/pivot and k_fb average per variant."""
import json
import sys
import time

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth

m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8192)
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
api = mvolps_amd.api()
mvolps_amd.require_device()
A, b, c = synth.dense_lp(m, n, 12345)
ref = None
variants = [(0, 1, 0), (8, 1, 0), (16, 1, 0), (32, 1, 0), (8, 1, 1), (16, 1, 1), (32, 1, 1), (0, 1, 0)]
for (tr, hot, nt) in variants:
    api.set_tuning(tr, hot, nt)
    P = api.create()
    P.load_dense(A, b, c)
    P.simplex(it_lim=30)
    api.sync()
    t = time.perf_counter()
    P.simplex(it_lim=steps)
    api.sync()
    el = time.perf_counter() - t
    api.profile_reset()
    api.profile_enable(1)
    P.simplex(it_lim=100)
    api.profile_enable(0)
    kms = api.profile_update_ms() / max(1, api.profile_update_launches())
    obj = P.obj
    if ref is None:
        ref = obj
    print(json.dumps({"tr": tr, "hot": hot, "nt": nt, "us_per_pivot": el / steps * 1e6, "k_fb_us": kms * 1e3,
                      "GBps": 16 * (m + 1) * (n + 1) / (kms * 1e-3) / 1e9, "same_bits": obj == ref}), flush=True)
    del P
