"""The bulk pass alone (HIP events around every k_fbc3 launch) by chain length and row-tile depth, and the whole pivot:
python scripts/bulktime.py [m n]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
m, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 8192)
api = mvolps_amd.api()
mvolps_amd.require_device()
A, b, c = synth.dense_lp(m, n, 12345)
P0 = api.create()
P0.load_dense(A, b, c)
P0.simplex(it_lim=40)
for tr in (16, 8):
    for chain in (1, 4, 10, 16, 32):
        api.set_tuning(tr, 1, -1)
        api.set_chain(chain)
        P = P0.copy()
        P.simplex(it_lim=2 * chain)
        api.sync()
        piv = 20 if chain == 1 else 10 * chain
        t = time.perf_counter()
        P.simplex(it_lim=piv)
        api.sync()
        dt = time.perf_counter() - t
        api.profile_reset()
        api.profile_enable(1)
        P.simplex(it_lim=piv)
        api.profile_enable(0)
        ms, k = api.profile_update_ms(), api.profile_update_launches()
        us = ms / max(1, k) * 1e3
        print(json.dumps({"m": m, "n": n, "tr": tr, "chain": chain, "bulk_us": round(us, 2), "launches": k,
                          "bulk_TBps": round(16.0 * (m + 1) * (n + 1) / us / 1e6, 3), "us_per_pivot": round(dt / piv * 1e6, 2)}), flush=True)
