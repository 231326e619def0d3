"""Which store policy streams fastest at which size: k_fb time per launch with plain, non-temporal and write-through
(sc1) access, over tableau sizes around the 256 MiB Infinity Cache.  usage: ntsweep.py [pivots]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
api = mvolps_amd.api()
mvolps_amd.require_device()
for (m, n) in ((2048, 4096), (3072, 6144), (4096, 6144), (4096, 8192), (4608, 8192), (5120, 8192), (6144, 8192), (4096, 12288), (8192, 8192)):
    A, b, c = synth.dense_lp(m, n, 12345)
    row = {"m": m, "n": n, "MB": (m + 1) * ((n + 32) // 32 * 32) * 8 / 2**20}
    for nt in (0, 1, 2):  # plain, non-temporal, write-through stores
        api.set_tuning(16, 1, nt)
        P = api.create()
        P.load_dense(A, b, c)
        P.simplex(it_lim=30)
        api.profile_reset()
        api.profile_enable(1)
        P.simplex(it_lim=steps)
        api.profile_enable(0)
        us = api.profile_update_ms() / max(1, api.profile_update_launches()) * 1e3
        row["nt%d_us" % nt] = us
        row["nt%d_GBps" % nt] = 16 * (m + 1) * (n + 1) / (us * 1e-6) / 1e9
        del P
    print(json.dumps(row), flush=True)
    del A
