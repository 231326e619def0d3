"""HIP-event cost per pivot (why bench.py times its per-kernel pass separately)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mvolps_amd
from mvolps_amd import synth
api = mvolps_amd.api()
A, b, c = synth.dense_lp(4096, 8192, 12345)
for rep in range(2):
    for prof in (0, 1, 0, 1):
        P = api.create(); P.load_dense(A, b, c); P.simplex(it_lim=50); api.sync()
        api.profile_reset(); api.profile_enable(prof)
        t = time.perf_counter(); P.simplex(it_lim=600); api.sync(); el = time.perf_counter() - t
        api.profile_enable(0)
        print("profile", prof, "us/pivot %.2f" % (el / 600 * 1e6), "k_fb avg us %.2f" % (api.profile_update_ms() / max(1, api.profile_update_launches()) * 1e3), flush=True)
        del P
