"""Full solves of small dense LPs: us per pivot with the resident-tableau kernel (k_persist) and with the cluster chain
(k_chain); usage: smalltime.py  (env MVX_PERSIST=0/1)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
api = mvolps_amd.api()
for (m, n) in ((128, 256), (192, 384), (256, 512), (384, 768), (512, 1024), (640, 1280)):
    A, b, c = synth.dense_lp(m, n, 12345)
    best = None
    for rep in range(3):
        P = api.create()
        P.load_dense(A, b, c)
        P.simplex(it_lim=0)
        api.sync()
        t = time.perf_counter()
        P.simplex()
        api.sync()
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    print(json.dumps({"m": m, "n": n, "persist": os.environ.get("MVX_PERSIST"), "pivots": P.it_cnt, "us_per_pivot": round(best / P.it_cnt * 1e6, 2)}), flush=True)
