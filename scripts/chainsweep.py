"""Pivots per bulk launch (MVX_CHAIN) against the tableau size: us per pivot of a 400-pivot primal run.
One process per setting (the variable is read once): chainsweep.py M N"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
m, n = int(sys.argv[1]), int(sys.argv[2])
api = mvolps_amd.api()
mvolps_amd.require_device()
A, b, c = synth.dense_lp(m, n, 12345)
P = api.create()
P.load_dense(A, b, c)
P.simplex(it_lim=40)
api.sync()
t = time.perf_counter()
P.simplex(it_lim=400)
api.sync()
dt = time.perf_counter() - t
print(json.dumps({"m": m, "n": n, "chain": int(os.environ.get("MVX_CHAIN", "0")), "persist": os.environ.get("MVX_PERSIST"), "us_per_pivot": round(dt / 400 * 1e6, 2), "pivots_per_s": round(400 / dt)}), flush=True)
