"""The calibrated config-5 tree (tests/golden/config5.json) through the window driver: wall time and the driver's phase
times (MVX_BNB_TIMING=1).  usage: config5time.py [WINDOW]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth, treedigest
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5.json")))
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
window = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=200, window=window)
t = time.perf_counter()
r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, window=window)
dt = time.perf_counter() - t
print(json.dumps({"window": window, "nodes": r["count"], "pivots": r["total_pivots"], "secs": dt, "nodes_per_s": r["count"] / dt,
                  "us_per_pivot": dt / r["total_pivots"] * 1e6, "same_tree": treedigest.digest(r) == fx["full"]["sha256"]}), flush=True)
