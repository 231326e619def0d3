"""Wide 512x1024 tree through the window driver: nodes/s against the window size (run once per MVX_BNB_DEPTH setting;
the environment variable is read once per process).  usage: MVX_BNB_DEPTH=d bnbdepth.py [NODES]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=200)
for window in (32, 64, 128, 256):
    t = time.perf_counter()
    r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=nodes, window=window)
    dt = time.perf_counter() - t
    print(json.dumps({"depth": os.environ.get("MVX_BNB_DEPTH", "2"), "slots": os.environ.get("MVX_BATCH_SLOTS"), "window": window, "nodes": r["count"], "pivots": r["total_pivots"], "nodes_per_s": round(r["count"] / dt)}), flush=True)
