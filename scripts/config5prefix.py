"""First N nodes of the calibrated config-5 tree through the window driver (profiler input)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5.json")))
api = mvolps_amd.api()
A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=int(sys.argv[1]) if len(sys.argv) > 1 else 600, window=64)
print(r["count"], r["total_pivots"])
