"""Window size of the FIFO window driver (nodes solved and replayed per round): wide tree, config-5 tree, one cut mode."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth, treedigest
api = mvolps_amd.api()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fx = json.load(open(os.path.join(root, "tests", "golden", "config5.json")))
wide = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
c5 = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
bnb.branch_and_bound(synth.load_ilp(api, *wide), quirks=0, max_nodes=2000, window=64)
for W in (64, 96, 128, 192, 256):
    out = {"window": W}
    for name, inst, kw, reps in (("wide", wide, dict(quirks=0, max_nodes=2000), 5), ("wide4000", wide, dict(quirks=0, max_nodes=4000), 3),
                                 ("config5", c5, dict(quirks=0), 1), ("cuts_bug", wide, dict(quirks=1, cut_strat=1, max_nodes=600), 3)):
        ts, dg = [], None
        for _ in range(reps):
            t = time.perf_counter()
            r = bnb.branch_and_bound(synth.load_ilp(api, *inst), window=W, **kw)
            ts.append(time.perf_counter() - t)
            dg = treedigest.digest(r)[:10]
        ts.sort()
        out[name] = {"ms": round(ts[len(ts) // 2] * 1e3, 1), "nodes_per_s": round(r["count"] / ts[len(ts) // 2]), "digest": dg}
    print(json.dumps(out), flush=True)
