"""B&B fuzz at medium size: ILPs of 30..80 rows x 60..160 columns, up to 3000 nodes, GPU driver + engine against the
oracle restatement + engine: events, prune labels, bounds, pivots, incumbent.  usage: fuzz_bnb_mid.py SEED [CASES]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, bnb
from oracle import oracle
from tests import lpgen
gpu, orc = mvolps_amd.api(), oracle.api()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 16
keys = ("events", "prune", "parent", "count", "total_pivots", "node_bound", "x", "incumbent_oid", "has_incumbent")
bad = []
t0 = time.time()
for k in range(cases):
    m, n, U = int(rng.integers(30, 80)), int(rng.integers(60, 160)), int(rng.integers(1, 4))
    seed = int(rng.integers(1, 10**6))
    A, b, c, UU = synth.dense_ilp(m, n, seed, U)
    for kw in (dict(quirks=0, max_nodes=3000), dict(quirks=0, cut_strat=1, max_nodes=600), dict(quirks=1, max_nodes=1500)):
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, UU), **kw)
        got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, UU), **kw)
        d = [x for x in keys if repr(got[x]) != repr(ref[x])]
        if d: bad.append((m, n, seed, U, kw, d))
    print(k, m, n, round(time.time() - t0, 1), "bad", len(bad), flush=True)
print(bad)
