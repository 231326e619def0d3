"""CPU-only fuzz: the C++ B&B driver (serial and window mode) over the oracle LP engine against the oracle restatement of bs.cpp, six strategy / cut modes per ILP.  usage: cpufuzz_bnb.py SEED CASES"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvolps_amd import synth, bnb
from oracle import oracle
from tests import lpgen
orc = oracle.api(); tab = bnb.table_from(orc)
keys = ("events", "prune", "parent", "count", "total_pivots", "node_bound", "x", "incumbent_oid", "has_incumbent")
bad = []
t = time.time()
rng = np.random.default_rng(int(sys.argv[1]))
for k in range(int(sys.argv[2])):
    m, n = int(rng.integers(3, 14)), int(rng.integers(4, 24)); U = int(rng.integers(1, 4)); seed = int(rng.integers(1, 10**6))
    A, b, c, UU = synth.dense_ilp(m, n, seed, U)
    for kw in (dict(quirks=1, max_nodes=300), dict(quirks=0, max_nodes=800), dict(quirks=1, cut_strat=1, max_nodes=100),
               dict(quirks=0, cut_strat=1, cut_select=1, cut_chance=0.5, node_strat=1, max_nodes=400), dict(quirks=0, node_strat=1, var_strat=2, max_nodes=400),
               dict(quirks=1, cut_strat=1, var_strat=1, node_strat=1, max_nodes=100), dict(quirks=0, cut_strat=1, max_nodes=300)):
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, UU), **kw)
        for window in ((1, 16, 64) if kw.get("node_strat", 0) == 0 else (1,)):
            got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, UU), table=tab, window=window, **kw)
            if any(repr(got[x]) != repr(ref[x]) for x in keys):
                bad.append((m, n, seed, U, kw, window, [x for x in keys if repr(got[x]) != repr(ref[x])]))
print("cases", int(sys.argv[2]), "bad", len(bad), round(time.time() - t, 1))
for x in bad[:10]: print(x)
