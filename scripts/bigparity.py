"""One-off: full solves of larger dense LPs, GPU against the CPU oracle, bitwise (tableau, basis, pivot count)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth
from oracle import oracle
gpu, orc = mvolps_amd.api(), oracle.api()
for (m, n, seed) in [(1024, 2048, 12345), (777, 3001, 5), (2048, 4096, 12345), (1500, 600, 9), (4096, 8192, 12345)]:
    A, b, c = synth.dense_lp(m, n, seed)
    g, o = gpu.create(), orc.create()
    t = time.time(); g.load_dense(A, b, c); g.simplex(); tg = time.time() - t
    t = time.time(); o.load_dense(A, b, c); o.simplex(); to = time.time() - t
    same = g.status == o.status and g.it_cnt == o.it_cnt and np.array_equal(g.tableau(), o.tableau()) and \
        all(np.array_equal(x, y) for x, y in zip(g.basis(), o.basis()))
    print(json.dumps({"m": m, "n": n, "seed": seed, "pivots": g.it_cnt, "oracle_pivots": o.it_cnt, "bitwise_equal": bool(same),
                      "obj": g.obj, "gpu_s": tg, "oracle_s": to}), flush=True)
