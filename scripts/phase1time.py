"""General-bounds LPs at scale (phase 1 + phase 2 + dual): pivots, wall time; run under rocprofv3 for the kernel split.
usage: phase1time.py M N [SEED]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from tests import lpgen
api = mvolps_amd.api()
m, n = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
# same construction as lpgen.random_general_lp at a fixed size
A = np.round(rng.normal(size=(m, n)) * 3); A[rng.random((m, n)) < 0.3] = 0
x0 = rng.integers(0, 4, size=n).astype(float); act = A @ x0
from mvolps_amd.capi import DB, FR, FX, LO, MAX, MIN, UP
row_b, col_b = [], []
for i in range(m):
    t = int(rng.choice([LO, UP, DB, FX, FR], p=[0.25, 0.35, 0.2, 0.1, 0.1])); l = act[i] - rng.integers(0, 3); u = act[i] + rng.integers(0, 3)
    if t == FX: l = u = act[i]
    if t == DB and l == u: u = l + 1
    row_b.append((t, float(l), float(u)))
for j in range(n):
    t = int(rng.choice([LO, UP, DB, FX, FR], p=[0.4, 0.1, 0.35, 0.05, 0.1])); l = x0[j] - rng.integers(0, 3); u = x0[j] + rng.integers(0, 4)
    if t == FX: l = u = x0[j]
    if t == DB and l == u: u = l + 1
    col_b.append((t, float(l), float(u)))
c = np.round(rng.normal(size=n) * 5)
P = api.create(); P.load_general(A, row_b, col_b, c, direction=MAX)
t = time.perf_counter(); rc = P.simplex(); api.sync(); el = time.perf_counter() - t
print(json.dumps({"m": m, "n": n, "rc": rc, "status": P.status, "obj": P.obj, "pivots": P.it_cnt, "secs": el, "us_per_pivot": el / max(1, P.it_cnt) * 1e6,
                  "pert": P.pert_cnt, "bland": P.bland_cnt}))
