"""Fuzz at larger sizes (more quotients per solve): general-bounds LPs up to 150x200 with warm starts, dense LPs up to
700x1200, degenerate LPs up to 400x500; GPU engine vs CPU oracle, bitwise.  usage: fuzz_large.py SEED"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, capi
from oracle import oracle
from tests import lpgen
gpu, orc = mvolps_amd.api(), oracle.api()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = []
def same(g, o):
    return g.status == o.status and g.it_cnt == o.it_cnt and np.array_equal(g.tableau(), o.tableau()) and \
        all(np.array_equal(x, y) for x, y in zip(g.basis(), o.basis()))
t0 = time.time()
for k in range(300):
    A, row_b, col_b, c, d = lpgen.random_general_lp(rng, mmax=150, nmax=200)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_general(A, row_b, col_b, c, direction=d); P.simplex()
    if not same(g, o): bad.append(("general", k)); continue
    j = int(rng.integers(1, A.shape[1] + 1)); v = float(rng.integers(-2, 4))
    for P in (g, o):
        P.api.set_col_bnds(P.h, j, int(rng.integers(2, 6)) if False else capi.DB, v, v + 1.0); P.simplex()
    if not same(g, o): bad.append(("general-warm", k))
print("general done", round(time.time() - t0, 1), "bad", len(bad), flush=True)
for k in range(40):
    m, n = int(rng.integers(200, 700)), int(rng.integers(300, 1200))
    A, b, c = synth.dense_lp(m, n, 31000 + k)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c); P.simplex()
    if not same(g, o): bad.append(("dense", k, m, n))
print("dense done", round(time.time() - t0, 1), "bad", len(bad), flush=True)
for k in range(30):
    m, n = int(rng.integers(100, 400)), int(rng.integers(100, 500))
    A, b, c = lpgen.degenerate_lp(m, n, 41000 + k, frac0=float(rng.choice([0.8, 0.95])))
    g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
    for P in (g, o): P.simplex()
    if not same(g, o): bad.append(("degenerate", k, m, n))
print("degenerate done", round(time.time() - t0, 1), "bad", len(bad), flush=True)
print(bad)
