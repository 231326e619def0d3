"""One-off fuzz: many seeded LPs / ILPs, GPU engine vs CPU oracle, bitwise (tests/ hold the small fixed set)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import synth, capi, bnb
from oracle import oracle
from tests import lpgen

gpu, orc = mvolps_amd.api(), oracle.api()
bad = []

def same(g, o):
    if g.status != o.status or g.it_cnt != o.it_cnt:
        return False
    if not np.array_equal(g.tableau(), o.tableau()):
        return False
    return all(np.array_equal(x, y) for x, y in zip(g.basis(), o.basis()))

t0 = time.time()
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
n_gen = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for k in range(n_gen):
    A, row_b, col_b, c, direction = lpgen.random_general_lp(rng, mmax=30, nmax=40)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_general(A, row_b, col_b, c, c0=-2.0, direction=direction)
        P.simplex()
    if not same(g, o):
        bad.append(("general", k))
    # warm start: change a random column bound and re-solve
    j = int(rng.integers(1, A.shape[1] + 1))
    t, lo = int(rng.choice([capi.UP, capi.LO, capi.FX])), float(rng.integers(-2, 4))
    for P in (g, o):
        P.api.set_col_bnds(P.h, j, t, lo, lo)
        P.simplex()
    if not same(g, o):
        bad.append(("general-warm", k))
    # more warm starts on the same handles: a row bound, the objective, the direction, a second solve as is
    i = int(rng.integers(1, A.shape[0] + 1))
    t2, v2 = int(rng.choice([capi.UP, capi.LO, capi.DB, capi.FR])), float(rng.integers(-6, 7))
    for P in (g, o):
        P.api.set_row_bnds(P.h, i, t2, v2, v2 + 2.0)
        P.simplex()
    if not same(g, o):
        bad.append(("row-warm", k))
    jc = int(rng.integers(1, A.shape[1] + 1)); vc = float(rng.integers(-5, 6))
    for P in (g, o):
        P.api.set_obj_coef(P.h, jc, vc)
        P.simplex()
    if not same(g, o):
        bad.append(("obj-warm", k))
    for P in (g, o):
        P.api.set_obj_dir(P.h, capi.MIN if direction == capi.MAX else capi.MAX)
        P.simplex()
        P.simplex()
    if not same(g, o):
        bad.append(("dir-warm", k))
print("general done", time.time() - t0, "bad", len(bad), flush=True)
for k in range(120):
    m, n = int(rng.integers(20, 400)), int(rng.integers(20, 700))
    A, b, c = synth.dense_lp(m, n, 5000 + k)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        P.simplex()
    if not same(g, o):
        bad.append(("dense", k, m, n))
print("dense done", time.time() - t0, "bad", len(bad), flush=True)
n_pert = n_bland = 0
for k in range(60):
    m, n = int(rng.integers(20, 260)), int(rng.integers(20, 320))
    A, b, c = lpgen.degenerate_lp(m, n, 9000 + k, frac0=float(rng.choice([0.5, 0.8, 0.95, 1.0])))
    g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
    lim = int(rng.integers(20, 400))
    for P in (g, o):
        P.simplex(it_lim=lim)
    if not same(g, o):
        bad.append(("degenerate-limit", k, m, n))
    for P in (g, o):
        P.simplex()
    if not same(g, o) or g.pert_cnt != o.pert_cnt or g.bland_cnt != o.bland_cnt:
        bad.append(("degenerate", k, m, n))
    n_pert += o.pert_cnt
    n_bland += o.bland_cnt
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
    for j in frac[:2]:
        for P in (g, o):
            P.kid = P.copy()
            P.api.set_col_bnds(P.kid.h, j, capi.UP, 0.0, float(np.floor(x[j - 1])))
            P.kid.simplex()
        if not same(g.kid, o.kid):
            bad.append(("degenerate-child", k, j))
        n_bland += o.kid.bland_cnt
print("degenerate done", time.time() - t0, "bad", len(bad), "perturbations", n_pert, "bland pivots", n_bland, flush=True)
for k in range(40):
    m, n = int(rng.integers(4, 20)), int(rng.integers(6, 36))
    A, b, c, U = synth.dense_ilp(m, n, 7000 + k, int(rng.integers(1, 4)))
    for kw in (dict(quirks=1, max_nodes=400), dict(quirks=0, max_nodes=1500), dict(quirks=1, cut_strat=1, max_nodes=120),
               dict(quirks=0, cut_strat=1, cut_select=1, cut_chance=0.5, node_strat=1, max_nodes=800)):
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
        got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
        ok = all(got[x] == ref[x] for x in ("events", "prune", "parent", "count", "total_pivots", "node_bound", "x", "incumbent_oid"))
        if not ok:
            bad.append(("bnb", k, m, n, kw))
print("bnb done", time.time() - t0, "bad", len(bad), flush=True)
print(json.dumps(bad[:20], default=str))
