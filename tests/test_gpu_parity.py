"""GPU parity: the HIP engine against the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): basis indices and statuses bit-exact; tableau entries and
objective are compared BITWISE as well (np.array_equal), which is stronger than the 1e-9
relative tolerance asked for -- both sides run the same fma/div sequence.
"""
import ctypes as C

import numpy as np
import pytest

from mvolps_amd import capi, synth
from mvolps_amd.capi import DB, LO, MAX, UP

from . import lpgen

pytestmark = pytest.mark.gpu


def assert_same_state(g, o, what=""):
    assert g.status == o.status, (what, g.status, o.status)
    assert g.it_cnt == o.it_cnt, (what, g.it_cnt, o.it_cnt)
    hg, ho = g.basis(), o.basis()
    for x, y, nm in zip(hg, ho, ("head", "nonbasic", "flag")):
        assert np.array_equal(x, y), (what, nm)
    tg, to = g.tableau(), o.tableau()
    assert np.array_equal(tg, to), (what, "tableau max abs diff %g" % np.nanmax(np.abs(tg - to)))
    assert g.obj == o.obj
    assert np.array_equal(g.col_prim(), o.col_prim())
    assert np.array_equal(g.row_prim(), o.row_prim())
    assert np.array_equal(g.col_stat(), o.col_stat())
    assert np.array_equal(g.row_stat(), o.row_stat())


@pytest.mark.parametrize("m,n,seed", [(3, 5, 1), (17, 33, 2), (64, 128, 12345), (128, 256, 12345), (100, 37, 5), (256, 512, 12345),
                                      (512, 1024, 12345), (333, 1500, 8), (1024, 2048, 12345), (1500, 600, 9)])
def test_dense_lp_bit_exact(gpu, orc, m, n, seed):
    A, b, c = synth.dense_lp(m, n, seed)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        assert P.simplex() == 0
    assert g.status == capi.OPT
    assert_same_state(g, o, "dense %dx%d" % (m, n))


def test_iteration_limit_then_resume(gpu, orc):
    A, b, c = synth.dense_lp(128, 256, 7)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        assert P.simplex(it_lim=10) == capi.EITLIM
    assert g.it_cnt == 10
    assert_same_state(g, o, "after 10 pivots")
    for P in (g, o):
        assert P.simplex() == 0
    assert_same_state(g, o, "resumed")


def test_general_bounds_bit_exact(gpu, orc):
    """All bound types, both directions: exercises phase 1, the dual simplex and bound flips."""
    rng = np.random.default_rng(7)
    seen = set()
    for trial in range(120):
        A, row_b, col_b, c, direction = lpgen.random_general_lp(rng)
        g, o = gpu.create(), orc.create()
        for P in (g, o):
            P.load_general(A, row_b, col_b, c, c0=1.5, direction=direction)
            P.rc = P.simplex()
        assert g.rc == o.rc, trial
        assert_same_state(g, o, "general trial %d" % trial)
        seen.add(g.status)
    assert capi.OPT in seen and len(seen) >= 2


def test_branch_children_warm_start(gpu, orc):
    """bs.cpp:269-288: clone the solved node twice, change one column bound, re-solve."""
    A, b, c, U = synth.dense_ilp(24, 48, seed=6, U=2)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    assert_same_state(g, o, "root")
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
    assert frac
    pick = frac[0]
    for P in (g, o):
        P.s2, P.s3 = P.copy(), P.copy()
        P.api.set_col_bnds(P.s2.h, pick, UP, 0.0, float(np.floor(x[pick - 1])))
        P.api.set_col_bnds(P.s3.h, pick, LO, float(np.ceil(x[pick - 1])), 0.0)
        P.s2.simplex()
        P.s3.simplex()
    assert_same_state(g.s2, o.s2, "S2")
    assert_same_state(g.s3, o.s3, "S3")
    # the parent is untouched by its children
    assert_same_state(g, o, "parent after children")
    # re-solve of an optimal clone costs zero pivots (bs.cpp:116-117)
    for P in (g, o):
        P.again = P.s2.copy(names=capi.OFF)
        P.again.simplex()
    assert g.again.it_cnt == g.s2.it_cnt
    assert_same_state(g.again, o.again, "re-solve")


def test_cut_row_append(gpu, orc):
    """cut.cpp:23-43: add one >= row to a solved problem, then warm-start children from it."""
    A, b, c, U = synth.dense_ilp(16, 32, seed=5, U=2)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    x = o.col_prim()
    n = len(x)
    rng = np.random.default_rng(3)
    v = np.round(rng.normal(size=n) * 2)
    ind = np.arange(n + 1, dtype=np.int32)
    val = np.concatenate([[0.0], v])
    lb = float(v @ x) + 0.75  # violated by the current vertex
    for P in (g, o):
        r = P.api.add_rows(P.h, 1)
        assert r == 17
        P.set_mat_row(r, ind, val)
        P.api.set_row_bnds(P.h, r, LO, lb, 0.0)
    assert np.array_equal(g.tableau(), o.tableau())
    assert np.array_equal(g.row_prim(), o.row_prim())
    for P in (g, o):
        P.ch = P.copy()
        P.ch.simplex()
    assert_same_state(g.ch, o.ch, "after cut")


def test_objective_change_recomputes_cost_row(gpu, orc):
    A, b, c = synth.dense_lp(40, 70, 11)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        P.simplex()
        P.api.set_obj_coef(P.h, 3, 2.5)
        P.api.set_obj_coef(P.h, 0, -1.0)
    assert np.array_equal(g.tableau(), o.tableau())
    for P in (g, o):
        P.simplex()
    assert_same_state(g, o, "after objective change")


def test_eval_tab_row_and_queries(gpu, orc):
    A, b, c, U = synth.dense_ilp(12, 20, seed=4, U=3)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        P.simplex()
    m = g.m
    stat = o.col_stat()
    for j in range(1, g.n + 1):
        if stat[j - 1] == capi.BS:
            ig, vg = g.eval_tab_row(m + j)
            io, vo = o.eval_tab_row(m + j)
            assert np.array_equal(ig, io) and np.array_equal(vg, vo)
    for i in range(1, m + 1):
        assert g.api.get_row_ub(g.h, i) == o.api.get_row_ub(o.h, i)
        ig, vg = g.get_mat_row(i)
        io, vo = o.get_mat_row(i)
        assert np.array_equal(ig, io) and np.array_equal(vg, vo)
    for j in range(1, g.n + 1):
        assert g.api.get_col_kind(g.h, j) == o.api.get_col_kind(o.h, j)
        assert g.api.get_col_ub(g.h, j) == o.api.get_col_ub(o.h, j)
        assert g.api.get_col_dual(g.h, j) == o.api.get_col_dual(o.h, j)


def test_infeasible_and_unbounded(gpu, orc):
    # infeasible: x1 + x2 <= 1, x1 + x2 >= 3
    A = np.array([[1.0, 1.0], [1.0, 1.0]])
    rows = [(UP, 0.0, 1.0), (LO, 3.0, 0.0)]
    cols = [(LO, 0.0, 0.0)] * 2
    for api in (gpu, orc):
        P = api.create()
        P.load_general(A, rows, cols, np.array([1.0, 1.0]), direction=MAX)
        P.simplex()
        assert P.status == capi.NOFEAS
    # unbounded: max x1, x1 - x2 <= 1
    A = np.array([[1.0, -1.0]])
    for api in (gpu, orc):
        P = api.create()
        P.load_general(A, [(UP, 0.0, 1.0)], cols, np.array([1.0, 0.0]), direction=MAX)
        P.simplex()
        assert P.status == capi.UNBND


def test_batch_solve_equals_sequential(gpu, orc):
    """mvx_simplex_batch: a window of independent node LPs on separate streams, same bits as one by one."""
    import ctypes as C

    A, b, c, U = synth.dense_ilp(24, 48, seed=6, U=2)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        P.simplex()
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
    kids_g, kids_o = [], []
    for j in frac[:6]:
        for (t, lo, hi) in ((UP, 0.0, float(np.floor(x[j - 1]))), (LO, float(np.ceil(x[j - 1])), 0.0)):
            for P, kids in ((g, kids_g), (o, kids_o)):
                ch = P.copy()
                P.api.set_col_bnds(ch.h, j, t, lo, hi)
                kids.append(ch)
    # a fresh (never solved) dense LP rides along: exercises tableau build + the primal fast path in a batch
    A2, b2, c2 = synth.dense_lp(40, 90, 9)
    for api, kids in ((gpu, kids_g), (orc, kids_o)):
        Q = api.create()
        Q.load_dense(A2, b2, c2)
        kids.append(Q)
    arr = (C.c_void_p * len(kids_g))(*[k.h for k in kids_g])
    rcs = (C.c_int * len(kids_g))()
    assert gpu.simplex_batch(arr, len(kids_g), None, rcs) == 0
    for k in kids_o:
        k.simplex()
    assert len(kids_g) >= 9
    for kg, ko, rc in zip(kids_g, kids_o, rcs):
        assert rc == 0
        assert_same_state(kg, ko, "batched child")


def test_batch_mixed_shapes_phase1_and_limits(gpu, orc):
    """One batched launch carrying handles of different shapes, a phase-1 start (falls back to the
    single-handle path), an infeasible LP, an iteration limit and more handles than slots."""
    import ctypes as C

    rng = np.random.default_rng(21)
    specs = []
    for k in range(40):
        specs.append(("gen", lpgen.random_general_lp(rng)))
    for (m, n, seed) in [(5, 9, 1), (33, 70, 2), (64, 40, 3), (120, 200, 4)]:
        specs.append(("dense", synth.dense_lp(m, n, seed)))
    pairs = []
    for kind, sp in specs:
        hs = []
        for api in (gpu, orc):
            P = api.create()
            if kind == "gen":
                A, row_b, col_b, c, direction = sp
                P.load_general(A, row_b, col_b, c, c0=0.5, direction=direction)
            else:
                P.load_dense(*sp)
            hs.append(P)
        pairs.append(hs)
    arr = (C.c_void_p * len(pairs))(*[g.h for g, _ in pairs])
    rcs = (C.c_int * len(pairs))()
    parm = capi.Smcp()
    gpu.init_smcp(C.byref(parm))
    parm.it_lim = 25
    assert gpu.simplex_batch(arr, len(pairs), C.byref(parm), rcs) == 0
    seen = set()
    for (g, o), rc in zip(pairs, rcs):
        assert o.simplex(it_lim=25) == rc
        assert_same_state(g, o, "mixed batch")
        seen.add(g.status)
    assert {capi.OPT, capi.UNBND} <= seen and capi.FEAS in seen  # FEAS: the 120x200 LP hit the limit


@pytest.mark.parametrize("tr,hot,nt", [(4, 1, 0), (8, 1, 0), (16, 1, 0), (32, 1, 0), (16, 0, 0), (8, 1, 1)])
def test_every_update_variant_is_bit_exact(gpu, orc, tr, hot, nt):
    """All instantiations of the streamed update (row-block depth, batched loads, non-temporal access)
    on a ragged shape, against the oracle, pivot for pivot; then the default selection again."""
    A, b, c = synth.dense_lp(301, 1031, 77)
    o = orc.create()
    o.load_dense(A, b, c)
    try:
        gpu.set_tuning(tr, hot, nt)
        g = gpu.create()
        g.load_dense(A, b, c)
        for lim in (1, 2, 37):
            assert g.simplex(it_lim=lim) == o.simplex(it_lim=lim)
            assert np.array_equal(g.tableau(), o.tableau()), (tr, hot, nt, lim)
        g.simplex()
        o.simplex()
        assert_same_state(g, o, "variant %s" % ((tr, hot, nt),))
    finally:
        gpu.set_tuning(0, 1, 0)


def test_large_grid_first_pivots_bit_exact(gpu, orc):
    """The 16-row-deep tiles the 4096x8192 headline runs with (>= 2048 workgroups), checked bitwise
    against the oracle on the first pivots of a 2048x8192 LP (the full solve is checked against the
    HiGHS golden in test_gpu_golden.py)."""
    A, b, c = synth.dense_lp(2048, 8192, 4242)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
    for lim in (3, 21):
        assert g.simplex(it_lim=lim) == o.simplex(it_lim=lim) == capi.EITLIM
        for x, y in zip(g.basis(), o.basis()):
            assert np.array_equal(x, y)
        assert np.array_equal(g.tableau(), o.tableau())


def test_fused_primal_path_bit_exact(gpu, orc):
    """Primal phase 2 on the k_fa / k_fb pair against the oracle: iteration limits that end a batch at
    every position of the queue, and bound flips of boxed columns inside the fused path."""
    A, b, c = synth.dense_lp(150, 420, 31)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
    for lim in (1, 2, 3, 10, 11):
        assert g.simplex(it_lim=lim) == o.simplex(it_lim=lim)
        assert np.array_equal(g.tableau(), o.tableau()), lim
    g.simplex()
    o.simplex()
    assert_same_state(g, o, "fused")
    A2, b2, c2, U = synth.dense_ilp(40, 90, 17, 2)
    g2 = lpgen.load_ilp(gpu, A2, b2, c2, U)
    o2 = lpgen.load_ilp(orc, A2, b2, c2, U)
    for P in (g2, o2):
        P.simplex()
    assert_same_state(g2, o2, "fused boxed")


def test_tall_tableau_dual_path_bit_exact(gpu, orc):
    """A tableau taller than one k_select pass (m > 1024 lanes): warm-started children (bs.cpp:274-288)
    one by one and through mvx_simplex_batch, then a grandchild, against the oracle -- the dual simplex
    with multi-pass row scans and 8-row update tiles."""
    import ctypes as C

    A, b, c, U = synth.dense_ilp(1100, 300, seed=21, U=3)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    assert_same_state(g, o, "tall root")
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
    assert len(frac) >= 2
    kids_g, kids_o = [], []
    for j in frac[:3]:
        for (t, lo, hi) in ((UP, 0.0, float(np.floor(x[j - 1]))), (LO, float(np.ceil(x[j - 1])), 0.0)):
            for P, kids in ((g, kids_g), (o, kids_o)):
                ch = P.copy()
                P.api.set_col_bnds(ch.h, j, t, lo, hi)
                kids.append(ch)
    # first two children one by one, the rest as a batch
    for kg, ko in zip(kids_g[:2], kids_o[:2]):
        assert kg.simplex() == ko.simplex()
        assert kg.it_cnt > g.it_cnt  # the dual simplex really pivoted
        assert_same_state(kg, ko, "tall child")
    rest_g, rest_o = kids_g[2:], kids_o[2:]
    arr = (C.c_void_p * len(rest_g))(*[k.h for k in rest_g])
    rcs = (C.c_int * len(rest_g))()
    assert gpu.simplex_batch(arr, len(rest_g), None, rcs) == 0
    for kg, ko in zip(rest_g, rest_o):
        ko.simplex()
        assert_same_state(kg, ko, "tall batched child")
    # grandchildren: a second bound change on an already re-solved child
    xg = kids_o[0].col_prim()
    frac2 = [j + 1 for j in range(len(xg)) if np.trunc(xg[j]) != xg[j]]
    if frac2:
        j = frac2[-1]
        for P in (kids_g[0], kids_o[0]):
            P.api.set_col_bnds(P.h, j, LO, float(np.ceil(xg[j - 1])), 0.0)
            P.simplex()
        assert_same_state(kids_g[0], kids_o[0], "tall grandchild")


def test_stalling_lps_bit_exact(gpu, orc):
    """The anti-stalling rules on the device against the oracle: textbook cycling LPs, a massively degenerate
    LP that needs the bound perturbation (applied by k_select, removed when phase 2 ends, cleaned up by the
    dual simplex), the same LP cut short by an iteration limit while perturbed, and Bland's rule forced on
    from the first degenerate pivot of a warm-started dual solve."""
    for name, (A, b, c) in lpgen.CYCLING.items():
        A, b, c = (np.array(v, float) for v in (A, b, c))
        g, o = gpu.create(), orc.create()
        for P in (g, o):
            P.load_dense(A, b, c)
            assert P.simplex() == 0
        assert g.pert_cnt == o.pert_cnt and g.bland_cnt == o.bland_cnt
        assert_same_state(g, o, name)
    A, b, c = lpgen.degenerate_lp(150, 150, 7)
    g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
    for P in (g, o):
        assert P.simplex() == 0
    assert g.pert_cnt == o.pert_cnt == 1
    assert g.bland_cnt == o.bland_cnt
    assert_same_state(g, o, "degenerate 150x150")
    # an iteration limit that falls inside the perturbed stretch: bounds are restored before returning
    g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
    for lim in (120, 30, 400):
        assert g.simplex(it_lim=lim) == o.simplex(it_lim=lim)
        assert g.pert_cnt == o.pert_cnt and g.bland_cnt == o.bland_cnt
        assert_same_state(g, o, "degenerate, limit %d" % lim)
    for P in (g, o):
        P.simplex()
    assert_same_state(g, o, "degenerate resumed")
    # warm-started dual simplex on a degenerate vertex (children of the solved LP): Bland fallback territory
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
    for j in frac[:4]:
        for P in (g, o):
            P.kid = P.copy()
            P.api.set_col_bnds(P.kid.h, j, UP, 0.0, float(np.floor(x[j - 1])))
            P.kid.simplex()
        assert g.kid.bland_cnt == o.kid.bland_cnt and g.kid.pert_cnt == o.kid.pert_cnt
        assert_same_state(g.kid, o.kid, "degenerate child %d" % j)


@pytest.mark.parametrize("limit", [1, 2])
def test_bland_fallback_bit_exact(gpu, orc, limit):
    """With the stall limit forced down to 1-2 degenerate pivots the perturbation is spent early and Bland's
    rule takes over in every phase (primal 2, primal 1, dual): same choices on both sides."""
    try:
        gpu.set_stall_limit(limit)
        orc.set_stall_limit(limit)
        rng = np.random.default_rng(7)
        n_bland = n_pert = 0
        for trial in range(120):
            A, row_b, col_b, c, direction = lpgen.random_general_lp(rng)
            g, o = gpu.create(), orc.create()
            for P in (g, o):
                P.load_general(A, row_b, col_b, c, c0=1.5, direction=direction)
                P.rc = P.simplex()
            assert g.rc == o.rc and g.bland_cnt == o.bland_cnt and g.pert_cnt == o.pert_cnt, trial
            assert_same_state(g, o, "limit %d trial %d" % (limit, trial))
            n_bland += o.bland_cnt
            n_pert += o.pert_cnt
        A, b, c = lpgen.degenerate_lp(60, 80, 2)
        g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
        for P in (g, o):
            assert P.simplex() == 0
        assert g.bland_cnt == o.bland_cnt and g.pert_cnt == o.pert_cnt == 1
        assert_same_state(g, o, "degenerate, limit %d" % limit)
        n_bland += o.bland_cnt
        assert n_pert >= 1
        if limit == 1:
            assert n_bland >= 5  # the fallback really ran
    finally:
        gpu.set_stall_limit(0)
        orc.set_stall_limit(0)



def test_resolve_without_edits_bit_exact(gpu, orc):
    """bs.cpp:116-117 solves every node again when it is popped.  A second call on an unedited handle is a no-op
    only after OPT; after NOFEAS / UNBND the fresh devex weights may pick another row / column and pivot on.
    Whatever it does, the device does the same as the oracle."""
    rng = np.random.default_rng(11)
    seen = {}
    for trial in range(150):
        A, row_b, col_b, c, direction = lpgen.random_general_lp(rng, mmax=14, nmax=16)
        if trial % 3 == 0:  # make it infeasible: two contradictory copies of row 0
            A[1] = A[0]
            if not A[0].any():
                A[0, 0] = A[1, 0] = 1.0
            row_b[0] = (UP, 0.0, 1.0)
            row_b[1] = (LO, 3.0, 0.0)
            col_b = [(DB, -5.0, 5.0)] * A.shape[1]
        g, o = gpu.create(), orc.create()
        for P in (g, o):
            P.load_general(A, row_b, col_b, c, direction=direction)
            P.simplex()
        first = (o.status, o.it_cnt)
        for P in (g, o):
            P.rc2 = P.simplex()
        assert g.rc2 == o.rc2
        assert_same_state(g, o, "second solve, trial %d" % trial)
        seen.setdefault(first[0], []).append(o.it_cnt - first[1])
    assert capi.OPT in seen and capi.NOFEAS in seen and capi.UNBND in seen
    assert all(d == 0 for d in seen[capi.OPT])


@pytest.mark.parametrize("slots", [2, 3, 5])
def test_batch_with_few_slots_refills_and_compacts(gpu, orc, slots):
    """More handles than slots: finished slots are refilled from the pending list at every sync, and once
    that list is empty the occupied slots are compacted so that launches carry no idle slot.  Same bits as
    one solve per handle, whatever the slot count."""
    import ctypes as C

    try:
        gpu.set_batch_slots(slots)
        A, b, c, U = synth.dense_ilp(30, 60, seed=9, U=3)
        g = lpgen.load_ilp(gpu, A, b, c, U)
        o = lpgen.load_ilp(orc, A, b, c, U)
        for P in (g, o):
            P.simplex()
        x = o.col_prim()
        frac = [j + 1 for j in range(len(x)) if np.trunc(x[j]) != x[j]]
        kids_g, kids_o = [], []
        for j in frac[:7]:
            for (t, lo, hi) in ((UP, 0.0, float(np.floor(x[j - 1]))), (LO, float(np.ceil(x[j - 1])), 0.0)):
                for P, kids in ((g, kids_g), (o, kids_o)):
                    ch = P.copy()
                    P.api.set_col_bnds(ch.h, j, t, lo, hi)
                    kids.append(ch)
        assert len(kids_g) >= 8
        arr = (C.c_void_p * len(kids_g))(*[k.h for k in kids_g])
        rcs = (C.c_int * len(kids_g))()
        assert gpu.simplex_batch(arr, len(kids_g), None, rcs) == 0
        for kg, ko, rc in zip(kids_g, kids_o, rcs):
            assert rc == ko.simplex()
            assert_same_state(kg, ko, "batched child, %d slots" % slots)
    finally:
        gpu.set_batch_slots(64)


def test_glpk_default_tolerances_behind_null_parameters(gpu, orc):
    """mvx_set_default_tolerances(1e-7, 1e-7, 1e-9): GLPK's own defaults [GLPK-recalled] behind every NULL-parameter
    solve -- all MVOLPS ever makes (bs.cpp:117,279,287).  Oracle and device take the same switch and stay bit-identical,
    LP and branch-and-bound alike."""
    from mvolps_amd import bnb
    from oracle import oracle

    try:
        for a in (gpu, orc):
            a.set_default_tolerances(1e-7, 1e-7, 1e-9)
        parm = capi.Smcp()
        gpu.init_smcp(C.byref(parm))
        assert (parm.tol_bnd, parm.tol_dj, parm.tol_piv) == (1e-7, 1e-7, 1e-9)
        A, b, c = synth.dense_lp(96, 160, 3)
        g, o = gpu.create(), orc.create()
        for P in (g, o):
            P.load_dense(A, b, c)
            assert P.simplex() == 0
        assert g.it_cnt == o.it_cnt and np.array_equal(g.tableau(), o.tableau())
        A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
        for quirks in (1, 0):
            ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=300)
            got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=quirks, max_nodes=300)
            for k in ("events", "prune", "parent", "node_bound", "total_pivots", "x"):
                assert got[k] == ref[k], (quirks, k)
    finally:
        for a in (gpu, orc):
            a.set_default_tolerances(1e-9, 1e-9, 1e-9)


@pytest.mark.parametrize("case", [(358, 124, 41004, 0.8), (288, 252, 41002, 0.8), (314, 433, 41000, 0.95), (390, 253, 41024, 0.8)],
                         ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_degenerate_lps_through_the_resident_tableau_kernel(gpu, orc, case):
    """Massively degenerate, boxed LPs at sizes the resident-tableau kernel takes (k_persist): ties in every ratio
    test, bound flips, steps of length zero.  (scripts/fuzz_large.py found the first version of that kernel choosing
    the right leaving row with the wrong bound on exactly these -- a candidate struct left uninitialised on one
    path; every reduction candidate is value-initialised now.)"""
    m, n, seed, frac0 = case
    A, b, c = lpgen.degenerate_lp(m, n, seed, frac0=frac0)
    gpu.set_cluster(0)  # the cluster chain would take these sizes first; this test is about k_persist
    try:
        g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
        for P in (g, o):
            P.simplex()
    finally:
        gpu.set_cluster(1)
    assert g.status == o.status and g.it_cnt == o.it_cnt
    assert np.array_equal(g.tableau(), o.tableau())
    for u, v in zip(g.basis(), o.basis()):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("persist,cluster", [(0, 1), (1, 0), (0, 0)], ids=["cluster-chain", "resident-tableau", "two-launch"])
def test_short_calls_taken_over_at_their_first_pivot(gpu, orc, persist, cluster):
    """A primal call starts in the fused pipeline (k_fboot does select_step's opening feasibility check and restarts the
    devex weights), the generic step closes each batch and k_fa reports the pivot limit itself: a run of short calls --
    limits 1, 2, 3, 5, 8, 13, ... -- must leave the same bits as the oracle after every call, with the two-kernel path
    (resident-tableau kernel off) and with the default choice.  The bounded columns bring flips among the first steps;
    the last call ends on the optimum instead of the limit.  Three ways of running primal phase 2: the cluster chain
    (k_chain, the default), the resident-tableau kernel (k_persist: what serves small LPs when the cluster is off) and two
    launches per chained step (k_pc / k_pr)."""
    gpu.set_persist(persist)
    gpu.set_cluster(cluster)
    try:
        for (m, n, seed) in ((96, 400, 3), (300, 700, 11)):
            A, b, c = synth.dense_lp(m, n, seed)
            col_b = [(capi.DB, 0.0, 0.5 + (j % 3)) if j % 4 == 0 else (capi.LO, 0.0, 0.0) for j in range(n)]
            row_b = [(capi.UP, 0.0, float(v)) for v in b]
            g, o = gpu.create(), orc.create()
            for P in (g, o):
                P.load_general(A, row_b, col_b, c, direction=capi.MAX)
            lim, prev = 1, 1
            for call in range(40):
                rcs = [P.simplex(it_lim=lim) for P in (g, o)]
                assert rcs[0] == rcs[1], (m, n, call)
                assert_same_state(g, o, "%dx%d call %d (limit %d)" % (m, n, call, lim))
                if rcs[0] == 0:
                    break
                lim, prev = lim + prev, lim
            assert g.status == capi.OPT
    finally:
        gpu.set_persist(1)
        gpu.set_cluster(1)


def test_first_call_not_primal_feasible_leaves_the_fused_path_alone(gpu, orc):
    """k_fboot declines a call whose starting basis has a basic variable outside its bounds (select_step then chooses
    between the dual simplex and phase 1) and one that has bound edits waiting in the control block."""
    A, b, c, U = synth.dense_ilp(200, 420, seed=9, U=3)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    x = g.col_prim()
    frac = [j + 1 for j in range(len(x)) if abs(x[j] - round(x[j])) > 1e-6]
    assert frac
    for P, api in ((g, gpu), (o, orc)):
        api.set_col_bnds(P.h, frac[0], capi.DB, 0.0, float(np.floor(x[frac[0] - 1])))  # basic variable now above its bound
        P.rc = P.simplex(it_lim=4)  # a primal-hinted call (no clone, no dual hint) with an edit waiting
    assert g.rc == o.rc
    assert_same_state(g, o, "after the edited call")
    for P in (g, o):
        P.rc = P.simplex()
    assert g.rc == o.rc
    assert_same_state(g, o, "finished")
