"""GPU: edge cases of the engine's bookkeeping paths (row-capacity growth, zero-pivot limits, tiny and
ragged shapes, repeated clone/free cycles through the slab cache)."""
import numpy as np
import pytest

from mvolps_amd import capi, synth
from mvolps_amd.capi import LO, UP

from . import lpgen

pytestmark = pytest.mark.gpu


def test_many_appended_rows_grow_the_slab(gpu, orc):
    """More cut rows than the spare capacity: the slab is re-allocated and copied (grow_rows)."""
    A, b, c, U = synth.dense_ilp(12, 24, 7, 3)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    rng = np.random.default_rng(11)
    n = 24
    ind = np.arange(n + 1, dtype=np.int32)
    for k in range(80):
        v = np.round(rng.normal(size=n) * 2)
        x = o.col_prim()
        lb = float(v @ x) - (0.0 if k % 3 else -0.25)
        for P in (g, o):
            r = P.api.add_rows(P.h, 1)
            P.set_mat_row(r, ind, np.concatenate([[0.0], v]))
            P.api.set_row_bnds(P.h, r, LO, lb, 0.0)
            P.simplex()
        assert g.status == o.status and g.it_cnt == o.it_cnt, k
        assert np.array_equal(g.tableau(), o.tableau()), k
    assert g.m == 12 + 80


def test_zero_iteration_limit(gpu, orc):
    A, b, c = synth.dense_lp(30, 50, 4)
    for api in (gpu, orc):
        P = api.create()
        P.load_dense(A, b, c)
        assert P.simplex(it_lim=0) == capi.EITLIM
        assert P.it_cnt == 0 and P.status == capi.FEAS
        assert P.obj == 0.0


@pytest.mark.parametrize("m,n", [(1, 1), (1, 7), (9, 1), (2, 511), (33, 513), (257, 31)])
def test_ragged_shapes(gpu, orc, m, n):
    A, b, c = synth.dense_lp(m, n, 100 + m + n)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        P.simplex()
    assert g.status == o.status and g.it_cnt == o.it_cnt
    assert np.array_equal(g.tableau(), o.tableau())


def test_empty_problem_is_rejected(gpu, orc):
    for api in (gpu, orc):
        P = api.create()
        assert P.simplex() == capi.EFAIL and P.status == capi.UNDEF


def test_clone_free_cycles_reuse_slabs(gpu, orc):
    A, b, c, U = synth.dense_ilp(16, 32, 5, 2)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        P.simplex()
    x = o.col_prim()
    frac = [j + 1 for j in range(32) if np.trunc(x[j]) != x[j]]
    for rep in range(50):
        j = frac[rep % len(frac)]
        res = []
        for P in (g, o):
            ch = P.copy()
            P.api.set_col_bnds(ch.h, j, UP, 0.0, float(np.floor(x[j - 1])) - (rep % 2))
            ch.simplex()
            res.append((ch.status, ch.it_cnt, ch.tableau()))
            del ch
        assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])


def test_columns_added_after_a_solve(gpu, orc):
    """glp_add_cols on a solved problem (readers only do it before the first solve): the basis is dropped and
    the next solve starts again from the slack basis, on both sides alike."""
    from mvolps_amd import capi, synth

    A, b, c = synth.dense_lp(20, 30, 3)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        assert P.simplex() == 0
        j0 = P.api.add_cols(P.h, 2)
        assert j0 == 31
        for j, (cost, ub) in ((31, (2.5, 4.0)), (32, (0.75, 1.0))):
            P.api.set_col_bnds(P.h, j, capi.DB, 0.0, ub)
            P.api.set_obj_coef(P.h, j, cost)
        ind = (np.arange(33)).astype(np.int32)
        for i in range(1, 21):
            val = np.concatenate(([0.0], A[i - 1], [0.5 + 0.01 * i, 0.25]))
            P.set_mat_row(i, ind, val)
        assert P.simplex() == 0
    assert g.status == o.status == capi.OPT
    assert g.it_cnt == o.it_cnt and g.obj == o.obj
    assert np.array_equal(g.tableau(), o.tableau())
    for x, y in zip(g.basis(), o.basis()):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n_edits", [3, 8, 9, 13])
def test_many_bound_edits_on_a_fresh_clone(gpu, orc, n_edits):
    """Bound edits of basic variables wait on the handle and ride in the next solve's control block (MAX_EDITS of
    them; more are flushed by launches), and a clone is only recorded until the next device call: edits made on a
    clone whose copy has not been launched yet must still land AFTER the copy (the flush launches the copy first)."""
    A, b, c, U = synth.dense_ilp(40, 80, 9, 3)
    g, o = lpgen.load_ilp(gpu, A, b, c, U), lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    x = o.col_prim()
    stat = o.col_stat()
    basic = [j + 1 for j in range(80) if stat[j] == capi.BS][:n_edits]
    assert len(basic) == n_edits
    kids = []
    for P in (g, o):
        Q = P.copy()  # recorded, not yet launched, on the device side
        for j in basic:
            P.api.set_col_bnds(Q.h, j, capi.DB, 0.0, float(np.floor(x[j - 1])))
        R = Q.copy()  # the clone of a handle with pending edits inherits them
        Q.simplex()
        R.simplex()
        kids.append((Q, R))
    (gq, gr), (oq, orr) = kids
    for a_, b_ in ((gq, oq), (gr, orr)):
        assert a_.status == b_.status and a_.it_cnt == b_.it_cnt
        assert np.array_equal(a_.tableau(), b_.tableau())


def test_values_read_behind_an_appended_cut_row(gpu, orc):
    """bs.cpp:249-261 appends a cut and THEN reads the branching variable's value.  Appending (or rewriting) a row behind
    the others leaves their values as they are, so the engine serves them from the mirrors of the last export instead of
    exporting again; the new row's own value, a bound edit or the next solve make it export.  Every value read on the way
    must be the oracle's bit for bit."""
    A, b, c, U = synth.dense_ilp(16, 32, 5, 3)
    g = lpgen.load_ilp(gpu, A, b, c, U)
    o = lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    n = 32
    ind = np.arange(n + 1, dtype=np.int32)
    rng = np.random.default_rng(3)
    for k in range(3):
        before = np.array(g.col_prim())
        obj_before = g.obj
        v = np.round(rng.normal(size=n) * 2)
        lb = float(v @ np.array(o.col_prim())) + 0.25
        for P in (g, o):
            r = P.api.add_rows(P.h, 1)
            P.set_mat_row(r, ind, np.concatenate([[0.0], v]))
            P.api.set_row_bnds(P.h, r, LO, lb, 0.0)
        # behind the cut, in front of the solve: the old rows' values, one at a time and all at once
        for j in (1, 7, n):
            assert g.api.get_col_prim(g.h, j) == o.api.get_col_prim(o.h, j) == before[j - 1]
        assert np.array_equal(np.array(g.col_prim()), before) and np.array_equal(np.array(o.col_prim()), before)
        assert g.obj == o.obj == obj_before
        # the appended row's own value is not in the mirrors: this one exports
        assert g.api.get_row_prim(g.h, g.m) == o.api.get_row_prim(o.h, o.m)
        assert np.array_equal(np.array(g.col_prim()), before)
        # a clone made behind the cut inherits the same view
        q = g.copy()
        assert np.array_equal(np.array(q.col_prim()), before)
        del q
        for P in (g, o):
            P.simplex()
        assert g.status == o.status and g.it_cnt == o.it_cnt, k
        assert np.array_equal(np.array(g.col_prim()), np.array(o.col_prim())), k
        assert np.array_equal(g.tableau(), o.tableau()), k
