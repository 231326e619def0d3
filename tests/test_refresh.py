"""Tableau refresh (DESIGN.md "Re-inversion"): a solve that ends optimal on a handle with enough pivots behind it looks
at the residual of the row equations and, above the tolerance, rebuilds the tableau from the model for the same
basis.  Same rule and arithmetic in the oracle and on the device.  The default tolerance (1e-9) is far above what these
chains accumulate (~1e-15), so the rule is driven here with tolerance 0: every look refreshes."""
import numpy as np
import pytest

from mvolps_amd import bnb, capi, synth

from . import lpgen


@pytest.fixture
def forced(orc):
    """check every 8 pivots, refresh whatever the residual; restores the defaults afterwards"""
    apis = [orc]

    def arm(extra=None):
        if extra is not None:
            apis.append(extra)
        for a in apis:
            a.set_refresh(8, 0.0)

    yield arm
    for a in apis:
        a.set_refresh(1024, 1e-9)


@pytest.mark.parametrize("shape", [(40, 64, 3), (96, 160, 5), (128, 256, 1)])
def test_oracle_refresh_keeps_the_vertex(orc, forced, shape):
    m, n, seed = shape
    A, b, c = synth.dense_lp(m, n, seed)
    P = orc.create()
    P.load_dense(A, b, c)
    P.simplex()
    forced()
    Q = orc.create()
    Q.load_dense(A, b, c)
    Q.simplex()
    assert orc.get_refresh_cnt(Q.h) >= 1 and orc.get_refresh_cnt(P.h) == 0
    assert Q.status == capi.OPT and abs(Q.obj - P.obj) <= 1e-12 * abs(P.obj)
    assert set(P.basis()[0][1:].tolist()) == set(Q.basis()[0][1:].tolist())  # same basis, rows in another order
    assert orc.row_residual(Q.h) <= 1e-13
    assert np.allclose(P.col_prim(), Q.col_prim(), rtol=0, atol=1e-11)


def test_oracle_bnb_result_does_not_depend_on_refreshes(orc, forced):
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(16, 32, 5, 2)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0)
    forced()
    got = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0)
    assert got["prune"] == ref["prune"] and got["parent"] == ref["parent"]
    assert abs(got["best_lower"] - ref["best_lower"]) <= 1e-9 * abs(ref["best_lower"])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(40, 64, 3), (96, 160, 5), (300, 500, 2)])
def test_device_refresh_is_bit_exact_with_the_oracle(gpu, orc, forced, shape):
    m, n, seed = shape
    forced(gpu)
    A, b, c = synth.dense_lp(m, n, seed)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        assert P.simplex() == 0
    assert gpu.get_refresh_cnt(g.h) == orc.get_refresh_cnt(o.h) >= 1
    assert g.it_cnt == o.it_cnt and np.array_equal(g.tableau(), o.tableau())
    for u, v in zip(g.basis(), o.basis()):
        assert np.array_equal(u, v)
    assert gpu.row_residual(g.h) == orc.row_residual(o.h)
    # warm-started children after the refresh
    x = o.col_prim()
    j = int(np.argmax(x)) + 1
    for P in (g, o):
        ch = P.copy()
        P.api.set_col_bnds(ch.h, j, capi.UP, 0.0, float(np.floor(x[j - 1]) - 1.0))
        ch.simplex()
        P._ch = ch
    assert g._ch.it_cnt == o._ch.it_cnt and np.array_equal(g._ch.tableau(), o._ch.tableau())


@pytest.mark.gpu
def test_the_simplex_runs_again_on_the_rebuilt_tableau(gpu, orc, forced):
    """After a refresh the device runs the simplex on the rebuilt tableau, as the oracle does (no "already optimal"
    shortcut: a rebuilt tableau may be off by more than the tolerance): the leg leaves a device time behind, and with
    zero tolerances -- where rounding noise of the rebuild decides whether pivots follow -- the device still equals the
    oracle bit for bit."""
    forced(gpu)
    tol = (0.0, 0.0, 1e-9)
    for k, (m, n) in enumerate(((30, 50), (60, 90), (40, 120), (96, 160))):
        A, b, c = lpgen.degenerate_lp(m, n, 4100 + k, frac0=0.8)
        g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
        rcs = [P.simplex(tol=tol) for P in (g, o)]
        assert rcs[0] == rcs[1] and g.status == o.status
        assert gpu.get_refresh_cnt(g.h) == orc.get_refresh_cnt(o.h)
        assert g.it_cnt == o.it_cnt and np.array_equal(g.tableau(), o.tableau())
        if gpu.get_refresh_cnt(g.h) and g.status == capi.OPT:
            assert gpu.last_solve_ms(g.h) > 0.0, "the leg after the refresh never reached the device"


@pytest.mark.gpu
def test_bnb_with_forced_refreshes_on_the_device(gpu, orc, forced):
    """Refreshes inside batched child solves (window driver, device work queue) and after migration-free clones."""
    from oracle import oracle

    forced(gpu)
    A, b, c, U = synth.dense_ilp(24, 48, 6, 2)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0, max_nodes=800)
    for window in (64, 1):
        got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0, max_nodes=800, window=window)
        for k in ("events", "prune", "parent", "node_bound", "total_pivots", "count", "x"):
            assert got[k] == ref[k], (window, k)


@pytest.mark.gpu
def test_long_chain_keeps_the_row_equations(gpu, orc):
    """A lineage with more than 5000 pivots behind it (best-bound dive on the 512x1024 ILP): the residual of the row
    equations stays below 1e-9 under the default rule, and the device equals the oracle along the way."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3)
    kw = dict(quirks=0, node_strat=1, max_nodes=260)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
    assert got["events"] == ref["events"] and got["total_pivots"] == ref["total_pivots"] > 5000
    # one explicit chain: root, then keep tightening the first fractional column
    g, o = lpgen.load_ilp(gpu, A, b, c, U), lpgen.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        P.simplex()
    steps = 0
    while o.it_cnt < 5200 and steps < 400:
        x = o.col_prim()
        frac = [k + 1 for k in range(1024) if abs(x[k] - round(x[k])) > 1e-9]
        if not frac or o.status != capi.OPT:
            break
        j = frac[0]
        for P in (g, o):
            P.api.set_col_bnds(P.h, j, capi.DB, 0.0, float(np.floor(x[j - 1])))
            P.simplex()
        assert g.it_cnt == o.it_cnt and g.status == o.status
        steps += 1
    assert np.array_equal(g.tableau(), o.tableau())
    if o.status == capi.OPT:
        assert gpu.row_residual(g.h) == orc.row_residual(o.h) <= 1e-9
