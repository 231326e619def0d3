"""CPU: the product's C++ branch-and-bound driver (mvolps_amd/csrc/bnb.cpp) run over the ORACLE's
LP-engine table, against the oracle's own C restatement of bs.cpp.  Both must take identical
decisions: events, oids, parents, prune labels, picks -- integers bit-exact, doubles bit-exact too
(the LP engine underneath is the same one here)."""
import numpy as np
import pytest

from mvolps_amd import bnb, capi, synth

from . import lpgen


def oracle_table(orc):
    return bnb.table_from(orc)


def same_result(a, b):
    for k in ("n_nodes", "parent", "prune", "count", "has_incumbent", "incumbent_oid", "hit_limit", "total_pivots"):
        assert a[k] == b[k], k
    assert a["events"] == b["events"]
    assert a["node_bound"] == b["node_bound"]
    assert a["x"] == b["x"]
    assert a["best_lower"] == b["best_lower"] or (np.isinf(a["best_lower"]) and np.isinf(b["best_lower"]))


CASES = [(4, 8, 1, 3), (6, 12, 2, 3), (8, 16, 3, 2), (10, 20, 4, 3)]


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("node_strat", [0, 1])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_driver_matches_oracle_restatement(orc, case, node_strat, quirks):
    from oracle import oracle

    m, n, seed, U = case
    A, b, c, U = synth.dense_ilp(m, n, seed, U)
    tab = oracle_table(orc)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), node_strat=node_strat, quirks=quirks, max_nodes=3000)
    got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), node_strat=node_strat, quirks=quirks, max_nodes=3000, table=tab)
    same_result(got, ref)
    assert got["count"] > 5


@pytest.mark.parametrize("window", [1, 2, 7, 64])
def test_window_mode_is_serial_equivalent(orc, window):
    """mvx_bnb_params.window: the front W nodes solved together, decisions replayed in queue order."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    tab = oracle_table(orc)
    assert tab.simplex_batch  # the oracle library exports a (sequential) batch entry for this test
    for quirks, mx in ((0, 0), (1, 500)):
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=mx)
        got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=mx, table=tab, window=window)
        same_result(got, ref)


@pytest.mark.parametrize("lazy", [0, 1])
@pytest.mark.parametrize("var_strat", [0, 1, 2])
def test_driver_with_gmi_cuts_and_var_strategies(orc, var_strat, lazy):
    """-cm 1 (GMI) path incl. the pool-persistence quirk; lazy_pool must not change anything."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(6, 12, 2, 3)
    tab = oracle_table(orc)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), var_strat=var_strat, cut_strat=1, max_nodes=400)
    got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), var_strat=var_strat, cut_strat=1, max_nodes=400, lazy_pool=lazy, table=tab)
    same_result(got, ref)


def test_tree_labels_follow_the_reference(orc):
    """bs.cpp:26-52: root pid 0 / direction M; S2 gets the even oid (R), S3 the odd one (L); FIFO order."""
    A, b, c, U = synth.dense_ilp(6, 12, 2, 3)
    r = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0, table=oracle_table(orc))
    ev = r["events"]
    assert ev[0][:4] == (0, 1, 0, 0)  # pregnant event of the root: oid 1, pid 0, direction M
    branched = [e for e in ev if e[0] == 4]
    cands = [e for e in ev if e[0] == 5]
    assert len(cands) == 2 * len(branched)
    for k, e in enumerate(branched):
        s2, s3 = cands[2 * k], cands[2 * k + 1]
        assert s2[1] % 2 == 0 and s2[3] == 1 and s3[1] == s2[1] + 1 and s3[3] == 2
        assert s2[2] == e[1] and s3[2] == e[1]
    # FIFO (util.cpp:165): nodes are solved in oid order
    solved = [e[1] for e in ev if e[0] == 0]
    assert solved == sorted(solved)
    assert r["prune"].count(4) == len(branched)  # NONE label stays on branched nodes


def test_print_info_and_getfract(orc):
    from oracle import oracle

    L = bnb.lib()
    for x in (0.0, 0.25, -0.25, 3.999999, -7.5, 2.0000000000000004, -0.0, 1e300):
        assert L.mvx_getFract(x) == orc.getFract(x)
    assert L.mvx_getFract(-0.25) == 0.75
    A, b, c, U = synth.dense_ilp(8, 16, 3, 2)
    P = lpgen.load_ilp(orc, A, b, c, U)
    P.simplex()
    tab = oracle_table(orc)
    for q in (1, 0):
        st, viol = bnb.print_info(P, quirks=q, table=tab)
        buf = (np.zeros(17, dtype=np.int32))
        import ctypes as C

        cnt = C.c_int(0)
        st_o = orc.printInfo_ex(P.h, q, buf.ctypes.data_as(C.POINTER(C.c_int)), C.byref(cnt))
        assert st == st_o and viol == buf[: cnt.value].tolist()
    # integer columns with a zero objective coefficient are never branched on (util.cpp:437)
    x = P.col_prim()
    frac = [j + 1 for j in range(16) if np.trunc(x[j]) != x[j]]
    P.api.set_obj_coef(P.h, frac[0], 0.0)
    P.simplex()
    st, viol = bnb.print_info(P, quirks=1, table=tab)
    assert all(P.api.get_obj_coef(P.h, j) != 0 for j in viol)


def test_generate_cut3_matches_oracle(orc):
    import ctypes as C

    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    P = lpgen.load_ilp(orc, A, b, c, U)
    P.simplex()
    tab = oracle_table(orc)
    stat = P.col_stat()
    n_cut = 0
    for j in range(1, 21):
        got = bnb.generate_cut3(P, j, table=tab)
        inds = np.zeros(21, dtype=np.int32)
        vals = np.zeros(21)
        lb = C.c_double(0)
        rc = orc.generateCut3(P.h, j, inds.ctypes.data_as(C.POINTER(C.c_int)), vals.ctypes.data_as(C.POINTER(C.c_double)), C.byref(lb))
        if stat[j - 1] != capi.BS:
            assert got is None and rc == -1  # gmi.cpp:23-27
            continue
        n_cut += 1
        assert rc == 0 and got is not None
        assert np.array_equal(got[0], inds) and np.array_equal(got[1], vals) and got[2] == lb.value
        assert got[1][0] == got[2]  # vals[0] = rhs = lb (gmi.cpp:95,109)
    assert n_cut >= 3


@pytest.mark.parametrize("cut_select,cut_chance", [(0, 1.0), (1, 0.3), (1, 1.0)])
def test_repaired_gmi_cuts_keep_the_optimum_and_match_oracle(orc, cut_select, cut_chance):
    """SURVEY 8(f) rank 4 (non-default): repaired GMI cuts + efficacy selection honouring -cf.  Valid cuts:
    the ILP optimum of the HiGHS golden is preserved; product driver and oracle restatement agree exactly."""
    import json
    import os

    from oracle import oracle

    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))["ilp"]
    tab = oracle_table(orc)
    for case in gold[2:6]:
        A, b, c, U = synth.dense_ilp(case["m"], case["n"], case["seed"], int(case["U"]))
        kw = dict(node_strat=1, quirks=0, cut_strat=1, cut_select=cut_select, cut_chance=cut_chance)
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
        got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), table=tab, **kw)
        same_result(got, ref)
        plain = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), table=tab, node_strat=1, quirks=0)
        assert abs(got["best_lower"] - case["ilp_obj"]) <= 1e-9 * abs(case["ilp_obj"])
        assert got["count"] < plain["count"]  # the cuts do prune


@pytest.mark.parametrize("window", [1, 64])
@pytest.mark.parametrize("node_strat", [0, 1])
def test_minimisation_problem_in_repaired_mode(orc, node_strat, window):
    """Direction-aware bounding (reference_quirks = 0): the C++ driver and the oracle's restatement agree on a
    minimisation ILP, in queue order, best-bound order and window mode; bug-compatible mode keeps bs.cpp's
    maximiser compares on both sides."""
    from oracle import oracle

    A, c = lpgen.setcover_ilp(40, 60, 3)
    tab = oracle_table(orc)
    ref = oracle.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=0, node_strat=node_strat, max_nodes=5000)
    got = bnb.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=0, node_strat=node_strat, max_nodes=5000, table=tab, window=window)
    same_result(got, ref)
    assert abs(got["best_lower"] - 22.0) < 1e-9 and got["count"] > 3
    # an LP relaxation that is integral at the root: bs.cpp:144-149 leaves without recording it, repaired mode keeps it
    A2, c2 = lpgen.setcover_ilp(30, 40, 2)
    r2 = bnb.branch_and_bound(lpgen.load_setcover(orc, A2, c2), quirks=0, node_strat=node_strat, table=tab, window=window)
    same_result(r2, oracle.branch_and_bound(lpgen.load_setcover(orc, A2, c2), quirks=0, node_strat=node_strat))
    assert r2["count"] == 0 and r2["has_incumbent"] and r2["best_lower"] == 33.0 and r2["incumbent_oid"] == 1
    refq = oracle.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=1, node_strat=node_strat, max_nodes=300)
    gotq = bnb.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=1, node_strat=node_strat, max_nodes=300, table=tab, window=window)
    same_result(gotq, refq)


def test_overlapped_and_inline_child_solves_agree(orc, monkeypatch):
    """The window driver solves each round's children on a worker thread while it replays the next window;
    MVX_BNB_SYNC=1 keeps everything on the calling thread.  Same result either way, and the same as the
    oracle's node-at-a-time restatement (infeasible children included: their pop-time re-solve is done ahead)."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    tab = oracle_table(orc)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0, max_nodes=4000)
    got = {}
    for mode in ("async", "sync"):
        if mode == "sync":
            monkeypatch.setenv("MVX_BNB_SYNC", "1")
        got[mode] = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0, max_nodes=4000, table=tab, window=8)
        same_result(got[mode], ref)
    assert any(p == 1 for p in ref["prune"]), "the case must contain infeasible nodes"
    assert ref["count"] > 50


def test_reference_cut_formula_with_non_finite_rows(orc):
    """Bug-compatible cuts (gmi.cpp:73 with absent bounds) can leave inf / NaN coefficients in the model; a later
    back-substitution (gmi.cpp:81-89) multiplies them by zero weights, which is NaN, not zero.  The driver must
    go through every row like bs.cpp does (found by scripts/fuzz.py, seed 777)."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(6, 11, 7025, 2)
    kw = dict(quirks=1, cut_strat=1, max_nodes=120)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
    got = bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), table=oracle_table(orc), **kw)
    same_result(got, ref)
