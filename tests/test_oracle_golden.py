"""CPU: the oracle against the committed HiGHS golden vectors (tests/golden/golden.json).

The reference holds no fixtures and its LP arithmetic is GLPK's, absent
here; the goldens come from an independent solver and pin status / objective / primal values.
"""
import json
import os

import numpy as np
import pytest

from mvolps_amd import capi, synth
from mvolps_amd.capi import MAX

from . import lpgen

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))
RTOL = 1e-9


def rel(a, b):
    return abs(a - b) / max(1.0, abs(b))


@pytest.mark.parametrize("case", [g for g in GOLD["dense"] if g["m"] <= 1024], ids=lambda g: "%dx%d_s%d" % (g["m"], g["n"], g["seed"]))
def test_dense_lp_objective_and_x(orc, case):
    A, b, c = synth.dense_lp(case["m"], case["n"], case["seed"])
    P = orc.create()
    P.load_dense(A, b, c)
    assert P.simplex() == 0
    assert P.status == capi.OPT
    assert rel(P.obj, case["obj"]) <= RTOL
    x = P.col_prim()
    assert np.all(x >= -1e-9) and np.all(A @ x <= b + 1e-7)
    assert rel(float(c @ x), case["obj"]) <= RTOL
    if "x" in case:
        assert np.allclose(x, np.array(case["x"]), rtol=1e-7, atol=1e-8)


def test_general_bounds_cases(orc):
    rng = np.random.default_rng(7)
    n_opt = 0
    for g in GOLD["general"]:
        A, row_b, col_b, c, direction = lpgen.random_general_lp(rng)
        P = orc.create()
        P.load_general(A, row_b, col_b, c, c0=1.5, direction=direction)
        P.simplex()
        lo, hi = lpgen.bounds_arrays(row_b)
        cl, cu = lpgen.bounds_arrays(col_b)
        if g["highs_status"] == 0:
            n_opt += 1
            assert P.status == capi.OPT, g["trial"]
            assert rel(P.obj, g["obj"]) <= RTOL, g["trial"]
            x = P.col_prim()
            ra = A @ x
            assert np.all(x >= cl - 1e-7) and np.all(x <= cu + 1e-7)
            assert np.all(ra >= lo - 1e-7) and np.all(ra <= hi + 1e-7)
            assert np.allclose(ra, P.row_prim(), atol=1e-8)
        else:
            # every case is feasible by construction, so "not optimal" can only mean unbounded
            # (HiGHS reports some of these as status 2 from presolve; status 3 otherwise)
            assert P.status == capi.UNBND, g["trial"]
    assert n_opt >= 60


def test_f1_lp_relaxation_and_ilp(orc):
    """Config 1: LP relaxation 36.6667 at (0,0,17/3,5/3,4/3); ILP optimum 35."""
    from oracle import oracle

    f1 = GOLD["ilp"][0]
    A, b, c = np.array(f1["A"]), np.array(f1["b"]), np.array(f1["c"])
    P = lpgen.load_ilp(orc, A, b, c, np.inf)
    P.simplex()
    assert rel(P.obj, f1["lp_obj"]) <= RTOL
    assert np.allclose(P.col_prim(), f1["lp_x"], atol=1e-9)
    assert rel(P.obj, 110.0 / 3.0) <= RTOL
    r = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, np.inf), quirks=0)
    assert rel(r["best_lower"], f1["ilp_obj"]) <= RTOL and f1["ilp_obj"] == 35.0
    assert np.allclose(r["x"], f1["ilp_x"], atol=1e-8)


@pytest.mark.parametrize("case", GOLD["ilp"][1:], ids=lambda g: g["name"])
def test_ilp_branch_and_bound_matches_milp(orc, case):
    """Repaired mode (reference_quirks=0) must find the true ILP optimum, FIFO and best-bound alike."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(case["m"], case["n"], case["seed"], int(case["U"]))
    P = lpgen.load_ilp(orc, A, b, c, U)
    P.simplex()
    assert rel(P.obj, case["lp_obj"]) <= RTOL
    strategies = (0, 1) if case["m"] <= 16 else (1,)
    for ns in strategies:
        r = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), node_strat=ns, quirks=0)
        assert not r["hit_limit"]
        assert rel(r["best_lower"], case["ilp_obj"]) <= RTOL
        x = np.array(r["x"])
        assert np.all(np.abs(x - np.round(x)) <= 1e-8) and np.all(A @ x <= b + 1e-7)


@pytest.mark.parametrize("case", GOLD["degenerate"], ids=lambda g: g.get("name") or "%dx%d_s%d" % (g["m"], g["n"], g["seed"]))
def test_stalling_lps_reach_the_optimum(orc, case):
    """Cycling textbook LPs and massively degenerate ones: plain Dantzig pricing spins on them (chvatal,
    150x150, 200x300, 250x400 never finish); devex pricing plus the anti-stalling rules must end on
    HiGHS's optimum."""
    if "name" in case:
        A, b, c = (np.array(v, float) for v in lpgen.CYCLING[case["name"]])
        P = orc.create()
        P.load_dense(A, b, c)
    else:
        A, b, c = lpgen.degenerate_lp(case["m"], case["n"], case["seed"])
        P = lpgen.load_degenerate(orc, A, b, c)
    assert P.simplex() == 0
    assert P.status == capi.OPT
    assert rel(P.obj, case["obj"]) <= RTOL
    x = P.col_prim()
    assert np.all(x >= -1e-9) and np.all(A @ x <= b + 1e-7)
    if case.get("m", 0) >= 150:
        assert P.pert_cnt == 1 and P.it_cnt < 20000  # the bound perturbation, not luck, gets these out of the vertex


@pytest.mark.parametrize("case", GOLD["setcover"], ids=lambda g: "%dx%d_s%d" % (g["m"], g["n"], g["seed"]))
def test_minimisation_ilp_repaired_mode_matches_milp(orc, case):
    """Set-cover ILPs (min): LP relaxation through the dual simplex from the slack basis, and the repaired
    branch-and-bound, which bounds and prunes a minimisation problem as one (bs.cpp:172,210 never do)."""
    from oracle import oracle

    A, c = lpgen.setcover_ilp(case["m"], case["n"], case["seed"])
    P = lpgen.load_setcover(orc, A, c)
    assert P.simplex() == 0 and P.status == capi.OPT
    assert rel(P.obj, case["lp_obj"]) <= RTOL
    for node_strat in (0, 1):
        r = oracle.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=0, node_strat=node_strat, max_nodes=50000)
        assert r["has_incumbent"] and not r["hit_limit"]
        assert rel(r["best_lower"], case["ilp_obj"]) <= RTOL
        x = np.array(r["x"])[-A.shape[1]:]
        assert np.all(A @ x >= 1 - 1e-7) and rel(float(c @ x), case["ilp_obj"]) <= RTOL


def test_config5_fixture_is_the_oracles_record(orc):
    """tests/golden/config5.json (BASELINE config 5) was written by tests/golden/make_config5.py from the oracle; a
    prefix of its tree is re-derived here, so a fixture recorded from anything else would be caught on the CPU."""
    import json
    import os

    from mvolps_amd import synth, treedigest
    from oracle import oracle

    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5.json")))
    assert fx["source"].startswith("oracle") and not fx["full"]["hit_limit"]
    assert 8000 <= fx["full"]["count"] <= 20000 and fx["full"]["incumbent_updates"] >= 2  # "~10k-node tree" that closes
    A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
    r = oracle.branch_and_bound(synth.load_ilp(orc, A, b, c, U), quirks=0, max_nodes=200)
    rec = fx["prefix"]["200"]
    assert treedigest.digest(r) == rec["sha256"] and r["total_pivots"] == rec["pivots"]


def test_config5_and_cut_path_optima_equal_highs_milp(orc):
    """Independent pins for whole B&B runs (tests/golden/milp_pins.json, HiGHS milp): the optimum recorded for the
    calibrated config-5 tree, and a 128x256 ILP closed by the oracle WITH repaired GMI cuts (config-3 path)."""
    from oracle import oracle

    here = os.path.dirname(__file__)
    pins = json.load(open(os.path.join(here, "golden", "milp_pins.json")))
    fx = json.load(open(os.path.join(here, "golden", "config5.json")))
    assert (fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"]) == tuple(pins["config5"][k] for k in ("m", "n", "seed", "U", "cap"))
    best = float.fromhex(fx["full"]["best_lower"])
    assert abs(best - pins["config5"]["milp_obj"]) <= 1e-9 * abs(best)
    pin = pins["cut_ilps"][0]
    A, b, c, U = synth.dense_ilp(pin["m"], pin["n"], pin["seed"], pin["U"], pin["cap"])
    r = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0, cut_strat=1)
    assert not r["hit_limit"] and r["count"] > 5000
    assert abs(r["best_lower"] - pin["milp_obj"]) <= 1e-9 * abs(pin["milp_obj"])
