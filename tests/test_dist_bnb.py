"""CPU, world_size 2 over gloo: the multi-GPU coordinator (mvolps_amd/dist_bnb.py) must reproduce the
serial driver's tree, decisions and incumbent exactly (serial equivalence)."""
import json

import numpy as np
import pytest

from mvolps_amd import bnb, synth

from . import dist_helpers, lpgen


def canon(r):
    r = json.loads(json.dumps(r))  # tuples -> lists, same float repr path as the workers
    return r


def assert_same(a, b):
    for k in ("n_nodes", "parent", "prune", "count", "has_incumbent", "incumbent_oid", "hit_limit", "total_pivots", "events",
              "node_bound", "x"):
        assert a[k] == b[k], k
    assert a["best_lower"] == b["best_lower"]


@pytest.mark.parametrize("quirks,max_nodes", [(0, 0), (1, 300)])
@pytest.mark.parametrize("per_rank", [1, 3])
def test_world2_matches_serial(orc, tmp_path, quirks, max_nodes, per_rank):
    case = (8, 16, 3, 2)
    m, n, seed, U = case
    A, b, c, U = synth.dense_ilp(m, n, seed, U)
    serial = canon(bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=max_nodes, table=bnb.table_from(orc)))
    kw = dict(quirks=quirks, max_nodes=max_nodes, per_rank=per_rank)
    res = dist_helpers.run_world(2, case, kw, str(tmp_path))
    assert_same(res[0], res[1])  # every rank ends with the same replicated result
    assert_same(res[0], serial)
    assert serial["count"] > 20


def test_world2_minimisation_problem(orc, tmp_path):
    """Repaired mode on a minimisation ILP: the coordinator turns the incumbent / prune compares round like the
    serial driver does."""
    case = ("setcover", 40, 60, 3)
    serial = canon(bnb.branch_and_bound(lpgen.load_case(orc, case), quirks=0, table=bnb.table_from(orc)))
    res = dist_helpers.run_world(2, case, dict(quirks=0, per_rank=2), str(tmp_path))
    assert_same(res[0], res[1])
    assert_same(res[0], serial)
    assert abs(serial["best_lower"] - 22.0) < 1e-9 and serial["count"] > 3


def test_world1_is_the_serial_driver(orc):
    """No process group: the coordinator degenerates to the serial loop."""
    from mvolps_amd import dist_bnb

    A, b, c, U = synth.dense_ilp(6, 12, 2, 3)
    eng = dist_helpers.OracleNodeEngine()
    for vs in (0, 1, 2):
        got = canon(dist_bnb.branch_and_bound(eng, lpgen.load_ilp(orc, A, b, c, U), var_strat=vs, quirks=0))
        ref = canon(bnb.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), var_strat=vs, quirks=0, table=bnb.table_from(orc)))
        assert_same(got, ref)
    # integral at the root (repaired mode keeps the solution, bs.cpp:144-149 drops it)
    case = ("setcover", 30, 40, 2)
    got = canon(dist_bnb.branch_and_bound(eng, lpgen.load_case(orc, case), quirks=0))
    ref = canon(bnb.branch_and_bound(lpgen.load_case(orc, case), quirks=0, table=bnb.table_from(orc)))
    assert_same(got, ref)
    assert got["count"] == 0 and got["has_incumbent"] and got["best_lower"] == 33.0


def test_pack_unpack_roundtrip(orc):
    eng = dist_helpers.OracleNodeEngine()
    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    root = lpgen.load_ilp(orc, A, b, c, U)
    P = root.copy()
    P.simplex()
    x = P.col_prim()
    j = [k + 1 for k in range(20) if np.trunc(x[k]) != x[k]][0]
    from mvolps_amd import capi

    orc.set_col_bnds(P.h, j, capi.UP, 0.0, float(np.floor(x[j - 1])))
    Q = eng.unpack(root, eng.pack(P))
    assert np.array_equal(P.tableau(), Q.tableau())
    P.simplex()
    Q.simplex()
    assert P.it_cnt == Q.it_cnt and np.array_equal(P.tableau(), Q.tableau()) and P.obj == Q.obj
