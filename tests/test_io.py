"""CPU: LP / MPS readers (model layer only -- no engine call), event-stream writer and tree report."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import mvolps_amd
from mvolps_amd import bnb, capi, synth

from . import lpgen

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _lib():
    L = bnb.lib()
    L.mvx_read_lp.restype = C.c_int
    L.mvx_read_lp.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
    L.mvx_read_mps.restype = C.c_int
    L.mvx_read_mps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
    L.mvx_bnb_write_events.argtypes = [C.POINTER(bnb.BnbResult), C.c_char_p]
    L.mvx_bnb_print_tree.argtypes = [C.POINTER(bnb.BnbResult), C.c_char_p]
    L.mvx_bnb_solution_string.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(bnb.BnbResult), C.c_char_p, C.c_int]
    return L


def read(path, kind):
    L = _lib()
    P = mvolps_amd.api().create()
    rc = L.mvx_read_lp(P.h, None, path.encode()) if kind == "lp" else L.mvx_read_mps(P.h, 2, None, path.encode())
    return rc, P


@pytest.mark.parametrize("kind", ["lp", "mps"])
def test_f1_fixture_reads_back(kind):
    rc, P = read(os.path.join(GOLD, "f1." + kind), kind)
    assert rc == 0
    api = P.api
    assert (P.m, P.n) == (3, 5) and api.get_obj_dir(P.h) == capi.MAX and api.get_num_int(P.h) == 5
    A = np.array([[2, 3, 1, 4, 2], [4, 1, 2, 3, 5], [3, 4, 2, 1, 3]], float)
    for i in range(3):
        ind, val = P.get_mat_row(i + 1)
        assert list(ind) == [1, 2, 3, 4, 5] and np.array_equal(val, A[i])
        assert api.get_row_type(P.h, i + 1) == capi.UP and api.get_row_ub(P.h, i + 1) == [15, 23, 17][i]
    assert [api.get_obj_coef(P.h, j) for j in range(6)] == [0, 5, 4, 3, 7, 6]
    for j in range(1, 6):
        assert api.get_col_type(P.h, j) == capi.LO and api.get_col_lb(P.h, j) == 0.0
        assert api.get_col_kind(P.h, j) == capi.IV
        assert api.get_col_name(P.h, j) == ("x%d" % j).encode()


def test_lp_format_features(tmp_path):
    text = """\\ comment line
minimize
 cost: 3 x + 2.5 y - z + 10
subject to
 r1: x + y >= 2
 - x + 2 z =< 8
 eq: x - y + z = 1
 r4: 2 x + 3 y + 1 <= 12
bounds
 -5 <= x <= 5
 y free
 z <= 4
 w >= 1
 2 <= v
 u = 3
binary
 b1
generals
 x w
end
"""
    p = tmp_path / "t.lp"
    p.write_text(text)
    rc, P = read(str(p), "lp")
    assert rc == 0
    api = P.api
    names = [api.get_col_name(P.h, j).decode() for j in range(1, P.n + 1)]
    assert names == ["x", "y", "z", "w", "v", "u", "b1"]  # order of first appearance
    assert api.get_obj_dir(P.h) == capi.MIN and api.get_obj_coef(P.h, 0) == 10.0
    assert [api.get_obj_coef(P.h, j) for j in (1, 2, 3)] == [3.0, 2.5, -1.0]
    assert P.m == 4
    assert (api.get_row_type(P.h, 1), api.get_row_lb(P.h, 1)) == (capi.LO, 2.0)
    assert (api.get_row_type(P.h, 2), api.get_row_ub(P.h, 2)) == (capi.UP, 8.0)
    assert api.get_row_type(P.h, 3) == capi.FX and api.get_row_lb(P.h, 3) == 1.0
    assert api.get_row_ub(P.h, 4) == 11.0  # constant moved to the right-hand side
    ind, val = P.get_mat_row(2)
    assert list(ind) == [1, 3] and list(val) == [-1.0, 2.0]
    col = {nm: j + 1 for j, nm in enumerate(names)}
    assert (api.get_col_type(P.h, col["x"]), api.get_col_lb(P.h, col["x"]), api.get_col_ub(P.h, col["x"])) == (capi.DB, -5.0, 5.0)
    assert api.get_col_type(P.h, col["y"]) == capi.FR
    assert (api.get_col_type(P.h, col["z"]), api.get_col_ub(P.h, col["z"])) == (capi.DB, 4.0)
    assert (api.get_col_type(P.h, col["w"]), api.get_col_lb(P.h, col["w"])) == (capi.LO, 1.0)
    assert api.get_col_lb(P.h, col["v"]) == 2.0
    assert api.get_col_type(P.h, col["u"]) == capi.FX and api.get_col_lb(P.h, col["u"]) == 3.0
    assert api.get_col_kind(P.h, col["b1"]) == capi.BV and api.get_col_kind(P.h, col["x"]) == capi.IV
    assert api.get_col_kind(P.h, col["y"]) == capi.CV


@pytest.mark.parametrize("text", ["", "maximize\n obj: x\nsubject to\n c: x <= \nend\n", "maximize\n x\nsubject to\n c1: x + y 3\nend\n",
                                  "maximize\n obj: x\nsubject to\n c: x <= 1\n"])
def test_lp_syntax_errors_return_nonzero(tmp_path, text):
    p = tmp_path / "bad.lp"
    p.write_text(text)
    rc, _ = read(str(p), "lp")
    assert rc != 0  # util.cpp:284-287 then calls exit(-1)
    assert read(str(tmp_path / "missing.lp"), "lp")[0] != 0


def test_mps_bounds_ranges_and_objsense(tmp_path):
    text = """NAME t
ROWS
 N cost
 G lim1
 L lim2
 E eq1
COLUMNS
    x cost 1.0 lim1 1.0
    x lim2 1.0
    y cost 2.0 lim1 1.0
    y eq1 -1.0
    z cost -1.0 eq1 1.0
RHS
    rhs cost -7.5 lim1 4.0
    rhs lim2 9.0 eq1 0.5
RANGES
    rng lim2 3.0
BOUNDS
 UP bnd x 4.0
 MI bnd y
 BV bnd z
ENDATA
"""
    p = tmp_path / "t.mps"
    p.write_text(text)
    rc, P = read(str(p), "mps")
    assert rc == 0
    api = P.api
    assert api.get_obj_dir(P.h) == capi.MIN and api.get_obj_coef(P.h, 0) == 7.5
    assert (api.get_row_type(P.h, 1), api.get_row_lb(P.h, 1)) == (capi.LO, 4.0)
    assert (api.get_row_type(P.h, 2), api.get_row_lb(P.h, 2), api.get_row_ub(P.h, 2)) == (capi.DB, 6.0, 9.0)
    assert api.get_row_type(P.h, 3) == capi.FX
    assert (api.get_col_type(P.h, 1), api.get_col_ub(P.h, 1)) == (capi.DB, 4.0)
    assert api.get_col_type(P.h, 2) == capi.FR
    assert api.get_col_kind(P.h, 3) == capi.BV


def test_event_stream_and_tree_report(orc, tmp_path):
    """message.h line format and the bs.cpp:329-343 tree report, from a run over the oracle's engine table."""
    L = _lib()
    A, b, c, U = synth.dense_ilp(6, 12, 2, 3)
    P = lpgen.load_ilp(orc, A, b, c, U)
    tab = bnb.table_from(orc)
    pr = bnb.BnbParams()
    L.mvx_bnb_default_params(C.byref(pr))
    pr.reference_quirks = 0
    res = bnb.BnbResult()
    L.mvx_branchAndBound(C.cast(C.pointer(tab), C.c_void_p), P.h, C.byref(pr), C.byref(res))
    ev, tree = str(tmp_path / "ev.txt"), str(tmp_path / "tree.txt")
    assert L.mvx_bnb_write_events(C.byref(res), ev.encode()) == 0
    assert L.mvx_bnb_print_tree(C.byref(res), tree.encode()) == 0
    lines = open(ev).read().splitlines()
    assert lines[-1] == "END" and len(lines) == res.n_events + 1
    f = lines[0].split()
    assert f[1:5] == ["pregnant", "1", "0", "M"] and f[-2:] == ["1", "2"]
    kinds = {l.split()[1] for l in lines[:-1]}
    assert {"pregnant", "branched", "candidate", "integer"} <= kinds
    for l in lines[:-1]:
        w = l.split()
        assert len(w) == {"pregnant": 8, "integer": 6, "infeasible": 7, "fathomed": 5, "branched": 10, "candidate": 6}[w[1]]
    t = open(tree).read().splitlines()
    assert t[0] == "[I = Integral node, F = Infeasible node, B = Worse bound node]"
    assert t[1] == "-1" and len(t) == res.n_nodes + 1
    depth = {1: 0}
    for oid in range(2, res.n_nodes + 1):
        depth[oid] = depth[res.parent[oid]] + 1
    for l in t[1:]:
        d = len(l) - len(l.lstrip(" "))
        oid = int(l.strip().lstrip("-").split()[0])
        assert d == depth[oid]
    buf = C.create_string_buffer(4096)
    assert L.mvx_bnb_solution_string(C.cast(C.pointer(tab), C.c_void_p), P.h, C.byref(res), buf, 4096) == 0
    s = buf.value.decode()
    assert s.startswith("[%d] Solution is: " % res.incumbent_oid) and s.rstrip().endswith("= 71")
    L.mvx_bnb_free_result(C.byref(res))
