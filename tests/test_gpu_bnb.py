"""GPU: the branch-and-bound driver over the gfx950 engine against the oracle's restatement of
bs.cpp on the oracle's engine.  Branching decisions, oids, prune labels and picks must be
bit-exact; LP bounds are compared bitwise too (same arithmetic on both sides)."""
import json
import os

import numpy as np
import pytest

from mvolps_amd import bnb, capi, synth

from . import lpgen

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def same_result(a, b):
    for k in ("n_nodes", "parent", "prune", "count", "has_incumbent", "incumbent_oid", "hit_limit", "total_pivots"):
        assert a[k] == b[k], k
    assert a["events"] == b["events"]
    assert a["node_bound"] == b["node_bound"]
    assert a["x"] == b["x"]
    assert a["best_lower"] == b["best_lower"] or (np.isinf(a["best_lower"]) and np.isinf(b["best_lower"]))


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("case", [(6, 12, 2, 3), (10, 20, 4, 3), (16, 32, 5, 2)], ids=lambda c: "%dx%d" % (c[0], c[1]))
def test_bnb_bit_exact_vs_oracle(gpu, orc, case, quirks):
    from oracle import oracle

    m, n, seed, U = case
    A, b, c, U = synth.dense_ilp(m, n, seed, U)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, max_nodes=600)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=quirks, max_nodes=600)
    same_result(got, ref)


def test_bnb_best_bound_and_cuts(gpu, orc):
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(10, 20, 4, 3)
    for kw in (dict(node_strat=1, quirks=0), dict(cut_strat=1, max_nodes=150), dict(cut_strat=1, var_strat=2, node_strat=1, max_nodes=150)):
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
        got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
        same_result(got, ref)
        if kw.get("cut_strat"):
            got_all = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), lazy_pool=0, **kw)
            same_result(got_all, ref)


def test_ilp_optimum_matches_milp_golden(gpu):
    """Repaired mode on the GPU finds the HiGHS milp optimum (tests/golden)."""
    for case in GOLD["ilp"][1:5]:
        A, b, c, U = synth.dense_ilp(case["m"], case["n"], case["seed"], int(case["U"]))
        r = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), node_strat=1, quirks=0)
        assert not r["hit_limit"]
        assert abs(r["best_lower"] - case["ilp_obj"]) <= 1e-9 * max(1.0, abs(case["ilp_obj"]))
        x = np.array(r["x"])
        assert np.all(np.abs(x - np.round(x)) <= 1e-8) and np.all(A @ x <= b + 1e-7)
    f1 = GOLD["ilp"][0]
    A, b, c = np.array(f1["A"]), np.array(f1["b"]), np.array(f1["c"])
    r = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, np.inf), quirks=0)
    assert abs(r["best_lower"] - 35.0) <= 1e-9 * 35 and np.allclose(r["x"], f1["ilp_x"], atol=1e-8)


def test_config3_shape_without_cuts_first_nodes(gpu, orc):
    """BASELINE config 3 shape (m=512, n=1024, integer data) with the reference's default -cm 0 (no cuts), bug-compatible
    mode: first nodes of the tree.  The cut path at this size is test_config3_cut_path_at_size below."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(512, 1024, seed=12345, U=3)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), max_nodes=6)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), max_nodes=6)
    same_result(got, ref)
    assert got["total_pivots"] > 100


_C3_REF = {}


def _config3_oracle(orc, quirks, **kw):
    """One oracle run per mode, shared by the parametrised cases below (0.1 s of CPU per node with cuts)."""
    from oracle import oracle

    key = (quirks,) + tuple(sorted(kw.items()))
    if key not in _C3_REF:
        A, b, c, U = synth.dense_ilp(512, 1024, seed=12345, U=3)
        _C3_REF[key] = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=quirks, cut_strat=1, **kw)
    return _C3_REF[key]


@pytest.mark.parametrize("quirks", [1, 0])
@pytest.mark.parametrize("lazy_pool,window", [(1, 64), (0, 64), (1, 1)], ids=["lazy-w64", "allcuts-w64", "lazy-serial"])
def test_config3_cut_path_at_size(gpu, orc, quirks, lazy_pool, window):
    """BASELINE config 3: ILP 512x1024 with GMI cuts on (-cm 1), 320 nodes of the FIFO tree, against the oracle's
    restatement of bs.cpp:249-258 + gmi.cpp:11-117 + cut.cpp:11-46 -- bug-compatible (persistent pool, positional
    back-substitution) and repaired.  Device side: tableau-row read, k_add_rows / k_rowcomb into a spare row of the
    slab, the children inherit the row and warm-start in the dual simplex; window 64 replays the pool in queue order."""
    ref = _config3_oracle(orc, quirks, max_nodes=320)
    A, b, c, U = synth.dense_ilp(512, 1024, seed=12345, U=3)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=quirks, cut_strat=1, max_nodes=320, lazy_pool=lazy_pool, window=window)
    same_result(got, ref)
    assert got["count"] == 320 and got["prune"].count(4) > 300  # a real tree: branchings, hence cuts, all the way


def test_deep_cut_rows_cross_the_spare_rows(gpu, orc):
    """Best-bound order dives: 2500 nodes of a 128x256 ILP with the bug-compatible cut path reach depth 52, one appended
    cut row per level -- past the 32 spare rows every slab keeps behind row m (ROW_SPARE), so the slab has to grow
    (grow_rows) under the driver, more than once along a path; the result still equals the oracle's.  (At 512x1024
    300 best-bound nodes only reach depth 33; the repaired mode crosses the boundary at size in the next test.)"""
    from oracle import oracle

    from mvolps_amd import treedigest

    A, b, c, U = synth.dense_ilp(128, 256, 7, 3)
    kw = dict(quirks=1, cut_strat=1, node_strat=1, max_nodes=2500)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
    assert treedigest.depth(ref) > 40
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
    same_result(got, ref)


def test_config3_efficacy_selected_cuts_at_size(gpu, orc):
    """-cf honoured: the 10 % most effective of each node's GMI cuts are appended, ~9 rows per
    branching, so depth 4 is already past the spare rows; window 64 against the oracle."""
    ref = _config3_oracle(orc, 0, cut_select=1, cut_chance=0.1, max_nodes=40)
    A, b, c, U = synth.dense_ilp(512, 1024, seed=12345, U=3)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0, cut_strat=1, cut_select=1, cut_chance=0.1, max_nodes=40)
    same_result(got, ref)


def test_cli_end_to_end_f1(gpu, orc, tmp_path):
    """Config 1 plumbing: `mvolps -f f1.lp` (2test.cpp main) -> reader -> B&B on the GPU -> event stream, tree report and
    solution line.  The three texts are compared WHOLE with what the same driver writes from a run over the oracle's
    engine (message.h:141-226 line format, bs.cpp:329-345 report): every event line, every tree line, the solution."""
    import ctypes as C
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "mvolps_amd", "bin", "mvolps")
    # the oracle-side rendering of the same run
    f1 = GOLD["ilp"][0]
    A, b, c = np.array(f1["A"]), np.array(f1["b"]), np.array(f1["c"])
    L = bnb.lib()
    L.mvx_bnb_write_events.argtypes = [C.POINTER(bnb.BnbResult), C.c_char_p]
    L.mvx_bnb_print_tree.argtypes = [C.POINTER(bnb.BnbResult), C.c_char_p]
    L.mvx_bnb_solution_string.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(bnb.BnbResult), C.c_char_p, C.c_int]
    P = lpgen.load_ilp(orc, A, b, c, np.inf)
    tab = bnb.table_from(orc)
    pr = bnb.make_params(quirks=0)
    res = bnb.BnbResult()
    L.mvx_branchAndBound(C.cast(C.pointer(tab), C.c_void_p), P.h, C.byref(pr), C.byref(res))
    ev_ref, tree_ref = str(tmp_path / "ev_ref.txt"), str(tmp_path / "tree_ref.txt")
    assert L.mvx_bnb_write_events(C.byref(res), ev_ref.encode()) == 0 and L.mvx_bnb_print_tree(C.byref(res), tree_ref.encode()) == 0
    buf = C.create_string_buffer(4096)
    assert L.mvx_bnb_solution_string(C.cast(C.pointer(tab), C.c_void_p), P.h, C.byref(res), buf, 4096) == 0
    sol_ref = buf.value.decode().strip()
    L.mvx_bnb_free_result(C.byref(res))
    assert sol_ref.endswith("= 35") and len(open(ev_ref).read().splitlines()) > 10
    for ext in ("lp", "mps"):
        ev = tmp_path / ("ev_%s.txt" % ext)
        r = subprocess.run([exe, "-f", os.path.join(root, "tests", "golden", "f1." + ext), "--repaired", "-v", "--events", str(ev)],
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert open(ev).read() == open(ev_ref).read()  # the whole event stream, line for line
        out = r.stdout.splitlines()
        t0 = out.index("[I = Integral node, F = Infeasible node, B = Worse bound node]")
        tree = open(tree_ref).read().splitlines()
        assert out[t0:t0 + len(tree)] == tree  # the whole tree report
        assert sol_ref in r.stdout and "Solution is: 3*(x[3] = 7) + 7*(x[4] = 2) + 0 = 35" in r.stdout
    r = subprocess.run([exe, "-f", "nope.txt"], capture_output=True, text=True)
    assert "Unrecognized filetype" in r.stdout


def test_five_thousand_node_tree_three_ways(gpu, orc):
    """A ~4.6k-node FIFO tree: oracle restatement, serial GPU driver and the window coordinator with
    batched node solves (32 nodes per round) all produce the same tree, decisions and incumbent."""
    from mvolps_amd import dist_bnb
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(16, 32, 5, 2)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), quirks=0)
    assert ref["count"] > 4000 and not ref["hit_limit"]
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0)
    same_result(got, ref)
    eng = dist_bnb.HipNodeEngine(0)
    win = json.loads(json.dumps(dist_bnb.branch_and_bound(eng, lpgen.load_ilp(gpu, A, b, c, U), quirks=0, per_rank=32)))
    ref_j = json.loads(json.dumps(ref))
    for k in ("n_nodes", "parent", "prune", "count", "events", "node_bound", "x", "total_pivots", "incumbent_oid", "best_lower"):
        assert win[k] == ref_j[k], k
    assert abs(ref["best_lower"] - 210.0) <= 1e-9 * 210  # HiGHS milp optimum (tests/golden: ilp_16x32_s5)


def _config5():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fx = json.load(open(os.path.join(root, "tests", "golden", "config5.json")))
    A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
    return fx, (A, b, c, U)


def _matches(r, rec):
    from mvolps_amd import treedigest

    assert r["count"] == rec["count"] and r["n_nodes"] == rec["n_nodes"] and r["total_pivots"] == rec["pivots"]
    assert r["hit_limit"] == rec["hit_limit"] and r["has_incumbent"] == rec["has_incumbent"]
    assert treedigest.digest(r) == rec["sha256"]


def test_config5_full_tree_equals_the_oracle_record(gpu):
    """BASELINE config 5: the calibrated 512x1024 ILP (tests/golden/config5.json, written by tests/golden/make_config5.py
    from the ORACLE alone).  Its FIFO tree finishes after ~15.7k nodes with the incumbent replaced several times
    (bs.cpp:172-174) and bound pruning at work (bs.cpp:210).  The windowed GPU driver reproduces the oracle's event
    stream, prune labels and parents (sha256 over all of them), node for node, and the incumbent itself."""
    fx, (A, b, c, U) = _config5()
    rec = fx["full"]
    assert not rec["hit_limit"] and rec["incumbent_updates"] >= 2 and rec["prune_counts"]["3"] > 0
    r = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0)
    _matches(r, rec)
    assert float(r["best_lower"]).hex() == rec["best_lower"] and r["incumbent_oid"] == rec["incumbent_oid"]
    assert [(j + 1, v) for j, v in enumerate(r["x"]) if v != 0.0] == [tuple(t) for t in rec["x_nonzero"]]
    # ... and the optimum is the one an independent solver finds (HiGHS milp, tests/golden/milp_pins.json)
    pin = _milp_pins()["config5"]["milp_obj"]
    assert abs(r["best_lower"] - pin) <= 1e-9 * abs(pin)


def _milp_pins():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(__file__), "golden", "milp_pins.json")))


@pytest.mark.parametrize("which", [0, 1])
def test_cut_path_closes_on_the_milp_optimum(gpu, orc, which):
    """Config-3 path end to end against an independent solver: a 128x256 ILP whose FIFO tree closes WITH repaired GMI cuts
    (about 10 000 nodes); the optimum equals the HiGHS milp optimum, tree and pivot counts equal the oracle's."""
    from oracle import oracle

    pin = _milp_pins()["cut_ilps"][which]
    A, b, c, U = synth.dense_ilp(pin["m"], pin["n"], pin["seed"], pin["U"], pin["cap"])
    kw = dict(quirks=0, cut_strat=1)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
    assert not got["hit_limit"] and got["count"] > 5000
    assert abs(got["best_lower"] - pin["milp_obj"]) <= 1e-9 * abs(pin["milp_obj"])
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
    same_result(got, ref)


@pytest.mark.parametrize("window", [1, 32])
def test_config5_prefix_other_drivers(gpu, window):
    """The first 3000 nodes through the node-at-a-time driver and a 32-wide window: same digest as the oracle's."""
    fx, (A, b, c, U) = _config5()
    r = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=0, max_nodes=3000, window=window)
    _matches(r, fx["prefix"]["3000"])


def test_config5_bug_compatible_prefix(gpu):
    """The same instance with bs.cpp's own bounds handling (GLP_UP / GLP_LO drop the opposite bound, bs.cpp:274,282):
    infeasibility all but disappears and the tree no longer closes; first 1000 nodes against the oracle's record."""
    fx, (A, b, c, U) = _config5()
    r = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), quirks=1, max_nodes=1000)
    _matches(r, fx["bugcompat_prefix"]["1000"])


def test_repaired_gmi_cuts_on_gpu(gpu, orc):
    """Non-default cut path (reference_quirks = 0): repaired GMI + efficacy selection with -cf, GPU vs oracle
    bit-exact, optimum equal to the HiGHS milp golden."""
    from oracle import oracle

    case = GOLD["ilp"][4]  # ilp_10x20_s4
    A, b, c, U = synth.dense_ilp(case["m"], case["n"], case["seed"], int(case["U"]))
    for kw in (dict(cut_select=0), dict(cut_select=1, cut_chance=0.4)):
        kw.update(node_strat=1, quirks=0, cut_strat=1)
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
        got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
        same_result(got, ref)
        assert abs(got["best_lower"] - case["ilp_obj"]) <= 1e-9 * abs(case["ilp_obj"])


def test_minimisation_ilp_on_gpu(gpu, orc):
    """Set-cover (min) in repaired mode: dual simplex from the slack basis, direction-aware bounding; device
    driver (window and node-at-a-time, queue and best-bound order) against the oracle's restatement."""
    from oracle import oracle

    A, c = lpgen.setcover_ilp(40, 60, 3)
    for node_strat, window in ((0, 64), (0, 1), (1, 1)):
        ref = oracle.branch_and_bound(lpgen.load_setcover(orc, A, c), quirks=0, node_strat=node_strat, max_nodes=5000)
        got = bnb.branch_and_bound(lpgen.load_setcover(gpu, A, c), quirks=0, node_strat=node_strat, max_nodes=5000, window=window)
        same_result(got, ref)
        assert abs(got["best_lower"] - 22.0) < 1e-9


def test_division_fixup_keeps_a_long_run_bit_exact(gpu, orc):
    """The fp64 division of the device can be one ulp off when the quotient lies next to a midpoint (here:
    -0x1.6666666666663p-1 / -0x1.ffffffffffffbp-1 in the scaled pivot row of node 634); xdiv() in kernels.hip and in
    the oracle repairs the quotient from its exact residual.  Without it this 641-node run differs from the
    oracle in one event field by one ulp (found by scripts/fuzz.py, seed 98765)."""
    from oracle import oracle

    A, b, c, U = synth.dense_ilp(8, 25, 7016, 3)
    kw = dict(quirks=0, max_nodes=1500)
    ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
    got = bnb.branch_and_bound(lpgen.load_ilp(gpu, A, b, c, U), **kw)
    same_result(got, ref)
    assert ref["count"] == 641


def test_first_bnb_of_a_fresh_process_is_the_same_tree(gpu):
    """The first B&B of a process creates what later ones reuse (batch contexts, slab arenas, streams): whatever is set
    up on the way must be ordered with the launches that use it.  A fresh interpreter runs the first 300 nodes of the
    wide 512x1024 tree as its very first engine work; pivots and digest must be those of this (warm) process.  (Round 3:
    a null-stream memset of a new batch context's job counters could land inside the first batch and hand LPs out twice
    -- trees still right after tableau refreshes, 10x the pivots.)"""
    import subprocess
    import sys

    from mvolps_amd import treedigest

    code = (
        "import json, mvolps_amd\n"
        "from mvolps_amd import bnb, synth, treedigest\n"
        "api = mvolps_amd.api()\n"
        "A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)\n"
        "r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=300, window=64)\n"
        "print(json.dumps({'count': r['count'], 'pivots': r['total_pivots'], 'digest': treedigest.digest(r)}))\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
    here = bnb.branch_and_bound(synth.load_ilp(gpu, A, b, c, U), quirks=0, max_nodes=300, window=64)
    for _ in range(2):
        out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        got = json.loads(out.stdout.strip().splitlines()[-1])
        assert got == {"count": here["count"], "pivots": here["total_pivots"], "digest": treedigest.digest(here)}
