"""GPU: the chained primal path (a chain's steps chosen by k_chain -- one launch, a cluster of resident workgroups --
or by k_pc / k_pr, two launches per step; ONE bulk pass k_fbc3 + k_fpatch for the whole chain) leaves the same bits as
the oracle, whatever the chain length and whoever chooses: the pivots after the first are chosen from column / row
slices carried through the earlier pivots of the chain entry by entry, so every case of that carry (pivot row, pivot
column, the same row leaving twice, a column re-entering, bounds and status moved by the earlier swaps) has to come out
as the bulk update would have left it.  Small LPs, resident-tableau kernel off, chain length forced; every case is
solved twice, the second time on the slab the first one gave back (a race inside a kernel shows as a difference between
two runs long before it shows against the oracle)."""
import numpy as np
import pytest

from mvolps_amd import capi, synth

from . import lpgen
from .test_gpu_parity import assert_same_state

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["cluster", "two-launch"])
def chained(gpu, request):
    """Both ways of choosing a chain's steps: k_chain (default) and k_pc / k_pr."""
    gpu.set_persist(0)
    gpu.set_cluster(request.param)
    yield gpu
    gpu.set_chain(0)
    gpu.set_cluster(1)
    gpu.set_persist(1)


def cluster_counts(api):
    import ctypes as C

    a, b = C.c_longlong(0), C.c_longlong(0)
    api.cluster_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def twice(api, load, check):
    """The same problem on a fresh handle, and again on the slab that handle has given back by then."""
    for rep in range(2):
        P = load(api)
        P.rc = P.simplex()
        check(rep, P)
        del P  # the last reference: the handle is deleted, its slab goes back to the cache


@pytest.mark.parametrize("chain", [2, 5, 16])
def test_dense_lps_full_solves(chained, orc, chain):
    chained.set_chain(chain)
    for (m, n, seed) in ((64, 128, 1), (100, 37, 5), (256, 512, 12345), (300, 700, 11), (512, 1024, 12345)):
        A, b, c = synth.dense_lp(m, n, seed)

        def load(api):
            P = api.create()
            P.load_dense(A, b, c)
            return P

        o = load(orc)
        assert o.simplex() == 0

        def check(rep, g):
            assert g.rc == 0
            assert_same_state(g, o, "dense %dx%d chain %d run %d" % (m, n, chain, rep))

        twice(chained, load, check)


@pytest.mark.parametrize("chain", [2, 3, 16])
def test_bounded_columns_bring_flips_into_the_chains(chained, orc, chain):
    """Boxed columns: a chain ends where the next step is a bound flip (the ordinary path takes it), an entering column
    may be one that left a few steps earlier (its bounds and status come from the chain's own record)."""
    chained.set_chain(chain)
    rng = np.random.default_rng(5)
    for (m, n, seed) in ((96, 400, 3), (200, 300, 8), (300, 700, 11)):
        A, b, c = synth.dense_lp(m, n, seed)
        col_b = [(capi.DB, 0.0, float(rng.integers(1, 4)) / 2) if j % 3 else (capi.LO, 0.0, 0.0) for j in range(n)]
        row_b = [(capi.UP, 0.0, float(v)) for v in b]

        def load(api):
            P = api.create()
            P.load_general(A, row_b, col_b, c, direction=capi.MAX)
            return P

        o = load(orc)
        o.rc = o.simplex()

        def check(rep, g):
            assert g.rc == o.rc
            assert_same_state(g, o, "boxed %dx%d chain %d run %d" % (m, n, chain, rep))

        twice(chained, load, check)


@pytest.mark.parametrize("chain", [2, 7, 16])
def test_limits_that_fall_inside_a_chain(chained, orc, chain):
    """Pivot limits 1, 2, 3, 5, 8, ...: the last chain of a call holds only what the limit leaves."""
    chained.set_chain(chain)
    A, b, c = synth.dense_lp(200, 500, 21)
    g, o = chained.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
    lim, prev = 1, 1
    for call in range(40):
        rcs = [P.simplex(it_lim=lim) for P in (g, o)]
        assert rcs[0] == rcs[1]
        assert_same_state(g, o, "call %d limit %d chain %d" % (call, lim, chain))
        if rcs[0] == 0:
            break
        lim, prev = lim + prev, lim
    assert g.status == capi.OPT


@pytest.mark.parametrize("chain", [2, 16])
def test_degenerate_lps_and_stalling(chained, orc, chain):
    """Degenerate vertices: stall counters run through the chain (a chain ends at the stall limit; Bland's rule and the
    perturbation stay with the generic path), the same row may leave twice within one chain."""
    chained.set_chain(chain)
    for case in ((358, 124, 41004, 0.8), (288, 252, 41002, 0.8), (314, 433, 41000, 0.95)):
        m, n, seed, dens = case
        A, b, c = lpgen.degenerate_lp(m, n, seed, frac0=dens)
        o = lpgen.load_degenerate(orc, A, b, c)
        o.rc = o.simplex()

        def check(rep, g):
            assert g.rc == o.rc
            assert g.pert_cnt == o.pert_cnt and g.bland_cnt == o.bland_cnt
            assert_same_state(g, o, "degenerate %s chain %d run %d" % (case, chain, rep))

        twice(chained, lambda api: lpgen.load_degenerate(api, A, b, c), check)
    for name, (A, b, c) in lpgen.CYCLING.items():
        A, b, c = (np.array(v, float) for v in (A, b, c))
        g, o = chained.create(), orc.create()
        for P in (g, o):
            P.load_dense(A, b, c)
            assert P.simplex() == 0
        assert_same_state(g, o, name)


def test_cluster_launches_are_counted_and_none_gives_up(gpu, orc):
    """The default path is the cluster kernel: launches are made, none aborts (an abort would silently move the rest
    of the process to k_pc / k_pr)."""
    gpu.set_persist(0)
    gpu.set_cluster(1)
    try:
        before = cluster_counts(gpu)
        A, b, c = synth.dense_lp(300, 700, 3)
        g, o = gpu.create(), orc.create()
        for P in (g, o):
            P.load_dense(A, b, c)
            assert P.simplex() == 0
        assert_same_state(g, o, "300x700")
        after = cluster_counts(gpu)
        assert after[0] > before[0], "no k_chain launch was made"
        assert after[1] == before[1], "a k_chain launch gave up waiting for its peers"
    finally:
        gpu.set_persist(1)


@pytest.mark.parametrize("m,n,seed,lim", [(40, 9000, 4, 120), (9000, 60, 6, 60), (8300, 8400, 2, 70)])
def test_wide_cluster_geometry(gpu, orc, m, n, seed, lim):
    """More than 8192 columns or rows: two columns / rows to a thread of k_chain (the CPT = RPT = 2 kernel), chains as
    long as its LDS holds."""
    gpu.set_cluster(1)
    before = cluster_counts(gpu)
    A, b, c = synth.dense_lp(m, n, seed)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        P.rc = P.simplex(it_lim=lim)
    assert g.rc == o.rc
    assert_same_state(g, o, "%dx%d, %d pivots" % (m, n, lim))
    after = cluster_counts(gpu)
    assert after[0] > before[0] and after[1] == before[1]


def test_chain_length_by_size_is_the_default(gpu, orc):
    """Default setting (length by tableau size): 1024x4096 is in the band that chains four pivots per pass."""
    gpu.set_chain(0)
    A, b, c = synth.dense_lp(1024, 4096, 7)
    g, o = gpu.create(), orc.create()
    for P in (g, o):
        P.load_dense(A, b, c)
        assert P.simplex(it_lim=150) == capi.EITLIM
    assert_same_state(g, o, "1024x4096, 150 pivots")


# ---------------------------------------------------------------- chained dual steps (dual_chain in k_select)
@pytest.fixture
def dual_chained(gpu):
    yield gpu
    gpu.set_dual_chain(0)


@pytest.mark.parametrize("chain", [2, 5, 8])
def test_dual_chains_in_branch_and_bound(dual_chained, orc, chain):
    """Every warm-started child runs the dual simplex: trees, events and pivot counts equal the oracle's with the dual
    pivots chained 2, 5 and 8 to a pass (window mode and node at a time, plain and with GMI cut rows appended)."""
    from mvolps_amd import bnb
    from oracle import oracle

    from .test_gpu_bnb import same_result

    dual_chained.set_dual_chain(chain)
    for case, kw in (((10, 20, 4, 3), dict(quirks=0)), ((16, 32, 5, 2), dict(quirks=1, max_nodes=600)),
                     ((10, 20, 4, 3), dict(quirks=0, cut_strat=1)), ((16, 32, 5, 2), dict(quirks=1, cut_strat=1, max_nodes=300))):
        A, b, c, U = synth.dense_ilp(*case)
        ref = oracle.branch_and_bound(lpgen.load_ilp(orc, A, b, c, U), **kw)
        for window in (1, 64):
            got = bnb.branch_and_bound(lpgen.load_ilp(dual_chained, A, b, c, U), window=window, **kw)
            same_result(got, ref)


@pytest.mark.parametrize("chain", [3, 8])
def test_dual_chains_on_512x1024_children(dual_chained, orc, chain):
    """The calibrated config-5 instance: root + both children of its first eight fractional columns, each child
    hundreds of dual pivots (n + 1 = 1025 columns on 1024 lanes: the lane that owns the entering column can be one
    that has no row to work on, which is where an unsynchronised read of the pivot element once raced with its
    rescaling), solved twice over (recycled slabs)."""
    import json
    import os

    fx = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config5.json")))
    A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
    dual_chained.set_dual_chain(chain)
    g, o = synth.load_ilp(dual_chained, A, b, c, U), synth.load_ilp(orc, A, b, c, U)
    for P in (g, o):
        assert P.simplex() == 0
    assert_same_state(g, o, "root")
    x = o.col_prim()
    frac = [j + 1 for j in range(len(x)) if abs(x[j] - round(x[j])) > 1e-9][:8]
    ref = {}
    for j in frac:
        for up in (0, 1):
            k = o.copy()
            orc.set_col_bnds(k.h, j, capi.DB, *((float(np.ceil(x[j - 1])), 1.0) if up else (0.0, float(np.floor(x[j - 1])))))
            k.simplex()
            ref[(j, up)] = k
    for rep in range(2):
        for j in frac:
            for up in (0, 1):
                k = g.copy()
                dual_chained.set_col_bnds(k.h, j, capi.DB, *((float(np.ceil(x[j - 1])), 1.0) if up else (0.0, float(np.floor(x[j - 1])))))
                k.simplex()
                assert_same_state(k, ref[(j, up)], "child %d/%d rep %d chain %d" % (j, up, rep, chain))


@pytest.mark.parametrize("chain", [3, 8])
def test_dual_chains_on_general_lps_with_warm_starts(dual_chained, orc, chain):
    dual_chained.set_dual_chain(chain)
    rng = np.random.default_rng(11)
    seen = set()
    for trial in range(80):
        A, row_b, col_b, c, direction = lpgen.random_general_lp(rng, mmax=60, nmax=90)
        g, o = dual_chained.create(), orc.create()
        for P in (g, o):
            P.load_general(A, row_b, col_b, c, direction=direction)
            P.rc = P.simplex()
        assert g.rc == o.rc
        assert_same_state(g, o, "general %d" % trial)
        j = int(rng.integers(1, A.shape[1] + 1))
        v = float(rng.integers(-2, 4))
        for P in (g, o):
            P.api.set_col_bnds(P.h, j, capi.DB, v, v + 1.0)
            P.rc = P.simplex()
        assert g.rc == o.rc
        assert_same_state(g, o, "general %d warm" % trial)
        seen.add(g.status)
    assert len(seen) >= 2
