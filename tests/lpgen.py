"""Shared random-LP builders for the tests (seeded, small)."""
import numpy as np

from mvolps_amd.capi import DB, FR, FX, LO, MAX, MIN, UP


def random_general_lp(rng, mmax=9, nmax=10):
    """Feasible-by-construction LP with every bound type on rows and columns."""
    m = int(rng.integers(2, mmax))
    n = int(rng.integers(2, nmax))
    A = np.round(rng.normal(size=(m, n)) * 3)
    A[rng.random((m, n)) < 0.3] = 0
    x0 = rng.integers(0, 4, size=n).astype(float)
    act = A @ x0
    row_b, col_b = [], []
    for i in range(m):
        t = int(rng.choice([LO, UP, DB, FX, FR], p=[0.25, 0.35, 0.2, 0.1, 0.1]))
        l = act[i] - rng.integers(0, 3)
        u = act[i] + rng.integers(0, 3)
        if t == FX:
            l = u = act[i]
        if t == DB and l == u:
            u = l + 1
        row_b.append((t, float(l), float(u)))
    for j in range(n):
        t = int(rng.choice([LO, UP, DB, FX, FR], p=[0.4, 0.1, 0.35, 0.05, 0.1]))
        l = x0[j] - rng.integers(0, 3)
        u = x0[j] + rng.integers(0, 4)
        if t == FX:
            l = u = x0[j]
        if t == DB and l == u:
            u = l + 1
        col_b.append((t, float(l), float(u)))
    c = np.round(rng.normal(size=n) * 5)
    direction = int(rng.choice([MIN, MAX]))
    return A, row_b, col_b, c, direction


def bounds_arrays(bnds):
    lo = np.array([l if t in (LO, DB, FX) else -np.inf for t, l, u in bnds])
    hi = np.array([u if t in (UP, DB) else (l if t == FX else np.inf) for t, l, u in bnds])
    return lo, hi


def load_ilp(api, A, b, c, U):
    from mvolps_amd import synth

    return synth.load_ilp(api, A, b, c, U)


def degenerate_lp(m, n, seed, frac0=0.9):
    """max c x, A x <= b, 0 <= x <= 2, integer data, about 90 % of b equal to zero: the slack basis is a
    massively degenerate vertex on which plain Dantzig pricing stalls (200x300 seed 4 never leaves it)."""
    rng = np.random.default_rng(seed)
    A = rng.integers(-3, 4, size=(m, n)).astype(float)
    b = np.where(rng.random(m) < frac0, 0.0, rng.integers(1, 5, size=m).astype(float))
    c = rng.integers(1, 6, size=n).astype(float)
    return A, b, c


def load_degenerate(api, A, b, c):
    from mvolps_amd.capi import DB, UP
    P = api.create()
    P.load_general(A, [(UP, 0.0, float(x)) for x in b], [(DB, 0.0, 2.0)] * A.shape[1], c)
    return P


# textbook LPs on which Dantzig pricing with lowest-index ties cycles (max c x, A x <= b, x >= 0)
CYCLING = {
    "beale": ([[0.25, -8, -1, 9], [0.5, -12, -0.5, 3], [0, 0, 1, 0]], [0, 0, 1.0], [0.75, -20, 0.5, -6]),
    "chvatal": ([[0.5, -5.5, -2.5, 9], [0.5, -1.5, -0.5, 1], [1, 0, 0, 0]], [0, 0, 1.0], [10, -57, -9, -24.0]),
}


def setcover_ilp(m, n, seed, dens=0.15):
    """min c x, A x >= 1, x binary (A 0/1, every row covered): a MINIMISATION ILP -- bs.cpp bounds and prunes as a
    maximiser whatever the direction (bs.cpp:172,210), so only the repaired mode gets these right."""
    rng = np.random.default_rng(seed)
    A = (rng.random((m, n)) < dens).astype(float)
    for i in range(m):
        if A[i].sum() == 0:
            A[i, rng.integers(n)] = 1.0
    c = rng.integers(1, 10, size=n).astype(float)
    return A, c


def load_setcover(api, A, c):
    from mvolps_amd.capi import DB, IV, LO, MIN
    m, n = A.shape
    P = api.create()
    P.load_general(A, [(LO, 1.0, 0.0)] * m, [(DB, 0.0, 1.0)] * n, c, kinds=[IV] * n, direction=MIN)
    return P


def load_case(api, case):
    """(m, n, seed, U) -> dense_ilp; ("setcover", m, n, seed) -> setcover_ilp"""
    from mvolps_amd import synth
    if case[0] == "setcover":
        return load_setcover(api, *setcover_ilp(*case[1:]))
    A, b, c, U = synth.dense_ilp(*case)
    return load_ilp(api, A, b, c, U)
