"""Synthetic LP / ILP generators.

PRNG: splitmix64 (state += 0x9E3779B97F4A7C15; mix 30/0xBF58476D1CE4E5B9,
27/0x94D049BB133111EB, 31), u = (z >> 11) * 2**-53.  Vectorised with numpy uint64.
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


class SplitMix64:
    def __init__(self, seed):
        self.state = np.uint64(seed)

    def uniform(self, count):
        """Next `count` draws of U[0,1) as float64."""
        with np.errstate(over="ignore"):
            k = np.arange(1, count + 1, dtype=np.uint64)
            z = self.state + k * _GAMMA
            self.state = self.state + np.uint64(count) * _GAMMA
            z = (z ^ (z >> np.uint64(30))) * _M1
            z = (z ^ (z >> np.uint64(27))) * _M2
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def dense_lp(m, n, seed=12345):
    """max c'x, Ax <= b, x >= 0 with A ~ U[0,1), b = n/4 + u*n/4, c ~ U[0,1).
    Draw order: A row-major, then b, then c.  A > 0 => bounded; b > 0 => slack basis feasible."""
    rng = SplitMix64(seed)
    A = rng.uniform(m * n).reshape(m, n)
    b = n / 4.0 + rng.uniform(m) * (n / 4.0)
    c = rng.uniform(n)
    return A, b, c


def dense_ilp(m, n, seed=12345, U=3, cap=0.4):
    """max c'x, Ax <= b, 0 <= x <= U integer; A, c integer in [1,20]; b_i = floor(cap*sum_j A_ij).
    Draw order: A row-major, then c."""
    rng = SplitMix64(seed)
    A = 1.0 + np.floor(rng.uniform(m * n) * 20.0).reshape(m, n)
    c = 1.0 + np.floor(rng.uniform(n) * 20.0)
    b = np.floor(cap * A.sum(axis=1))
    return A, b, c, float(U)


def load_ilp(api, A, b, c, U):
    """The ILP of dense_ilp as a problem handle of `api`: max c'x, Ax <= b, 0 <= x <= U, every column integer
    (GLP_IV) with a finite upper bound (GLP_DB) -- SURVEY.md section 8(d)."""
    from .capi import DB, IV, LO, MAX, UP

    m, n = A.shape
    P = api.create()
    colb = [(DB, 0.0, U) if np.isfinite(U) else (LO, 0.0, 0.0)] * n
    P.load_general(A, [(UP, 0.0, float(bi)) for bi in b], colb, c, kinds=[IV] * n, direction=MAX)
    return P
