#!/usr/bin/env python3
"""Generates tests/golden/*.json with scipy 1.15.3 HiGHS (an independent solver).

The reference has no tests, fixtures or sample inputs and GLPK is not in
the image, so these vectors -- not GLPK outputs -- pin the oracle at the API boundary: status,
objective, primal values.  Run from the repo root:  python tests/golden/make_golden.py
(the two large cases are taken from a cached run unless --big is given: 4096x8192 takes ~80 s).
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import Bounds, LinearConstraint, linprog, milp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mvolps_amd import synth  # noqa: E402
from mvolps_amd.capi import MAX, MIN  # noqa: E402
from tests import lpgen  # noqa: E402


def dense_cases(big):
    out = []
    sizes = [(3, 5, 1), (17, 33, 2), (64, 128, 12345), (128, 256, 12345), (256, 512, 12345), (100, 37, 5), (64, 128, 1), (64, 128, 2),
             (64, 128, 3), (512, 1024, 12345)]
    for (m, n, seed) in sizes:
        A, b, c = synth.dense_lp(m, n, seed)
        r = linprog(-c, A_ub=A, b_ub=b, bounds=(0, None), method="highs-ds")
        assert r.status == 0
        e = {"m": m, "n": n, "seed": seed, "obj": float(-r.fun)}
        if m * n <= 128 * 256:
            e["x"] = [float(v) for v in r.x]
        out.append(e)
    cached = {(1024, 2048, 12345): 1062.2098379939434, (4096, 8192, 12345): 4196.778725972496}
    for (m, n, seed), obj in cached.items():
        if big:
            A, b, c = synth.dense_lp(m, n, seed)
            r = linprog(-c, A_ub=A, b_ub=b, bounds=(0, None), method="highs-ds")
            assert r.status == 0
            obj = float(-r.fun)
        out.append({"m": m, "n": n, "seed": seed, "obj": obj})
    return out


def general_cases():
    """Random LPs with every bound type; kept only when HiGHS proves optimality (feasible by construction)."""
    rng = np.random.default_rng(7)
    out = []
    for trial in range(120):
        A, row_b, col_b, c, direction = lpgen.random_general_lp(rng)
        lo, hi = lpgen.bounds_arrays(row_b)
        cl, cu = lpgen.bounds_arrays(col_b)
        sg = 1.0 if direction == MIN else -1.0
        Aub, bub = [], []
        for i in range(A.shape[0]):
            if np.isfinite(hi[i]):
                Aub.append(A[i]); bub.append(hi[i])
            if np.isfinite(lo[i]):
                Aub.append(-A[i]); bub.append(-lo[i])
        r = linprog(sg * c, A_ub=np.array(Aub) if Aub else None, b_ub=np.array(bub) if Aub else None,
                    bounds=list(zip(cl, cu)), method="highs-ds")
        e = {"trial": trial, "highs_status": int(r.status)}
        if r.status == 0:
            e["obj"] = float(sg * r.fun + 1.5)
        out.append(e)
    return out


def ilp_cases():
    out = []
    A = np.array([[2, 3, 1, 4, 2], [4, 1, 2, 3, 5], [3, 4, 2, 1, 3]], float)
    b = np.array([15, 23, 17.0])
    c = np.array([5, 4, 3, 7, 6.0])
    r = linprog(-c, A_ub=A, b_ub=b, bounds=(0, None), method="highs-ds")
    ri = milp(-c, constraints=LinearConstraint(A, -np.inf, b), integrality=np.ones(5), bounds=Bounds(0, np.inf))
    out.append({"name": "F1", "A": A.tolist(), "b": b.tolist(), "c": c.tolist(), "U": None, "lp_obj": float(-r.fun),
                "lp_x": [float(v) for v in r.x], "ilp_obj": float(-ri.fun), "ilp_x": [float(v) for v in ri.x]})
    for (m, n, seed, U) in [(4, 8, 1, 3), (6, 12, 2, 3), (8, 16, 3, 2), (10, 20, 4, 3), (16, 32, 5, 2), (24, 48, 6, 2)]:
        A, b, c, U = synth.dense_ilp(m, n, seed, U)
        r = linprog(-c, A_ub=A, b_ub=b, bounds=(0, U), method="highs-ds")
        ri = milp(-c, constraints=LinearConstraint(A, -np.inf, b), integrality=np.ones(n), bounds=Bounds(0, U))
        out.append({"name": "ilp_%dx%d_s%d" % (m, n, seed), "m": m, "n": n, "seed": seed, "U": U, "lp_obj": float(-r.fun),
                    "ilp_obj": float(-ri.fun)})
    return out


def degenerate_cases():
    """Stalling / cycling LPs: the anti-stalling rules (bound perturbation, Bland fallback) must still land on
    the optimum HiGHS finds."""
    out = []
    for name, (A, b, c) in lpgen.CYCLING.items():
        r = linprog(-np.array(c), A_ub=np.array(A), b_ub=np.array(b), bounds=(0, None), method="highs-ds")
        assert r.status == 0
        out.append({"name": name, "obj": float(-r.fun)})
    for (m, n, seed) in [(30, 40, 1), (100, 150, 3), (300, 120, 6), (150, 150, 7), (200, 300, 4), (250, 400, 8)]:
        A, b, c = lpgen.degenerate_lp(m, n, seed)
        r = linprog(-c, A_ub=A, b_ub=b, bounds=(0, 2), method="highs-ds")
        assert r.status == 0
        out.append({"m": m, "n": n, "seed": seed, "obj": float(-r.fun)})
    return out


def setcover_cases():
    """Minimisation ILPs (set cover): optimum from HiGHS milp."""
    out = []
    for (m, n, seed) in [(15, 25, 1), (30, 40, 2), (40, 60, 3), (50, 80, 4)]:
        A, c = lpgen.setcover_ilp(m, n, seed)
        r = linprog(c, A_ub=-A, b_ub=-np.ones(m), bounds=(0, 1), method="highs-ds")
        ri = milp(c, constraints=LinearConstraint(A, 1, np.inf), integrality=np.ones(n), bounds=Bounds(0, 1))
        out.append({"m": m, "n": n, "seed": seed, "lp_obj": float(r.fun), "ilp_obj": float(ri.fun)})
    return out


def main():
    big = "--big" in sys.argv
    doc = {
        "generator": "tests/golden/make_golden.py",
        "solver": "scipy %s linprog(method='highs-ds') / milp (HiGHS)" % __import__("scipy").__version__,
        "dense": dense_cases(big),
        "general": general_cases(),
        "ilp": ilp_cases(),
        "degenerate": degenerate_cases(),
        "setcover": setcover_cases(),
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print("wrote golden.json: %d dense, %d general, %d ilp, %d degenerate, %d setcover" % (len(doc["dense"]), len(doc["general"]), len(doc["ilp"]),
                                                                                         len(doc["degenerate"]), len(doc["setcover"])))


if __name__ == "__main__":
    main()
