#!/usr/bin/env python3
"""Independent pins for whole B&B runs: ILP optima from scipy 1.15.3 HiGHS `milp` (build container only; an independent
solver, as in make_golden.py) -> tests/golden/milp_pins.json.

* the calibrated BASELINE config-5 instance (tests/golden/config5.json: 512x1024, the only full tree the tests close),
  whose optimum had been recorded from the oracle alone;
* two 128x256 ILPs whose FIFO trees close WITH GMI cuts in repaired mode (about 10 000 nodes each on the oracle):
  the cut path end to end against a solver that knows nothing of it.
Run from the repo root:  python tests/golden/make_milp_pins.py"""
import json
import os
import sys
import time

import numpy as np
from scipy.optimize import Bounds, LinearConstraint, milp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from mvolps_amd import synth  # noqa: E402


def solve(m, n, seed, U, cap):
    A, b, c, Ub = synth.dense_ilp(m, n, seed, U, cap)
    t = time.time()
    r = milp(-c, constraints=LinearConstraint(A, -np.inf, b), integrality=np.ones(n), bounds=Bounds(0, Ub if Ub is not None else np.inf))
    assert r.status == 0, r
    return {"m": m, "n": n, "seed": seed, "U": U, "cap": cap, "milp_obj": float(-r.fun), "milp_seconds": round(time.time() - t, 1)}


def main():
    fx = json.load(open(os.path.join(HERE, "config5.json")))
    out = {"generator": "tests/golden/make_milp_pins.py", "solver": "scipy.optimize.milp (HiGHS), scipy 1.15.3",
           "config5": solve(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"]),
           "cut_ilps": [solve(128, 256, 7, 1, 0.01), solve(128, 256, 9, 2, 0.01)]}
    json.dump(out, open(os.path.join(HERE, "milp_pins.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
