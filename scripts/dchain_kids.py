import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from mvolps_amd import capi, synth
from oracle import oracle
fx = json.load(open("tests/golden/config5.json"))
gpu, orc = mvolps_amd.api(), oracle.api()
A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
g, o = synth.load_ilp(gpu, A, b, c, U), synth.load_ilp(orc, A, b, c, U)
for P in (g, o):
    P.simplex()
x = o.col_prim()
frac = [j + 1 for j in range(len(x)) if abs(x[j] - round(x[j])) > 1e-9]
ref = {}
for j in frac[:12]:
    for up in (0, 1):
        k = o.copy()
        if up: orc.set_col_bnds(k.h, j, capi.DB, float(np.ceil(x[j - 1])), 1.0)
        else: orc.set_col_bnds(k.h, j, capi.DB, 0.0, float(np.floor(x[j - 1])))
        k.simplex()
        ref[(j, up)] = (k.it_cnt, k.tableau())
keep = os.environ.get("KEEP") == "1"
held = []
for rep in range(2):
    out = []
    for j in frac[:12]:
        for up in (0, 1):
            k = g.copy()
            if up: gpu.set_col_bnds(k.h, j, capi.DB, float(np.ceil(x[j - 1])), 1.0)
            else: gpu.set_col_bnds(k.h, j, capi.DB, 0.0, float(np.floor(x[j - 1])))
            k.simplex()
            ok = k.it_cnt - g.it_cnt == ref[(j, up)][0] - o.it_cnt and np.array_equal(k.tableau(), ref[(j, up)][1])
            out.append("." if ok else "X")
            if keep: held.append(k)
    print("rep", rep, "".join(out), flush=True)
