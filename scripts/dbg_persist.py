"""Find the first pivot at which the resident-tableau path parts from the oracle on a degenerate LP."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mvolps_amd
from oracle import oracle
from tests import lpgen
gpu, orc = mvolps_amd.api(), oracle.api()
m, n, seed, frac0 = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
A, b, c = lpgen.degenerate_lp(m, n, seed, frac0=frac0)
for mode in (0, 2):
    gpu.set_persist(mode)
    g, o = lpgen.load_degenerate(gpu, A, b, c), lpgen.load_degenerate(orc, A, b, c)
    step = int(sys.argv[5]) if len(sys.argv) > 5 else 16
    k = 0
    while True:
        prev = [x.copy() for x in o.basis()] if o.it_cnt else None
        prev_t = o.tableau() if o.it_cnt else None
        g.simplex(it_lim=step); o.simplex(it_lim=step)
        k += 1
        same = g.it_cnt == o.it_cnt and g.status == o.status and np.array_equal(g.tableau(), o.tableau()) and all(np.array_equal(x, y) for x, y in zip(g.basis(), o.basis()))
        if not same:
            tg, to = g.tableau(), o.tableau()
            d = np.argwhere(tg != to)
            print("mode", mode, "diverged in chunk", k, "it", g.it_cnt, o.it_cnt, "status", g.status, o.status, "ndiff", len(d), "first", d[:5].tolist(),
                  "pert", g.pert_cnt, o.pert_cnt, "bland", g.bland_cnt, o.bland_cnt, flush=True)
            hb, ob = g.basis(), o.basis()
            print("  basis head diff", np.argwhere(hb[0] != ob[0])[:5].tolist(), "nb diff", np.argwhere(hb[1] != ob[1])[:5].tolist(), "flag diff", np.argwhere(hb[2] != ob[2])[:5].tolist())
            if mode:
                import ctypes as C, struct
                cyc = (C.c_ulonglong * 64)()
                gpu.persist_cycles(cyc)
                w = list(cyc)
                f = lambda u: struct.unpack('d', struct.pack('Q', u))[0]
                print('   hdr kind', w[9], 'p', w[10] & 0xffffffff, 'p_up', w[10] >> 32, 'piv', f(w[11]), 'bound', f(w[12]), 'xq', f(w[13]), 'lbq', f(w[14]), 'ubq', f(w[15]), 'delta', f(w[16]), 'newflag', w[20], 'blb', f(w[21]), 'bub', f(w[22]), 'beta_p', f(w[23]))
                print('   r.k1', f(w[26]), 'r.k2', f(w[27]), 'r.idx', w[28], 'r.aux', w[29], 'sdir', w[30] - 10, 'row thread rb.idx/aux', w[31] >> 32, w[31] & 0xffffffff, 'col', f(w[32]), 'beta', f(w[33]), 'blb', f(w[34]), 'bub', f(w[35]), 'recomputed ok/aux', w[36] >> 32, w[36] & 0xffffffff, 't', f(w[37]), '| in loop: col', f(w[38]), 'beta', f(w[39]), 'blb', f(w[40]), 'bub', f(w[41]), 'ok/aux', w[42] >> 32, w[42] & 0xffffffff, 'rb before', w[43] >> 32, w[43] & 0xffffffff) if len(w) > 31 else None
            if prev is not None:
                for name, hb2 in (("gpu", hb), ("orc", ob)):
                    ch_h = np.argwhere(hb2[0] != prev[0]).ravel().tolist(); ch_f = np.argwhere(hb2[2] != prev[2]).ravel().tolist()
                    print("  ", name, "rows changed", ch_h[:6], [(int(prev[0][i]), int(hb2[0][i])) for i in ch_h[:6]], "flags changed", ch_f[:6], [(int(prev[2][j]), int(hb2[2][j])) for j in ch_f[:6]], "nb", [(int(prev[1][j]), int(hb2[1][j])) for j in ch_f[:6]])
                print("   objective row prev (first 8)", prev_t[0][:8].tolist())
            break
        if o.status == 5 or k > 4000:
            print("mode", mode, "same to the end:", o.it_cnt, "pivots, status", o.status, "pert", o.pert_cnt, "bland", o.bland_cnt, flush=True)
            break
