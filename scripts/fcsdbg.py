"""Phase stamps of the chained path's step kernels (lead workgroup), per chain position:
python scripts/fcsdbg.py [m n pivots]   (sets MVX_FCS_DBG=1)"""
import ctypes as C, os, sys
os.environ["MVX_FCS_DBG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import synth
m, n, piv = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 8192, 400)
api = mvolps_amd.api()
A, b, c = synth.dense_lp(m, n, seed=12345)
P = api.create()
P.load_dense(A, b, c)
P.simplex(it_lim=piv)
lib = mvolps_amd.load_library()
lib.mvx_fcs_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
buf = (C.c_ulonglong * (32 * 16))()
rows = lib.mvx_fcs_debug_stamps(buf)
print("k_pc: entry->winner, gather issue, carried state, column carry, ratio+reduce, stores | k_pr: entry->row, L2 issue, divs, row carry+step, price+reduce, stores  (us)")
for g in range(rows):
    a = [buf[g * 16 + k] for k in range(6)]
    r = [buf[g * 16 + 8 + k] for k in range(6)]
    if a[0] == 0:
        continue
    da = [(a[k + 1] - a[k]) / 100.0 for k in range(5)]
    dr = [(r[k + 1] - r[k]) / 100.0 for k in range(5)]
    print("%2d  pc %s = %6.2f | pr %s = %6.2f | pc->pr gap %6.2f" % (g, " ".join("%5.2f" % x for x in da), (a[5] - a[0]) / 100.0,
                                                                  " ".join("%5.2f" % x for x in dr), (r[5] - r[0]) / 100.0, (r[0] - a[5]) / 100.0))
