"""Phase stamps of k_chain (lead thread), per chain position, of the LAST chain of the run:
python scripts/chaindbg.py [m n pivots]   (sets MVX_FCS_DBG=1)"""
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
names = ["gather+carry", "ratio", "xchg C", "decide+rows", "row carry", "row step", "price", "xchg R"]
print("step  " + " ".join("%12s" % x for x in names) + "        total (us)")
for g in range(rows):
    a = [buf[g * 16 + k] for k in range(9)]
    if a[0] == 0 or a[8] == 0:
        continue
    da = [(a[k + 1] - a[k]) / 100.0 for k in range(8)]
    print("%4d  %s  %12.2f" % (g, " ".join("%12.2f" % x for x in da), (a[8] - a[0]) / 100.0))
