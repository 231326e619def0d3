"""Phase stamps of k_fcs (lead workgroup), per chain position: MVX_FCS_DBG=1 python scripts/fcsdbg.py [m n pivots]"""
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
buf = (C.c_ulonglong * (34 * 8))()
rows = lib.mvx_fcs_debug_stamps(buf)
names = ["entry", "to winner", "L2 loads", "row phase", "local best", "fresh col", "own step", "ratio/out"]
print("pos  " + " ".join("%11s" % x for x in names) + "   total_us")
for g in list(range(rows - 1)) + [rows - 1]:
    st = [buf[g * 8 + k] for k in range(8)]
    if st[0] == 0:
        continue
    d = [(st[k + 1] - st[k]) / 100.0 if st[k + 1] and st[k] else 0.0 for k in range(7)]
    print("%-4s " % ("boot" if g == rows - 1 else g) + " ".join("%11.2f" % x for x in [0.0] + d) + "   %8.2f" % ((max(st) - st[0]) / 100.0))

print("blocks: kept candidate %d, fresh column %d, no candidate %d" % (buf[33 * 8], buf[33 * 8 + 1], buf[33 * 8 + 2]))
