"""What the batched rounds of a B&B run do (g_round_hist in kernels.hip): chain lengths k_dsel reaches, what k_select
is left with.  usage: roundstats.py [config5|wide]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mvolps_amd
from mvolps_amd import bnb, synth
api = mvolps_amd.api()
lib = mvolps_amd.load_library()
which = sys.argv[1] if len(sys.argv) > 1 else "config5"
if which == "config5":
    fx = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5.json")))
    A, b, c, U = synth.dense_ilp(fx["m"], fx["n"], fx["seed"], fx["U"], fx["cap"])
    kw = dict(quirks=0, window=64)
else:
    A, b, c, U = synth.dense_ilp(512, 1024, 12345, 3, 0.4)
    kw = dict(quirks=0, window=64, max_nodes=2000)
bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), quirks=0, max_nodes=100, window=64)
out = (C.c_ulonglong * 18)()
fn = lib.mvx_debug_round_hist
lib.mvx_debug_stats(1)
fn(out, 1)
cyc = (C.c_ulonglong * 8)()
lib.mvx_debug_dsel_cycles(cyc, 1)
t = time.perf_counter()
r = bnb.branch_and_bound(synth.load_ilp(api, A, b, c, U), **kw)
dt = time.perf_counter() - t
fn(out, 0)
h = list(out)
print("nodes", r["count"], "pivots", r["total_pivots"], "secs %.3f" % dt)
print("k_dsel: not applicable", h[0], " chains by length", h[1:9], " applicable but no plain dual pivot", h[9])
print("k_select: passthrough", h[10], " idle slot", h[11], " primal step", h[12], " dual/start step", h[13], " other phase", h[14], " solve ended", h[15])
lib.mvx_debug_dsel_cycles(cyc, 0)
cy = list(cyc)
steps, launches = max(1, cy[6]), max(1, cy[7])
print("k_dsel shader-clock cycles (s_memtime on thread 0; ~2.3 cycles per ns, a barrier charges the slowest wave to the part that follows): entry %.0f per launch; per step: leaving row %.0f, row p + ratio %.0f, column q %.0f, bookkeeping %.0f; exit %.0f per launch; steps %d launches %d"
      % (cy[0] / launches, cy[1] / steps, cy[2] / steps, cy[3] / steps, cy[4] / steps, cy[5] / launches, cy[6], cy[7]))
