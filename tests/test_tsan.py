"""CPU: the branch-and-bound driver's threading under ThreadSanitizer.  mvx_branchAndBound solves each round's
children on a std::async worker while the calling thread replays the next window (clones, bound edits, deletes of
OTHER handles).  Here bnb.cpp is built WITHOUT the HIP engine -- its engine table forwarded to the CPU oracle by
tests/tsan/mvx_on_orc.c -- with -fsanitize=thread, and a 1500-node window-8 run must finish without a report.
(Sanitizers run on the CPU build only; the engine's own shared state -- slab cache, main context -- is guarded by
locks, see engine.cpp.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_window_driver_is_race_free_under_tsan(tmp_path):
    exe = str(tmp_path / "driver_tsan")
    flags = ["-fsanitize=thread", "-O1", "-g", "-fPIE", "-pie"]
    objs = []
    for src, cc, std in (("mvolps_amd/csrc/bnb.cpp", "g++", "-std=c++17"), ("tests/tsan/driver_tsan.cpp", "g++", "-std=c++17"),
                         ("tests/tsan/mvx_on_orc.c", "gcc", "-std=gnu11"), ("oracle/mvolps_oracle.c", "gcc", "-std=gnu11"),
                         ("oracle/mvolps_oracle_bnb.c", "gcc", "-std=gnu11")):
        obj = str(tmp_path / (os.path.basename(src) + ".o"))
        # the oracle is built without OpenMP here (its pragmas are ignored): libgomp is not TSan-instrumented
        subprocess.check_call([cc, std, "-ffp-contract=off", "-Wno-unknown-pragmas"] + flags[:3] + ["-fPIC", "-c", os.path.join(ROOT, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call(["g++", "-fsanitize=thread"] + objs + ["-o", exe, "-lm", "-lpthread"])
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "WARNING: ThreadSanitizer" not in r.stderr
    assert r.stdout.startswith("nodes ") and "incumbent 1" in r.stdout
