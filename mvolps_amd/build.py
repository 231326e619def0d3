"""Builds libmvolps_amd.so for gfx950 with hipcc (in-tree, explicit commands)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmvolps_amd.so")
RCCL_LIB = os.path.join(LIBDIR, "libmvolps_rccl.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["kernels.hip", "engine.cpp", "capi.cpp", "bnb.cpp", "bnb_dist.cpp", "io.cpp"]
BINDIR = os.path.join(HERE, "bin")
CLI = os.path.join(BINDIR, "mvolps")
# -ffp-contract=off: fma() only where written, on host and device alike (bit-exact parity
# with the CPU oracle)
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-math-errno", "-Wall",
         "-Wno-unused-result"]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hdrs += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    objs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(LIBDIR, src + ".o")
        if force or _stale(op, [sp] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(op)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # RCCL transport of the multi-GPU entry (include/mvx_dist.h): its own small library, so that the engine library
    # itself does not depend on librccl
    rccl_src = os.path.join(CSRC, "comm_rccl.cpp")
    if force or _stale(RCCL_LIB, [rccl_src] + hdrs):
        cmd = [HIPCC, "-O2", "-std=c++17", "-fPIC", "-shared", "--offload-arch=gfx950", "-x", "hip", rccl_src, "-o", RCCL_LIB,
               "-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    # command-line front end (the counterpart of 2test.cpp's main)
    os.makedirs(BINDIR, exist_ok=True)
    main_src = os.path.join(CSRC, "mvolps_main.cpp")
    if force or _stale(CLI, [main_src, LIB] + hdrs):
        cmd = [HIPCC, "-O2", "-std=c++17", main_src, "-o", CLI, "-L" + LIBDIR, "-lmvolps_amd", "-Wl,-rpath,$ORIGIN/../lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
