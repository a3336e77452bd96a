# -*- coding: utf-8 -*-
"""Build libdmn_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m dms_amd.build            # incremental
    python -m dms_amd.build --force

The library is built IN-TREE (dms_amd/lib/libdmn_hip.so) so that it travels to the GPU
box with the source snapshot; it is git-ignored.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdmn_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["abi.hip", "sru.hip", "sru_bwd.hip", "sru_fwd_k2.hip", "sru_fwd_k3.hip", "sru_fwd_k4.hip", "dynfilter.hip", "mix.hip", "mix_sru.hip", "glue.hip", "wino.hip", "tapconv.hip", "metrics.hip", "dpn_glue.hip", "gconv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# sru_bwd.hip: its register ring is loaded by inline asm that hipcc does not count.  With SLP
# vectorisation on, hipcc packs the step's scalar maths into v_pk_* pairs and MOVES ring registers
# into pair positions before their counted wait (copies of registers whose loads are still in
# flight; tools/check_asm_waits.py shows them).  Scalar code keeps every slot in its own register.
EXTRA_FLAGS = {"sru_bwd.hip": ["-fno-slp-vectorize"]}


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    mt = os.path.getmtime(out)
    return any(os.path.getmtime(d) > mt for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "sru_fwd.inl"), os.path.join(CSRC, "sru_ring.h"),
            os.path.join(os.path.dirname(HERE), "include", "dmn_hip.h")]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs, procs = [], []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if force or procs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
