"""Builds pointrcnn_amd/libpointrcnn_hip.so for gfx950 with hipcc (in-tree, no JIT).

-ffp-contract=off is part of the numerical contract: the reference's CPU sources are
built with plain -O2 (no FMA), and the index-producing kernels must make bit-identical
decisions.  MFMA intrinsics are unaffected by the flag.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpointrcnn_hip.so")
SOURCES = ["c_api.hip", "fps.hip", "fps_bucket.hip", "grouping.hip", "interp.hip", "nms_rotate.hip", "mlp.hip", "sa_chain.hip", "grad.hip", "nn_sorted.hip", "detections.hip", "knn.hip", "rpn_head.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function"]
# per-file extra flags (measured choices)
EXTRA = {}


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    deps = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "fps_sort.h"),
            os.path.join(HERE, "..", "include", "pointrcnn_hip.h")]
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        newest = max(os.path.getmtime(p) for p in [src] + deps)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [hipcc] + FLAGS + EXTRA.get(s, []) + os.environ.get("PRCNN_EXTRA_FLAGS", "").split() + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((s, subprocess.Popen(cmd)))
    for s, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on %s" % s)
    if procs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
