#!/usr/bin/env python3
"""Build liby3hip.so (gfx950 only) with hipcc.  In-tree output so the .so travels with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["conv_f32.hip", "conv_bf16.hip", "conv_f32x3.hip", "decode.hip", "nms.hip", "elementwise.hip", "preprocess.hip", "y3_api.cpp", "comm.cpp"]
OUT = os.path.join(os.path.dirname(HERE), "lib", "liby3hip.so")
# -ffp-contract=off: epilogue / decode / IoU arithmetic must not be re-associated into FMAs (parity with the
# reference's separate fp32 multiply and add); MFMA accumulation is unaffected.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    deps = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".h", ".hip", ".cpp"))]
    deps.append(os.path.join(HERE, "..", "..", "include", "y3.h"))
    newest = max(os.path.getmtime(d) for d in deps)
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            jobs.append(cmd)
    if jobs:   # the template-heavy conv files take a minute or two each: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(subprocess.check_call, jobs))
    if force or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs, "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
