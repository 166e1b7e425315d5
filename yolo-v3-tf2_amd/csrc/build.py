#!/usr/bin/env python3
"""Build liby3hip.so (gfx950 only) with hipcc.  In-tree output so the .so travels with the repo snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["conv_f32.hip", "conv_head.hip", "conv_res_f32.hip", "conv_stem.hip", "conv_bf16.hip", "conv_res_bf16.hip", "conv_f32x3.hip", "decode.hip", "nms.hip", "elementwise.hip", "preprocess.hip", "y3_api.cpp", "comm.cpp"]
OUT = os.path.join(os.path.dirname(HERE), "lib", "liby3hip.so")
# -ffp-contract=off: epilogue / decode / IoU arithmetic must not be re-associated into FMAs (parity with the
# reference's separate fp32 multiply and add); MFMA accumulation is unaffected.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def build(force=False, verbose=False, defines=(), out=None, only=None):
    """defines / out / only: build a variant of the library (extra -D flags, other output name, subset of sources
    recompiled with the flags; the other objects are shared with the default build) for tools/ab_libs.py."""
    if defines or out or os.environ.get("Y3_VARIANT_FLAGS"):
        return _build_variant(list(defines), out or OUT.replace(".so", "_var.so"), only or SOURCES, verbose)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # an object is stale when its own source or any header is newer (no source includes another source)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")] + [os.path.join(HERE, "..", "..", "include", "y3.h")]
    newest_header = max(os.path.getmtime(d) for d in headers)
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_header, os.path.getmtime(os.path.join(HERE, src))):
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


def _build_variant(defines, out, only, verbose):
    # The default objects of the OTHER sources are linked as they are; nothing of the default build is recompiled here.  (Until visit 4af this
    # called build() first -- with a source edited in place for the variant, that rebuilt the DEFAULT library from the edited source as well, and
    # an A/B of "default vs variant" compared a build with itself: profiles/r04_ab_bf16_shortcut.txt, r04_ab_f32_nofetch1x1.txt were re-measured.)
    objdir = os.path.join(HERE, "build")
    missing = [s for s in SOURCES if s not in only and not os.path.exists(os.path.join(objdir, s + ".o"))]
    if missing:
        raise SystemExit(f"build the default library first (python build.py): no object for {missing}")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("Y3_VARIANT_FLAGS", "").split()    # tools: extra compiler flags for the variant's sources (e.g. "-mllvm -amdgpu-sched-strategy=max-ilp")
    tag = "_".join(d.replace("=", "") for d in defines) or ("flags" if extra else "var")
    objs = []
    for src in SOURCES:
        if src in only:
            obj = os.path.join(objdir, f"{src}.{tag}.o")
            cmd = [hipcc, *FLAGS, *extra, *[f"-D{d}" for d in defines], "-x", "hip", "-c", os.path.join(HERE, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        else:
            obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, "-ldl"])
    return out


if __name__ == "__main__":
    if "--variant" in sys.argv:                # build.py --variant OUT.so SRC.hip -DNAME[=V] ...
        i = sys.argv.index("--variant")
        out_, src_ = sys.argv[i + 1], sys.argv[i + 2]
        print(build(defines=[a[2:] for a in sys.argv[i + 3:] if a.startswith("-D")], out=os.path.abspath(out_), only=[src_], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
