#!/usr/bin/env python3
"""Build libhpc_rll_hip.so (the C-ABI HIP library) for gfx950, in-tree.

    python di-hpc_amd/build.py [--force] [--jobs N]

One translation unit per op under csrc/*.hip, compiled with hipcc --offload-arch=gfx950 and
linked into di-hpc_amd/hpc_rll/_lib/libhpc_rll_hip.so.  hipcc cross-compiles without a GPU.
The object files and the .so are git-ignored but travel to the GPU box with the tree.
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build", "obj")
LIBDIR = os.path.join(HERE, "hpc_rll", "_lib")
LIB = os.path.join(LIBDIR, "libhpc_rll_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def newer(src_list, dst):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in src_list)


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "hpc_rll_hip.h"))
    todo = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(o)
        if force or newer([os.path.join(CSRC, s)] + hdrs, o):
            todo.append((os.path.join(CSRC, s), o))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            for src, rc, out in ex.map(cc, todo):
                if verbose or rc:
                    print(f"[hipcc] {os.path.relpath(src, ROOT)} -> rc={rc}")
                if out.strip() and (verbose or rc):
                    print(out)
                if rc:
                    raise RuntimeError(f"hipcc failed on {src}")
    if todo or force or newer(objs, LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[link] {os.path.relpath(LIB, ROOT)}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.jobs)
