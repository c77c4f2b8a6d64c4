#!/usr/bin/env python3
"""Build the native code of the package for gfx950, in-tree.

    python di-hpc_amd/build.py [--force] [--jobs N]

1. libhpc_rll_hip.so -- the C-ABI HIP library: one translation unit per op under csrc/*.hip, compiled with
   hipcc --offload-arch=gfx950 and linked into di-hpc_amd/hpc_rll/_lib/.  hipcc cross-compiles without a GPU.
2. hpc_rl_utils.so, hpc_torch_utils_network.so, hpc_models.so -- the PyTorch-ROCm extension modules (the reference's
   three pybind modules, setup.py:19-49): HOST-ONLY C++ under ext/*.cpp compiled with g++ against torch's headers and
   linked to libtorch + libhpc_rll_hip.so (rpath $ORIGIN/hpc_rll/_lib).  No device code, no hipify.
The object files and the .so files are git-ignored but travel to the GPU box with the tree.
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


EXT = os.path.join(HERE, "ext")
EXT_OBJ = os.path.join(HERE, "build", "ext")
CXX = os.environ.get("CXX", "g++")
# module name -> translation units under ext/
EXT_MODULES = {
    "hpc_rl_utils": ["rl_utils.cpp", "rl_utils_lists.cpp", "rl_utils_pad.cpp"],
    "hpc_torch_utils_network": ["network.cpp"],
    "hpc_models": ["models.cpp"],
}


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
    build_ext(force=force, jobs=jobs, verbose=verbose)
    return LIB


def _ext_flags():
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    flags = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
             "-Wno-unknown-pragmas", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
             "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for name in ("COMPILER_TYPE", "STDLIB", "BUILD_ABI"):   # the pybind11 ABI tags torch itself was built with
        v = getattr(torch._C, "_PYBIND11_" + name, None)
        if v is not None:
            flags.append('-DPYBIND11_%s="%s"' % (name, v))
    incs = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], os.path.join(ROOT, "include"), EXT]
    flags += ["-isystem" + i for i in incs[:-2]] + ["-I" + i for i in incs[-2:]]
    tlib = ce.library_paths()[0]
    link = ["-shared", "-L" + tlib, "-L" + LIBDIR, "-lhpc_rll_hip", "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10",
            "-lc10_hip", "-ltorch_hip", "-Wl,-rpath,$ORIGIN/hpc_rll/_lib", "-Wl,-rpath," + tlib, "-Wl,--no-as-needed"]
    return flags, link


def build_ext(force=False, jobs=None, verbose=True):
    """g++ the host-only torch extension modules (ext/*.cpp) into di-hpc_amd/<module>.so."""
    os.makedirs(EXT_OBJ, exist_ok=True)
    flags, link = _ext_flags()
    hdrs = [os.path.join(EXT, f) for f in os.listdir(EXT) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "hpc_rll_hip.h"))
    todo, mods = [], {}
    for mod, units in EXT_MODULES.items():
        objs = []
        for u in units:
            src, obj = os.path.join(EXT, u), os.path.join(EXT_OBJ, u[:-4] + ".o")
            objs.append(obj)
            if force or newer([src] + hdrs, obj):
                todo.append((src, obj, mod))
        mods[mod] = objs

    def cc(job):
        src, obj, mod = job
        cmd = [CXX] + flags + ["-DTORCH_EXTENSION_NAME=" + mod, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
            for src, rc, out in ex.map(cc, todo):
                if verbose or rc:
                    print(f"[g++] {os.path.relpath(src, ROOT)} -> rc={rc}")
                if out.strip() and (verbose or rc):
                    print(out)
                if rc:
                    raise RuntimeError(f"g++ failed on {src}")
    for mod, objs in mods.items():
        so = os.path.join(HERE, mod + ".so")
        if force or newer(objs + [LIB], so):
            r = subprocess.run([CXX, "-o", so] + objs + link, capture_output=True, text=True)
            if r.returncode:
                print(r.stdout + r.stderr)
                raise RuntimeError(f"link failed for {mod}")
            if verbose:
                print(f"[link] {os.path.relpath(so, ROOT)}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.jobs)
