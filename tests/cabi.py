"""TEST-ONLY ctypes access to the raw C ABI of libhpc_rll_hip.so (include/hpc_rll_hip.h).

The product binds the C ABI from compiled C++ (di-hpc_amd/ext/*.cpp -> hpc_rl_utils.so, hpc_torch_utils_network.so,
hpc_models.so); this module exists so that the tests can call individual entry points directly (expert launch
configurations, argument-error statuses, symbol coverage).  The ctypes prototypes are generated from the C header
itself, so they cannot drift from ``include/hpc_rll_hip.h``; a missing symbol raises at import.

torch is imported first on purpose: torch's bundled ``libamdhip64.so`` has the same soname
(``libamdhip64.so.7``) as the ROCm one the library was linked against, so once torch is loaded the
dynamic linker binds our library to the SAME HIP runtime instance torch uses, and torch's streams /
device pointers are valid inside our launches.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must precede the dlopen below, see docstring)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("HPC_RLL_LIB") or os.path.join(_ROOT, "di-hpc_amd", "hpc_rll", "_lib", "libhpc_rll_hip.so")  # env: kernel A/B tools
HEADER_PATH = os.path.join(_ROOT, "include", "hpc_rll_hip.h")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build it with `python di-hpc_amd/build.py` (hipcc --offload-arch=gfx950). "
        "There is no CPU/eager fallback.")

lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)

_CTYPES = {"int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t,
           "uint64_t": ctypes.c_uint64}


def _parse_header(path):
    """{name: (restype, [argtypes])} for every ``hpc_rll_*`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    sigs = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(hpc_rll_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.replace("const", "").split()[0]])
        sigs[name] = (restype, argtypes)
    return sigs


SIGNATURES = _parse_header(HEADER_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    if os.environ.get("HPC_RLL_LIB") and not hasattr(lib, _name):
        continue                # kernel A/B tools load OLDER builds of the library, which lack the newer entry points
    _fn = getattr(lib, _name)  # AttributeError if the .so is stale: loud by design
    _fn.argtypes = _args
    _fn.restype = _res

ABI_VERSION = 6
if lib.hpc_rll_abi_version() != ABI_VERSION and not os.environ.get("HPC_RLL_LIB"):   # A/B tools load older builds on purpose
    raise ImportError(f"libhpc_rll_hip.so ABI {lib.hpc_rll_abi_version()} != expected {ABI_VERSION}; rebuild")


def check(status: int, what: str = "") -> None:
    """Convert a C-ABI status into the reference's error behaviour (RuntimeError, cf. status.h:19-28)."""
    if status != 0:
        msg = lib.hpc_rll_status_string(status).decode()
        raise RuntimeError(f"{what}: {msg} (status {status})")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None) -> int:
    """The current torch HIP stream for ``device`` as an integer hipStream_t."""
    if _raw_stream is not None and device is not None and device.index is not None:
        return _raw_stream(device.index)          # ~0.3 us instead of ~5 us through torch.cuda.Stream
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def call(name, dev, *args):
    """Launch a C-ABI entry point on torch's current stream of ``dev`` (appended as the last argument).
    The device guard is only entered when ``dev`` is not already current (it costs ~10 us)."""
    fn = getattr(lib, name)
    if dev.index is None or dev.index == torch.cuda.current_device():
        st = fn(*args, stream_ptr(dev))
    else:
        with torch.cuda.device(dev):
            st = fn(*args, stream_ptr(dev))
    if st:
        check(st, name)
