"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/hpc_rll_hip.h declares (no compute calls: there is no GPU on the CPU test tier)."""
import ctypes
import os
import re

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "hpc_rll_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hpc_rll_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "hpc_rll_gae_forward" in syms and "hpc_rll_gae_backward" in syms


def test_library_exports_every_declared_symbol():
    import cabi as _native
    lib = ctypes.CDLL(_native.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"libhpc_rll_hip.so lacks {missing}"


def test_python_signature_table_matches_header():
    import cabi as _native
    declared = set(declared_symbols())
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert len(declared) >= 30


def test_abi_version_and_status_strings():
    import cabi as _native
    assert _native.lib.hpc_rll_abi_version() == _native.ABI_VERSION
    assert _native.lib.hpc_rll_status_string(0) == b"ok"
    assert b"invalid" in _native.lib.hpc_rll_status_string(-1)


def test_no_oracle_import_in_product():
    """The product package must never import the oracle (parity claims are void otherwise)."""
    pkg = os.path.join(ROOT, "di-hpc_amd")
    bad = []
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "ref_torch" in txt or "gae_ref" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_cpu_tensor_is_rejected_loudly():
    import pytest
    import torch
    import hpc_rl_utils
    v, r, a = torch.zeros(3, 2), torch.zeros(2, 2), torch.zeros(2, 2)
    with pytest.raises(RuntimeError):
        hpc_rl_utils.GaeForward([v, r], [a], 0.99, 0.97)


def test_argument_errors_are_status_codes_not_crashes():
    """Invalid arguments are rejected with a negative status BEFORE any HIP call is made (so this runs without a GPU)."""
    import cabi as _native
    L = _native.lib
    assert L.hpc_rll_gae_forward(None, None, None, None, 4, 4, 0.99, None) == -1            # null pointers
    assert L.hpc_rll_gae_forward(None, None, None, None, -1, 4, 0.99, None) == -1           # negative size
    assert L.hpc_rll_gae_backward(None, None, None, None, 4, -2, 0.99, None) == -1
    assert L.hpc_rll_td_lambda_forward(None, None, None, 3, None, None, None, 4, 4, 0.9, 0.8, 1.0, None) == -1  # bad mode
    assert L.hpc_rll_vtrace_forward(None, None, None, None, None, None, None, None, 4, 4, 0, 0.9, 0.9, 1., 1., 1., 1., None) == -1
    assert L.hpc_rll_lstm_forward(None, None, None, None, None, None, None, None, None, None, None, None,
                                  4, 4, 4, 4096, 1, 0.0, 0, None) == -3                     # H beyond the register path
    assert L.hpc_rll_pad_forward(None, None, None, 4, 8, 1, 1, 0, None) == -1
    assert L.hpc_rll_tune_set(99, 1) == -1
    assert b"invalid argument" in L.hpc_rll_status_string(-1)
    assert L.hpc_rll_partials_floats(100) >= 100
    assert L.hpc_rll_vtrace_workspace_floats(10, 20) >= 6 * 200


def test_tuning_knobs_documented_and_guarded():
    """hpc_rll_tune_set is host-only code with ONE table (csrc/tune.hip, round 5): at most 20 live keys (VERDICT r04 item 7 --
    there were 40), each documented in the header, accepting its shipped default and rejecting an out-of-range value with a
    status (no GPU needed); the keys retired in round 5 and undocumented keys are argument errors."""
    import ctypes
    import re
    import cabi as N
    hdr = open(N.HEADER_PATH).read()
    doc = hdr[hdr.index("Path switches"):hdr.index("int hpc_rll_tune_set")]
    documented = sorted({int(k) for k in re.findall(r"^ \*   key +(\d+) ", doc, flags=re.M)})
    N.lib.hpc_rll_tune_count.restype = ctypes.c_int
    N.lib.hpc_rll_tune_doc.restype = ctypes.c_char_p
    N.lib.hpc_rll_tune_doc.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    n = N.lib.hpc_rll_tune_count()
    live = []
    for i in range(n):
        k = ctypes.c_int(-1)
        text = N.lib.hpc_rll_tune_doc(i, ctypes.byref(k))
        assert text and len(text) > 20
        live.append(k.value)
    assert N.lib.hpc_rll_tune_doc(n, None) is None
    assert sorted(live) == documented and len(live) <= 20, (sorted(live), documented)
    defaults = {3: 1, 8: 1, 16: 1, 17: 1, 18: 0, 21: 1, 22: 0, 24: 0, 25: 1, 26: 9, 27: 10, 28: 1, 29: 2, 31: 3072, 32: 1, 33: 1, 35: 0,
                37: 1, 38: 1, 40: 0}
    assert sorted(defaults) == documented
    for k in live:
        assert N.lib.hpc_rll_tune_set(k, defaults[k]) == 0, k
        assert N.lib.hpc_rll_tune_set(k, -7) != 0, k
    for k in (0, 1, 2, 4, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15, 19, 20, 23, 30, 34, 36, 39, 41, 99):   # retired / never existed
        assert N.lib.hpc_rll_tune_set(k, 0) != 0 and N.lib.hpc_rll_tune_set(k, 1) != 0, k
    assert N.lib.hpc_rll_tune_set(26, 2) != 0 and N.lib.hpc_rll_tune_set(26, 137) == 0 and N.lib.hpc_rll_tune_set(26, 9) == 0   # bit mask
    assert N.lib.hpc_rll_tune_set(21, 2) != 0 and N.lib.hpc_rll_tune_set(17, 2) != 0 and N.lib.hpc_rll_tune_set(37, 2) != 0   # retired values


def test_c_program_links_and_runs(tmp_path):
    """The boundary is a plain C ABI: a C program (no Python, no torch) compiles against include/hpc_rll_hip.h, links
    the shared library and calls the entry points that need no GPU."""
    import shutil
    import subprocess
    import pytest
    LIB = os.path.join(ROOT, "di-hpc_amd", "hpc_rll", "_lib", "libhpc_rll_hip.so")
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    lib_dir = os.path.dirname(LIB)
    exe = str(tmp_path / "abi_smoke")
    rocm_lib = "/opt/rocm/lib"
    cmd = [gcc, "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "abi_smoke.c"),
           "-o", exe, "-L", lib_dir, "-lhpc_rll_hip", f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{rocm_lib}",
           f"-Wl,-rpath-link,{rocm_lib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "abi 6 ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_xcd_relabelling_is_a_permutation():
    """csrc/pad_scatter.hip (scatter backward, tune key 38) and csrc/lstm_block.hpp (row-block backward) re-label workgroups so that
    the ones that share cache lines run on one XCD (workgroup i runs on XCD i % 8).  The formulas, restated: every workgroup keeps
    exactly one logical index, and the workgroups of one group share i % 8."""
    for nx, ny in ((16, 4096), (8, 24), (4, 6), (16, 1)):          # scatter backward: nx channel groups x ny batch elements
        total = nx * ny
        if total % 8:
            continue
        seen = {}
        for i in range(total):
            L = (i % 8) * (total // 8) + i // 8
            b, cg = divmod(L, nx)
            assert (b, cg) not in seen and b < ny
            seen[(b, cg)] = i % 8
        assert len(seen) == total
        if (total // 8) % nx == 0:                                  # a batch element's groups never straddle two XCDs
            for b in range(ny):
                assert len({seen[(b, cg)] for cg in range(nx)}) == 1
    for nnt, nrb in ((8, 32), (8, 8), (4, 16)):                     # row-block backward: nnt tiles x nrb row blocks (nrb % 8 == 0)
        seen = {}
        for i in range(nnt * nrb):
            grp = 8 * nnt
            rbl, nt = (i // grp) * 8 + (i & 7), (i % grp) >> 3
            assert (rbl, nt) not in seen and rbl < nrb and nt < nnt
            seen[(rbl, nt)] = i % 8
        assert all(len({seen[(r, t)] for t in range(nnt)}) == 1 for r in range(nrb))
