#!/usr/bin/env python3
"""Registers / LDS / spills of the compiled kernels of one translation unit (CPU side, no GPU):
    tests/tools/kernel_regs.py categorical [name-filter]
reads di-hpc_amd/build/obj/<unit>.o (the hipcc fat object), unbundles the gfx950 code object and prints the
.vgpr_count / .agpr_count / .sgpr_count / LDS / spill counts of its kernels from the ELF notes."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
unit = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
obj = unit if unit.endswith(".o") else os.path.join(ROOT, "di-hpc_amd", "build", "obj", unit + ".o")
with tempfile.TemporaryDirectory() as d:
    subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", os.path.abspath(obj)], cwd=d, capture_output=True)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    if not co:   # objdump writes next to the input
        base = os.path.dirname(os.path.abspath(obj))
        co = [os.path.join(base, f) for f in os.listdir(base) if f.startswith(os.path.basename(obj) + ".") and "gfx950" in f]
        cleanup = [os.path.join(base, f) for f in os.listdir(base) if f.startswith(os.path.basename(obj) + ".0.")]
    else:
        co = [os.path.join(d, co[0])]
        cleanup = []
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co[0]], capture_output=True, text=True).stdout
    for f in cleanup:
        os.remove(f)
cur = {}
rows = []
for ln in notes.splitlines():
    m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", ln)
    if not m:
        continue
    k, v = m.group(1), m.group(2).strip()
    if k == "agpr_count" and cur.get("name"):
        rows.append(cur); cur = {}
    cur[k] = v
if cur.get("name"):
    rows.append(cur)
seen = set()
for r in rows:
    n = r.get("name", "")
    if n in seen or flt not in n:
        continue
    seen.add(n)
    dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"hpc_rll::\(anonymous namespace\)::|hpc_rll::|void ", "", dem).split("(")[0]
    print(f"{dem[:70]:70s} vgpr {r.get('vgpr_count','?'):>4s} agpr {r.get('agpr_count','?'):>4s} sgpr {r.get('sgpr_count','?'):>4s} "
          f"lds {r.get('group_segment_fixed_size','?'):>6s} spill v{r.get('vgpr_spill_count','?')} s{r.get('sgpr_spill_count','?')} scratch {r.get('private_segment_fixed_size','?')}")
