#!/usr/bin/env python3
"""Per basic block of one kernel in a hipcc -S listing: instruction count, scratch (spill) instructions, MFMA, transcendental,
vector-memory and s_waitcnt vmcnt instructions.   isa_blocks.py <file.s> <mangled-name-substring>"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and want in l)
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", 0, 0, 0, 0, 0, []]
for ln in lines[start:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", ln)
    if m:
        blocks.append(cur); cur = [m.group(1), 0, 0, 0, 0, 0, []]
        continue
    t = ln.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur[1] += 1
    if "scratch_" in t: cur[2] += 1
    if "v_mfma" in t: cur[3] += 1
    if re.match(r"(v_exp_f32|v_rcp_f32|v_log_f32)", t): cur[4] += 1
    if re.match(r"(buffer_|global_)", t): cur[5] += 1
    if t.startswith("s_waitcnt") and "vmcnt" in t: cur[6].append(re.search(r"vmcnt\((\d+)\)", t).group(1))
blocks.append(cur)
print("block            insts scratch mfma transc vmem  vmcnt-waits")
for b in blocks:
    if b[1] >= 20 or b[2]:
        print(f"{b[0]:16s} {b[1]:5d} {b[2]:7d} {b[3]:4d} {b[4]:6d} {b[5]:4d}  {','.join(b[6][:24])}")
