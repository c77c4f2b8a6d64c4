#!/usr/bin/env python3
"""Static VALU issue cost of the n-step TD forward kernels (no GPU needed): compiles csrc/dist_ops.hip to gfx950 assembly,
sorts every vector-ALU instruction of a kernel into the four issue classes tests/tools/micro/valu_rate.hip measured
(profiles/r05_valu_rate.txt: cycles a wave64 instruction occupies its SIMD for) and writes the weighted mean to
profiles/r05_td_valu_classes.json -- bench_suite.py prices the hardware's SQ_INSTS_VALU count with it.

    python tests/tools/r05_valu_classify.py"""
import json
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CYCLES = {"e32": 2.4, "vop3": 4.3, "pk": 4.4, "trans": 8.2}       # profiles/r05_valu_rate.txt (at 4 waves per SIMD, 2.4 GHz)
KERNELS = {"dist_nstep_td_fwd": "dist_nstep_fwd_batch_kernelILi16E", "qrdqn_nstep_td_fwd": "qrdqn_fwd_quad_kernelILi8ELi32ELb1E",
           "iqn_nstep_td_fwd": "iqn_fwd_group_kernelILi32E"}


def classify(lines):
    n = {k: 0 for k in CYCLES}
    for l in lines:
        t = l.split()
        if not t or not t[0].startswith("v_"):
            continue
        op = t[0]
        if re.match(r"v_(log|rcp|exp|sqrt|rsq|sin|cos)_", op):
            n["trans"] += 1
        elif op.startswith("v_pk_"):
            n["pk"] += 1
        elif op.endswith("_e32") and "dpp" not in op and "sdwa" not in op:
            n["e32"] += 1
        else:
            n["vop3"] += 1                                          # VOP3 encodings, DPP, v_readlane, 64-bit integer forms
    return n


def main():
    csrc = os.path.join(ROOT, "di-hpc_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "dist_ops.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "include"),
                               "-I" + csrc, "-S", "--cuda-device-only", "-o", asm, os.path.join(csrc, "dist_ops.hip")])
        text = open(asm).read().split("\n")
    out = {"cycles_per_class": CYCLES, "source": "tests/tools/r05_valu_classify.py over hipcc -S of csrc/dist_ops.hip; class costs from "
           "tests/tools/micro/valu_rate.hip (profiles/r05_valu_rate.txt)", "kernels": {}}
    for name, sym in KERNELS.items():
        start = next(i for i, l in enumerate(text) if l.startswith("_Z") and sym in l.split(":")[0])
        end = next(i for i in range(start, len(text)) if "s_endpgm" in text[i])
        n = classify(text[start:end])
        tot = sum(n.values())
        cyc = sum(n[k] * CYCLES[k] for k in n)
        out["kernels"][name] = {"symbol": sym, "static_valu_insts": n, "cycles_per_valu_inst": round(cyc / tot, 3)}
        print(name, n, "mean cycles per VALU instruction", round(cyc / tot, 3))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_td_valu_classes.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
