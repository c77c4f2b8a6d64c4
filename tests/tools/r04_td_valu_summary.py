#!/usr/bin/env python3
"""rocpd databases of tests/tools/r04_td_valu.sh -> profiles-ready JSON: VALU wave-instructions per SAMPLE of the last launch
of each forward (SQ_INSTS_VALU), with duration, clock and VALU issue share of that (instrumented) launch.
Usage: r04_td_valu_summary.py <out.json> <db> [<db> ...]"""
import collections
import json
import sqlite3
import sys

KERNELS = {"dist_nstep_fwd": ("dist_nstep_td_fwd", 1 << 18), "qrdqn_fwd": ("qrdqn_nstep_td_fwd", 1 << 18),
           "iqn_fwd": ("iqn_nstep_td_fwd", 1 << 16)}
out = {"valu_wave_insts_per_sample": {}, "detail": {}, "shape": "B=262144 (IQN 65536) N=64 nstep=5 atoms=51 tau=32",
       "source": "rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE (tests/tools/r04_td_valu.sh), last of three launches"}
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from "
                       "counters_collection").fetchall()
    disp = collections.OrderedDict()
    for did, k, grid, wg, c, v, dur in rows:
        key = next((kk for kk in KERNELS if kk in k), None)
        if key is None:
            continue
        e = disp.setdefault(did, {"key": key, "kernel": k[:160], "grid": grid, "wg": wg, "dur_ns": float(dur)})
        e[c] = e.get(c, 0.0) + float(v)
    last = {}
    for d in disp.values():
        last[d["key"]] = d
    for key, d in last.items():
        name, B = KERNELS[key]
        iv, act = d.get("SQ_INSTS_VALU"), d.get("GRBM_GUI_ACTIVE")
        if not iv:
            continue
        out["valu_wave_insts_per_sample"][name] = iv / B
        det = {"kernel": d["kernel"], "grid": d["grid"], "workgroup": d["wg"], "duration_us": d["dur_ns"] / 1e3, "SQ_INSTS_VALU": iv}
        if act:
            ghz = act / 8 / d["dur_ns"]
            det.update(clock_ghz=ghz, valu_issue_share=iv * 4 / (1024 * d["dur_ns"] * ghz))
        out["detail"][name] = det
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["valu_wave_insts_per_sample"]), json.dumps({k: round(v.get("valu_issue_share", 0), 3) for k, v in out["detail"].items()}))
