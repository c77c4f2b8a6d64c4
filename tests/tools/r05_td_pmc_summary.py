#!/usr/bin/env python3
"""rocpd databases of tests/tools/r05_td_pmc.sh -> one JSON: every counter of the LAST launch of each n-step TD forward kernel
(summed over its dimension instances), its duration and effective clock.  Usage: r05_td_pmc_summary.py <out.json> <db> ..."""
import collections
import json
import sqlite3
import sys

KEYS = ("dist_nstep_fwd", "qrdqn_fwd", "iqn_fwd")
out = {}
for db in sys.argv[2:]:
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from "
                       "counters_collection").fetchall()
    disp = collections.OrderedDict()
    for did, k, grid, wg, c, v, dur in rows:
        key = next((kk for kk in KEYS if kk in k), None)
        if key is None:
            continue
        e = disp.setdefault(did, {"key": key, "kernel": k.split("(")[0][-60:], "grid": grid, "wg": wg, "dur_ns": float(dur), "c": {}})
        e["c"][c] = e["c"].get(c, 0.0) + float(v)
    last = {}
    for d in disp.values():
        last[d["key"]] = d
    for key, d in last.items():
        o = out.setdefault(key, {"kernel": d["kernel"], "grid": d["grid"], "workgroup": d["wg"], "duration_us": [], "counters": {}})
        o["duration_us"].append(round(d["dur_ns"] / 1e3, 2))
        for c, v in d["c"].items():
            if c == "GRBM_GUI_ACTIVE":
                o.setdefault("clock_ghz", []).append(round(v / 8 / d["dur_ns"], 3))
            else:
                o["counters"][c] = v
json.dump(out, open(sys.argv[1], "w"), indent=1)
for key, o in out.items():
    c = o["counters"]
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    print(key, o["kernel"], "us", o["duration_us"], "GHz", o.get("clock_ghz"))
    print("   share of wave cycles:", {k: round(c[k] / wc, 3) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
                                                                       "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                                                                       "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC") if k in c})
    print("   counts:", {k: int(v) for k, v in c.items() if k.startswith("SQ_INSTS") or k in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_LDS_BANK_CONFLICT",
                                                                                         "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_THREAD_CYCLES_VALU",
                                                                                         "SQ_INST_CYCLES_SALU")})
