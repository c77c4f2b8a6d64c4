#!/usr/bin/env python3
"""Reduce the rocprofv3 --pmc databases of tests/tools/r03_td_pmc.sh on the GPU box: per kernel (last launch of each), the
counters, the dispatch duration and the derived VALU issue share = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x duration x clock),
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import collections
import re
import sqlite3
import sys

out = []
for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    rows = con.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration from "
                       "counters_collection where kernel_name like '%dist_nstep_fwd%' or kernel_name like '%qrdqn_fwd%'").fetchall()
    disp = collections.OrderedDict()
    for did, k, grid, wg, c, v, dur in rows:
        m = re.search(r"((?:dist_nstep_fwd|qrdqn_fwd)\w*(?:<[^>]*>)?)", k)
        e = disp.setdefault(did, {"kernel": m.group(1) if m else k[:70], "grid": grid, "wg": wg, "dur_ns": float(dur)})
        e[c] = e.get(c, 0.0) + float(v)
    last = collections.OrderedDict()
    for d in disp.values():
        last[(d["kernel"], d["grid"])] = d
    for (k, grid), d in last.items():
        ctr = {c: v for c, v in d.items() if c not in ("kernel", "grid", "wg", "dur_ns")}
        line = f"{k} grid={grid} wg={d['wg']} duration {d['dur_ns'] / 1e3:.1f} us  " + "  ".join(f"{c}={v:.5g}" for c, v in ctr.items())
        act, iv = ctr.get("GRBM_GUI_ACTIVE"), ctr.get("SQ_INSTS_VALU")
        if act and d["dur_ns"]:
            ghz = act / 8 / d["dur_ns"]
            line += f"  -> clock {ghz:.2f} GHz"
            if iv:
                line += f", VALU issue share {iv * 4 / (1024 * d['dur_ns'] * ghz) * 100:.1f} % ({iv / (grid / d['wg'] * d['wg'] / 64):.0f} VALU instructions per wave)"
        out.append(line)
print("\n".join(out))
