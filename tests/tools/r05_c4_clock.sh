#!/bin/bash
# The shader clock under the C4 LSTM's kernels (exact-fp32 matrix work draws the chip to its power budget): GRBM_GUI_ACTIVE /
# duration per kernel of one bench_suite.py c4 run, rocprofv3 --pmc (no trace domains beside it).
#   gpurun -- 'bash tests/tools/r05_c4_clock.sh'  -> gpurun_out/r05_c4_clock.txt (copy to profiles/)
set -u
REPO=$(pwd)
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4clk
SUITE_PREROLL_S=1 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/c4clk -o pmc -- python "$REPO/bench_suite.py" c4 > "$REPO/gpurun_out/r05_c4_clock.log" 2>&1
python - "$REPO/gpurun_out/r05_c4_clock.txt" $(find /tmp/c4clk -name "*.db") <<'PY'
import collections, sqlite3, sys
out = open(sys.argv[1], "w")
con = sqlite3.connect(sys.argv[2])
rows = con.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection").fetchall()
disp = {}
for did, k, c, v, dur in rows:
    e = disp.setdefault(did, {"k": k.split("(")[0][-48:], "dur": float(dur)})
    e[c] = e.get(c, 0.0) + float(v)
agg = collections.OrderedDict()
for d in disp.values():
    if d["dur"] < 1e6 or "GRBM_GUI_ACTIVE" not in d:
        continue
    a = agg.setdefault(d["k"], [])
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 4 SIMDs x 256 CUs; GRBM_GUI_ACTIVE / 8 = cycles of the launch
    a.append((d["dur"], d["GRBM_GUI_ACTIVE"] / 8 / d["dur"], d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / max(d["GRBM_GUI_ACTIVE"] / 8 * 1024, 1.0)))
print("# kernel: launches, mean duration ms, shader clock GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration) mean [min, max], 157.3 TF x clock / 2.4,", file=out)
print("#         SQ_VALU_MFMA_BUSY_CYCLES / (launch cycles x 1024 SIMDs) -- uncalibrated: read it as a ratio between kernels", file=out)
for k, a in agg.items():
    ghz = [x[1] for x in a]
    line = "%-50s n=%3d  %8.3f ms  %.3f GHz [%.3f, %.3f]  fp32 matrix peak at that clock %.1f TF  MFMA busy %.3f" % (
        k, len(a), sum(x[0] for x in a) / len(a) / 1e6, sum(ghz) / len(ghz), min(ghz), max(ghz), 157.3 * sum(ghz) / len(ghz) / 2.4,
        sum(x[2] for x in a) / len(a))
    print(line, file=out)
    print(line)
PY
