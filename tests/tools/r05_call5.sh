#!/bin/bash
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/r05_pytest_all.log" 2>&1; tail -6 "$OUT/r05_pytest_all.log"
timeout 600 python bench.py --steps 20 --warmup 5 > "$OUT/r05_bench_a.json" 2> "$OUT/r05_bench_a.err"; tail -c 600 "$OUT/r05_bench_a.err"
python - <<'P'
import json
d=json.loads(open('gpurun_out/r05_bench_a.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'roofline',d['roofline']['frac'])
sd=d['scaling_detail']
for n,v in sd['strong_per_rank_probe'].items(): print(n,{k:round(x['ms_per_step']*1e3,2) for k,x in v.items()})
print(json.dumps(sd['projected_strong_x']))
s=d['suite']
for k in ('scatter_cover','scatter_add','lstm','pad1d_packed_api'):
    if k in s: print(k,{a:(round(b,4) if isinstance(b,float) else b) for a,b in s[k].items() if a in ('fwd_ms','bwd_ms','fwd_frac','bwd_frac')})
P
