#!/bin/bash
set -u
for t in 32:1 32:0 32:1; do HPC_RLL_TUNE=$t timeout 300 python bench_suite.py c3 2>&1 | grep '"op": "ppo"' | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('key $t ppo fwd %.1f us %.3f bwd %.1f us' % (d['fwd_ms']*1e3, d['fwd_frac'], d['bwd_ms']*1e3))"; done
timeout 300 python bench_suite.py td 2>&1 | grep '"op"' | python -c "
import sys,json
seen={}
for l in sys.stdin:
    d=json.loads(l); seen[d['op']]=d
for k,d in seen.items(): print(k, 'fwd %.1f us %.3f' % (d['fwd_ms']*1e3, d['fwd_frac']), 'bwd %.1f us' % (d.get('bwd_ms',0)*1e3))
"
