#!/bin/bash
# usage: tools/msm_lines.sh <log2n> [repeats]: MSM-only bench lines (value, ms per step, phases)
L=${1:-20}; R=${2:-1}
for i in $(seq $R); do
timeout -k 10 200 python bench.py --log2n $L --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 20 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', 'frac', round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['phases_ms'].items()})
" || exit 1
done
