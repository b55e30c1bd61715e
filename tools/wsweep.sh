#!/bin/bash
# usage: tools/wsweep.sh <log2n> <c...>   -- MSM-only bench lines per window width
L=$1; shift
for w in "$@"; do
  timeout -k 10 120 python bench.py --log2n $L --window $w --no-cpu-baseline --ipa-log2n 0 --steps 10 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print('n=2^$L c=$w', round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['phases_ms'].items()})
" || exit 1
done
