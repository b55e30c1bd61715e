#!/bin/bash
# usage: tools/wsweep2.sh <log2n> c1 c2 ...: MSM bench against the window width
L=$1; shift
for c in "$@"; do
echo "== log2n=$L c=$c"
timeout -k 10 200 python bench.py --log2n $L --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 10 --window $c | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', {k: round(v,3) for k,v in d['phases_ms'].items()})
" || exit 1
done
