#!/bin/bash
# usage: tools/ab.sh <log2n> lib1 lib2 ...   (MSM-only bench lines per variant)
L=$1; shift
for lib in "$@"; do
  echo "== $lib log2n=$L"
  timeout -k 10 300 python tools/variant_bench.py $lib --log2n $L --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 20 | python -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(round(d['value']/1e6,1), 'Mpts/s', round(d['ms_per_step'],3), 'ms', 'frac', round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['phases_ms'].items()})
" || exit 1
done
