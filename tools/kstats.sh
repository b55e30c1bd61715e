#!/bin/bash
# usage: tools/kstats.sh <tag> <lib.so> [bench args]: rocprofv3 kernel-trace stats of an MSM-only bench with a variant library
set -e
TAG=$1; LIB=$2; shift 2
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 tools/variant_bench.py $LIB --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 20 --warmup 3 "$@" > $O/bench.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"].split("(")[0].replace("void bpgpu::", "")
    print(f'{n[:40]:40s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
