#!/bin/bash
# usage: tools/kstats_ipa.sh <tag> [log2n]: rocprofv3 kernel-trace stats of two device-resident IPA sessions (tools/ipa_rounds.py)
set -e
TAG=$1; K=${2:-20}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o r -- python3 tools/ipa_rounds.py $K > $O/run.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void bpgpu::", "")
    print(f'{n[:40]:40s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.2f} min_us {float(r["MinNs"])/1e3:9.2f} max_us {float(r["MaxNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
