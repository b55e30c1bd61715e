#!/bin/bash
# usage: [env] tools/ktimeline.sh <tag> [bench args]: kernel start/end times of the last MSM of an MSM-only bench run
set -e
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o r -- python3 bench.py --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 5 --warmup 2 "$@" > $O/bench.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last k_part_count starts the last MSM
idx = max(i for i, r in enumerate(rows) if "k_part_count" in r["Kernel_Name"])
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:]:
    n = r["Kernel_Name"].split("(")[0].replace("void bpgpu::", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f'{n[:28]:28s} q{r.get("Queue_Id","?"):>3s} start {s/1e3:9.1f} us  end {e/1e3:9.1f} us  dur {(e-s)/1e3:8.1f}')
    if "k_reduce2" in n: break
PY
