#!/bin/bash
# usage: tools/ktimeline_ipa.sh <tag> [log2n]: kernel start/end times of one short prover round (tools/ipa_rounds.py)
set -e
TAG=$1; K=${2:-14}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o r -- python3 tools/ipa_rounds.py $K > $O/run.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_part_count" in r["Kernel_Name"]]
i0 = starts[-4]
# back up to the round's first kernel (the scalars of the L, R operands)
while i0 > 0 and "k_fold_scalars" not in rows[i0 - 1]["Kernel_Name"] and "k_frozen_update" not in rows[i0 - 1]["Kernel_Name"]: i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:starts[-3]]:
    n = r["Kernel_Name"].split("(")[0].replace("void bpgpu::", "")
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f'{n[:30]:30s} start {s/1e3:8.1f} us  end {e/1e3:8.1f} us  dur {(e-s)/1e3:7.1f}')
PY
