#!/bin/bash
# usage: tools/kpmc.sh <tag> <lib.so> "<counters>" [bench args]: one rocprofv3 PMC pass of an MSM-only bench with a variant library
set -e
TAG=$1; LIB=$2; CNT=$3; shift 3
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
timeout -s KILL 120 rocprofv3 --pmc $CNT --output-format csv -d $O/pmc -o p -- python3 tools/variant_bench.py $LIB --grown-points --no-cpu-baseline --ipa-log2n 0 --steps 5 --warmup 2 "$@" > $O/bench.log 2>&1
python3 tools/pmc_summary.py $O/pmc
