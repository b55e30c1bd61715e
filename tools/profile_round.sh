#!/bin/bash
# Round profile: kernel-trace stats of the bench commands, then PMC passes (each its own run).
# usage: tools/profile_round.sh <outdir under gpurun_out>
set -e
O=gpurun_out/$1
mkdir -p $O
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
# MSM-only command: every k_accum launch is the 2^20 bench MSM, so the stats average is the roofline's kernel time
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msm -o r -- python3 bench.py --steps 20 --warmup 3 --ipa-log2n 0 --second-log2n 0 --no-next-rows > $O/bench_stats_msm.log 2>&1
echo "stats msm done"
# the default command (adds the 2^24 leg, the fold-kernel, next-row and IPA prove / verify legs: k_accum launches of many sizes)
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_full -o r -- python3 bench.py --steps 20 --warmup 3 > $O/bench_stats_full.log 2>&1
echo "stats full done"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_msm24 -o r -- python3 bench.py --log2n 24 --steps 5 --warmup 2 --ipa-log2n 0 --no-cpu-baseline --second-log2n 0 --no-next-rows > $O/bench_stats_msm24.log 2>&1
echo "stats 2^24 done"
# PMC passes of the MSM-only bench (k_accum, sort kernels, reduce, scalar fold, inner product)
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  timeout -s KILL 200 rocprofv3 --pmc $pass --output-format csv -d $O/pmc_$name -o p -- python3 bench.py --no-cpu-baseline --ipa-log2n 0 --second-log2n 0 --no-next-rows --steps 5 --warmup 2 > $O/pmc_$name.log 2>&1
  echo "pass $name done"
done
# PMC pass of the fold-kernel leg (k_fold_joint4 / k_fold_monic at the sizes of a 2^20 proof)
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fold -o p -- python3 bench.py --no-cpu-baseline --second-log2n 0 --steps 2 --warmup 1 > $O/pmc_fold.log 2>&1
echo "pass fold done"
find $O -name "*.csv" | head -30
