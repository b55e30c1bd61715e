#!/bin/bash
# usage: tools/collect_profiles.sh <gpurun_out subdir written by tools/profile_round.sh>  -> profiles/r01_*
O=gpurun_out/$1
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU | grep -v '^#\|^kernel,' > /tmp/pmc.csv
(echo "kernel,dispatches,counter,mean_per_dispatch"; cat /tmp/pmc.csv) > profiles/r01_pmc_summary.csv
python tools/pmc_summary.py $O/stats_msm | grep -v '^#' > profiles/r01_kernel_stats_msm.csv
python tools/pmc_summary.py $O/stats_full | grep -v '^#' > profiles/r01_kernel_stats_full.csv
grep -h '^{' $O/bench_stats_msm.log > profiles/r01_bench_msm_only.json
grep -h '^{' $O/bench_stats_full.log > profiles/r01_bench_final.json
