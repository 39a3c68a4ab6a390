#!/bin/bash
# Issue-rate counters of every LDS-resident workload of bench.py, one rocprofv3 --pmc pass + one kernel trace each (run on the GPU box):
#   bash tests/profile_issue_all.sh <tag>
set -e
TAG=${1:-r03b}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp
for WL in ${WORKLOADS:-winograd cyclic 4x4x4_L 4x4x4_P kmethod tril cob}; do
  OUT=$R/gpurun_out/prof_${TAG}_${WL}
  mkdir -p $OUT
  ARGS="--workload $WL --no-cpu-baseline --steps 2 --warmup 0"
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $OUT/pmc_sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
  python3 $R/tests/rocpd_summary.py $OUT $R/gpurun_out/${TAG}_${WL}
  echo "== $WL"; head -4 $R/gpurun_out/${TAG}_${WL}_kernel_stats.csv | tail -2
done
