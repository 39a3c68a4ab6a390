#!/bin/bash
# Issue-rate counters (SQ_INSTS_* / SQ_BUSY_CYCLES) of an LDS-resident workload of bench.py (run on the GPU box via gpurun):
#   bash tests/profile_issue.sh <tag> <workload> [steps]
set -e
TAG=${1:-r02h}; WL=${2:-kmethod}; STEPS=${3:-2}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_${WL}
mkdir -p $OUT
cd /tmp
ARGS="--workload $WL --no-cpu-baseline --steps $STEPS --warmup 0"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $OUT/pmc_sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $OUT/pmc_tcc -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_tcc.json 2> $OUT/pmc_tcc.err
python3 $R/tests/rocpd_summary.py $OUT $OUT/summary
cat $OUT/summary_kernel_stats.csv | head -6; grep "plo::" $OUT/summary_pmc.csv
