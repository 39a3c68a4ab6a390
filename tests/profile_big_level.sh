#!/bin/bash
# Average latency of LDS / vector-memory / scalar-memory instructions of the HBM kernel on config 5
# (SQ_INST_LEVEL_* / SQ_INSTS_*; run on the GPU box via gpurun):  tests/profile_big_level.sh TAG [library]
set -e
TAG=${1:-r03d}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export PLINOPT_HIP_LIB=$2
OUT=$R/gpurun_out/prof_${TAG}_level
mkdir -p $OUT
cd /tmp
ARGS="--workload 32x32x32 --no-cpu-baseline --batch 512 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_BUSY_CU_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -d $OUT/p1 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/p2 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err
python3 $R/tests/rocpd_counters.py $OUT > $OUT/counters.csv
cat $OUT/counters.csv
