#!/bin/bash
# Issue-slot utilisation of the HBM kernel on config 5 (run on the GPU box via gpurun): VALU / LDS / VMEM busy
# cycles and LDS bank conflicts, 512 candidates in flight.
set -e
TAG=${1:-r01f}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_util
mkdir -p $OUT
cd /tmp
ARGS="--workload 32x32x32 --no-cpu-baseline --batch 512 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY -d $OUT/p1 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/p2 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD -d $OUT/p3 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p3.json 2> $OUT/p3.err
python3 $R/tests/rocpd_counters.py $OUT > $OUT/counters.csv
cat $OUT/counters.csv
