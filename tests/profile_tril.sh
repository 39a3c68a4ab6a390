#!/bin/bash
# rocprofv3 passes for the trilplacer kernel (run on the GPU box via gpurun)
set -e
TAG=${1:-r01h}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_tril
mkdir -p $OUT
cd /tmp
ARGS="--workload tril --no-cpu-baseline --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $R/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES -d $OUT/pmc_sq -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $OUT/pmc_tcc -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_tcc.json 2> $OUT/pmc_tcc.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- python3 $R/bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err
ls $OUT
