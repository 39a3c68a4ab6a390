#!/bin/bash
# LDS / issue counters of the HBM kernel on config 5 for one build of the library (run on the GPU box via gpurun):
#   tests/profile_big_lds.sh TAG [path of libplinopt_hip.so]
set -e
TAG=${1:-r03d}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export PLINOPT_HIP_LIB=$2
OUT=$R/gpurun_out/prof_${TAG}_lds
mkdir -p $OUT
cd /tmp
ARGS="--workload 32x32x32 --no-cpu-baseline --batch 512 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY -d $OUT/p1 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p1.json 2> $OUT/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU -d $OUT/p2 -o pmc -- python3 $R/bench.py $ARGS > $OUT/p2.json 2> $OUT/p2.err
python3 $R/tests/rocpd_counters.py $OUT > $OUT/counters.csv
cat $OUT/counters.csv
grep -o '"value": [0-9.]*' $OUT/p1.json | head -1
