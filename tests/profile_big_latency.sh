#!/bin/bash
# Memory-system counters of the HBM kernel on config 5 at two in-flight counts (run on the GPU box via gpurun):
# address-translation misses, L2->fabric read/atomic latency (LEVEL/REQ), credit stalls.
set -e
TAG=${1:-r01e}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_latency
mkdir -p $OUT
cd /tmp
for B in 64 512; do
  ARGS="--workload 32x32x32 --no-cpu-baseline --batch $B --steps 1 --warmup 0"
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum -d $OUT/b${B}_utcl -o pmc -- python3 $R/bench.py $ARGS > $OUT/b${B}_utcl.json 2> $OUT/b${B}_utcl.err
  rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_EA0_ATOMIC_sum -d $OUT/b${B}_ea -o pmc -- python3 $R/bench.py $ARGS > $OUT/b${B}_ea.json 2> $OUT/b${B}_ea.err
  rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum -d $OUT/b${B}_lat -o pmc -- python3 $R/bench.py $ARGS > $OUT/b${B}_lat.json 2> $OUT/b${B}_lat.err
  echo "batch $B done"
done
python3 $R/tests/rocpd_counters.py $OUT > $OUT/counters.csv
ls $OUT
