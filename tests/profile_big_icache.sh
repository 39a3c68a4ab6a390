#!/bin/bash
# Instruction-fetch counters of the HBM kernel on config 5 with one and with two workgroups per CU (run on the GPU box via gpurun)
TAG=${1:-r04}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_icache
mkdir -p $OUT
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_WAIT_INST_ANY\|SQC_INST_[A-Z_]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_ACTIVE_INST[A-Z_]*" | sort -u > $OUT/avail.txt
for S in 256 512; do
  PER=$([ $S = 256 ] && echo 1 || echo 2)
  PLO_BIG_WG_PER_CU=$PER PLO_BIG_SLICES=$S rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES -d $OUT/p_$S -o pmc -- python3 $R/tests/run_config5.py $S > $OUT/p_$S.log 2> $OUT/p_$S.err
  PLO_BIG_WG_PER_CU=$PER PLO_BIG_SLICES=$S rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES -d $OUT/q_$S -o pmc -- python3 $R/tests/run_config5.py $S > $OUT/q_$S.log 2> $OUT/q_$S.err
  PLO_BIG_WG_PER_CU=$PER PLO_BIG_SLICES=$S rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_INSTS_BRANCH -d $OUT/r_$S -o pmc -- python3 $R/tests/run_config5.py $S > $OUT/r_$S.log 2> $OUT/r_$S.err
done
python3 $R/tests/rocpd_counters.py $OUT > $R/gpurun_out/${TAG}_icache_counters.csv 2> $OUT/counters.err
cat $OUT/avail.txt | tr '\n' ' '; echo; cat $R/gpurun_out/${TAG}_icache_counters.csv; grep -h "candidates" $OUT/*.log
