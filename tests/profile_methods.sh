#!/bin/bash
# rocprofv3 kernel traces of bin/optimizer's methods on 4x4x4_49_156_L with 10^6 restarts (run on the GPU box via gpurun)
set -e
TAG=${1:-r01k}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_methods
mkdir -p $OUT
make -s -C $R/plinopt_amd/csrc/host
cd /tmp
for m in D G A K E; do
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$m -o trace -- $R/bin/optimizer -q 131071 --only $m -O 1000000 $R/tests/golden/data/4x4x4_49_156_L.sms > $OUT/prog_$m.slp 2> $OUT/log_$m.txt
  grep "Found\|GPU" $OUT/log_$m.txt | head -3
done
ls $OUT
