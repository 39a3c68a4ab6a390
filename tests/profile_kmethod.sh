#!/bin/bash
# rocprofv3 kernel trace of bin/optimizer -K with the decompositions on the device (plo::kmethod_kernel) and, for comparison,
# on the host (--host-decomp: plo::cse_chain_batch_kernel), 4x4x4_49_156_L (run on the GPU box via gpurun)
set -e
TAG=${1:-r02g}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${TAG}_kmethod
mkdir -p $OUT
make -s -C $R/plinopt_amd/csrc/host
cd /tmp
( time $R/bin/optimizer -q 131071 --only K -O 1000000 $R/tests/golden/data/4x4x4_49_156_L.sms > $OUT/prog_dev.slp ) 2> $OUT/wall_device.txt
( time $R/bin/optimizer -q 131071 --only K -O 100000 --host-decomp $R/tests/golden/data/4x4x4_49_156_L.sms > $OUT/prog_host.slp ) 2> $OUT/wall_hostdecomp.txt
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $R/bin/optimizer -q 131071 --only K -O 1000000 $R/tests/golden/data/4x4x4_49_156_L.sms > /dev/null 2> $OUT/log_dev.txt
grep -E "GPU \(K\)|Found K|real" $OUT/wall_device.txt $OUT/wall_hostdecomp.txt
python3 $R/tests/rocpd_summary.py $OUT $OUT/summary
head -8 $OUT/summary_kernel_stats.csv
