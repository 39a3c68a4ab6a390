#!/bin/bash
# A/B of builds of libplinopt_hip.so on config 5, alternating on ONE box (boxes differ by +-5 %):
# usage tests/ab_config5.sh "libA.so libB.so ..." [ncand] [rounds]; prints candidates/s at full load and the time of a candidate alone
LIBS=$1; N=${2:-1024}; R=${3:-3}
for L in $LIBS; do
  PLO_HIP_LIB=$PWD/$L PLO_BIG_SLICES=1 python tests/run_config5.py 1 2>/dev/null | grep "candidates 1 " | sed "s|^|$L alone: |"
done
for r in $(seq 1 $R); do
  for L in $LIBS; do
    PLO_HIP_LIB=$PWD/$L python tests/run_config5.py $N 2>/dev/null | grep "candidates $N" | sed "s|^|$L: |"
  done
done
