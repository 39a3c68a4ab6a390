#!/bin/bash
# Latency of a config-5 candidate against the number of workgroups in flight (one candidate per workgroup): S workgroups, S candidates.
# With PLO_BIG_WG_PER_CU=1 the grid is capped at one workgroup per CU.  usage: tests/concurrency_config5.sh "1 8 32 64 128 256" [per_cu]
for S in $1; do
  PLO_BIG_WG_PER_CU=${2:-2} PLO_BIG_SLICES=$S PLO_BIG_STATS=1 python tests/run_config5.py $S 2>&1 | grep -a "candidates $S \|phase us" | sed "s|^|S=$S: |" | cut -c1-330
done
