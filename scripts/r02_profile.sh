#!/bin/bash
# rocprofv3 kernel trace + PMC passes of the bench configs named on the command line (tag = r02_<version>_<config>)
V=$1; shift
for c in "$@"; do
  scripts/profile_bench.sh ${V}_$c --config $c --steps 1 --warmup 1 > gpurun_out/prof_${V}_$c.log 2>&1
  tail -40 gpurun_out/prof_${V}_$c/summary.txt
done
