#!/bin/bash
# Round-3 profile batch: scripts/r03_profile.sh <tag> [configs...]  -> gpurun_out/prof_r03_<tag>_<config>/ (summary.txt, traffic.json, kernel stats)
TAG=$1; shift
CFGS=${@:-c3 c3hg c2 c5 multimesh parenchyma c3bio het mis}
for c in $CFGS; do
  echo "== $c"
  scripts/profile_bench.sh r03_${TAG}_$c --config $c > gpurun_out/prof_r03_${TAG}_$c.log 2>&1
  grep -E "derived|VALU|lane util|HBM traffic|wave cycles" gpurun_out/prof_r03_${TAG}_$c.log | head -8
  rm -rf gpurun_out/prof_r03_${TAG}_$c/pmc_*/*/*.db gpurun_out/prof_r03_${TAG}_$c/trace/*/*.db 2>/dev/null
done
