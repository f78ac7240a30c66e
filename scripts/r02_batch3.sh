#!/bin/bash
# round-2 GPU batch: tests, bench lines, VALU attribution (experiment build), C3 profile
scripts/r02_quick.sh v3 || exit 1
echo "== attribution"; scripts/exp_cost_pmc.sh 2>&1 | tee gpurun_out/r02/v3/exp_pmc.txt
echo "== profile c3"; scripts/r02_profile.sh r02_v3 c3 > gpurun_out/r02/v3/prof_c3.txt 2>&1; tail -8 gpurun_out/r02/v3/prof_c3.txt
