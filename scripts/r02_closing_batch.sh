#!/bin/bash
# round-2 closing GPU batch: GPU tests, bench lines of every config (with CPU baselines), rocprofv3 profiles of the configs named as arguments
mkdir -p gpurun_out/r02/final
python -u -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/r02/final/tests.log 2>&1; tail -2 gpurun_out/r02/final/tests.log
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r02/final/bench_c3.json 2> gpurun_out/r02/final/bench_c3.err; head -c 100 gpurun_out/r02/final/bench_c3.json; echo
for c in c3bio c2 c4 c5 parenchyma multimesh; do
  python3 bench.py --config $c --steps 5 --warmup 1 > gpurun_out/r02/final/bench_$c.json 2> gpurun_out/r02/final/bench_$c.err; echo "$c done: $(head -c 100 gpurun_out/r02/final/bench_$c.json)"
done
for c in "$@"; do
  echo "== profile $c"; scripts/r02_profile.sh r02_v15 $c > gpurun_out/r02/final/prof_$c.txt 2>&1; tail -5 gpurun_out/r02/final/prof_$c.txt
done
