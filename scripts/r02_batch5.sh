#!/bin/bash
# round-2 final GPU batch: bench lines of every config (with CPU baseline for C3), profiles of all configs
mkdir -p gpurun_out/r02/final
python3 bench.py --steps 20 --warmup 3 > gpurun_out/r02/final/bench_c3.json 2> gpurun_out/r02/final/bench_c3.err; tail -c 400 gpurun_out/r02/final/bench_c3.json; echo
for c in c3bio c2 c4 c5 parenchyma; do
  python3 bench.py --config $c --steps 5 --warmup 1 > gpurun_out/r02/final/bench_$c.json 2> gpurun_out/r02/final/bench_$c.err; echo "$c done: $(head -c 120 gpurun_out/r02/final/bench_$c.json)"
done
for c in c3 c3bio parenchyma; do
  echo "== profile $c"; scripts/r02_profile.sh r02_v11 $c > gpurun_out/r02/final/prof_$c.txt 2>&1; tail -5 gpurun_out/r02/final/prof_$c.txt
done
