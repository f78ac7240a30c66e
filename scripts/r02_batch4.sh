#!/bin/bash
# round-2 GPU batch 2: remaining profiles (bio, path, PRB, biovolpath06) and the C4 bench line
mkdir -p gpurun_out/r02/v3
python3 bench.py --config c4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/v3/bench_c4.json 2> gpurun_out/r02/v3/bench_c4.err; tail -c 600 gpurun_out/r02/v3/bench_c4.json; tail -3 gpurun_out/r02/v3/bench_c4.err
for c in c3bio c2 c5 parenchyma; do
  echo "== profile $c"; scripts/r02_profile.sh r02_v3 $c > gpurun_out/r02/v3/prof_$c.txt 2>&1; tail -5 gpurun_out/r02/v3/prof_$c.txt
done
