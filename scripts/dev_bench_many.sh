#!/bin/bash
# Same-box comparison of many developer builds (no parity check: use scripts/dev_ab.sh for the ones worth keeping): scripts/dev_bench_many.sh lib1.so lib2.so ...
CFG=${CFG:-c3}; REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
  for so in "$@"; do
    LRT_LIBRARY=$PWD/$so python3 bench.py --config $CFG --steps 4 --warmup 1 --no-cpu-baseline --main-only 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$so rep $rep:', j['ms_per_step'], 'ms', j['value'], 'Msamples/s')"
  done
done
