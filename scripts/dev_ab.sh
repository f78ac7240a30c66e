#!/bin/bash
# Same-box A/B of developer builds on C3: scripts/dev_ab.sh libA.so libB.so ...  (paths relative to the repo root)
# For each library: per-lane parity of a small volpath render against the oracle, then bench.py --config c3 (alternating, 2 rounds).
CFG=${CFG:-c3}
for so in "$@"; do
  DEV_SCENE=$CFG LRT_LIBRARY=$PWD/$so python3 scripts/dev_parity.py || echo "PARITY FAILED for $so"
done
for rep in 1 2; do
  for so in "$@"; do
    LRT_LIBRARY=$PWD/$so python3 bench.py --config $CFG --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$so rep $rep:', j['ms_per_step'], 'ms', j['value'], 'Msamples/s  kernel', round(r['avg_launch_ms'],2), 'ms  it/s', round(r['iterations_per_sample'],4), 'rec/s', round(r['records_per_sample'],4))"
  done
done
