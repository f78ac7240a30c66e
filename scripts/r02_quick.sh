#!/bin/bash
# quick GPU check: full GPU test suite + bench lines of the main configs (no CPU baseline); args: tag
TAG=${1:-q}
O=gpurun_out/r02/$TAG
mkdir -p $O
python -u -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for c in c3 c3bio c2 c5 parenchyma multimesh; do
  python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python3 -c "
import json,sys
try:
    j=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1]); r=j['roofline']
    print('$c', j['value'], 'Msamples/s', j['ms_per_step'], 'ms  frac', r['frac'], 'it/s', round(r['iterations_per_sample'],3), 'rec/s', round(r['records_per_sample'],3))
except Exception as e: print('$c FAILED', e); print(open('$O/bench_$c.err').read()[-800:])
"
done
