#!/bin/bash
O=gpurun_out/r02/exp768; mkdir -p $O
python -u -m pytest tests/test_bio_gpu.py -q --timeout 300 -p no:cacheprovider -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for v in "" "LRT_BLOCK768=1"; do
 for c in c3 c3bio parenchyma; do
  env $v python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/b.json 2> $O/b.err
  python3 -c "
import json
try:
    j=json.loads(open('$O/b.json').read().strip().splitlines()[-1]); r=j['roofline']; print('$v $c', j['value'], 'Msamples/s', j['ms_per_step'], 'ms it/s', round(r['iterations_per_sample'],3))
except Exception as e: print('$v $c FAILED', e); print(open('$O/b.err').read()[-500:])
"
 done
done
