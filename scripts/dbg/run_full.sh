set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/tests10.log 2>&1 || { tail -30 gpurun_out/r03/tests10.log; exit 1; }
tail -2 gpurun_out/r03/tests10.log
for cfg in c3 c3bio multimesh parenchyma; do
  for w in 0 1; do
    if [ $w = 1 ]; then export LRT_WIDE_RECORDS=1; else unset LRT_WIDE_RECORDS; fi
    python3 bench.py --config $cfg --steps 5 --warmup 1 --no-cpu-baseline --main-only 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$cfg wide=$w:', j['ms_per_step'], 'ms', j['value'], 'Msamples/s frac', r['frac'], 'rec B', r['record_bytes'], r['kernel'])"
  done
done
