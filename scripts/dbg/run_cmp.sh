LRT_LIBRARY=$PWD/scripts/dbg/libliverrt_dev_cmp.so python3 scripts/dev_parity.py || echo "PARITY FAILED"
for rep in 1 2; do
for v in "dev:0" "dev_cmp:0" "dev_cmp:1"; do
  so=scripts/dbg/libliverrt_${v%%:*}.so; w=${v#*:}
  if [ $w = 1 ]; then export LRT_WIDE_RECORDS=1; else unset LRT_WIDE_RECORDS; fi
  LRT_LIBRARY=$PWD/$so python3 bench.py --config c3 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$so wide=$w rep $rep:', j['ms_per_step'], 'ms', j['value'], 'Msamples/s')"
done; done
