set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/tests12.log 2>&1 || { tail -30 gpurun_out/r03/tests12.log; exit 1; }
tail -n 2 gpurun_out/r03/tests12.log
for cfg in c3 c3bio multimesh c5; do
    python3 bench.py --config $cfg --steps 4 --warmup 1 --no-cpu-baseline --main-only 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('$cfg:', j['ms_per_step'], 'ms', j['value'], 'Msamples/s frac', r['frac'], 'rec B', r['record_bytes'], r['kernel'])"
done
timeout -k 10 500 python3 scripts/fuzz_sweep.py 110000 110500 > gpurun_out/r03/fuzz_r1_110000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r1_110000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r1_110000.txt
timeout -k 10 500 python3 scripts/fuzz_sweep.py 110000 110800 r2 > gpurun_out/r03/fuzz_r2_110000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r2_110000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r2_110000.txt
