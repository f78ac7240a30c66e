set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03/tests11.log 2>&1 || { tail -30 gpurun_out/r03/tests11.log; exit 1; }
tail -n 2 gpurun_out/r03/tests11.log
timeout -k 10 500 python3 scripts/fuzz_sweep.py 100000 101000 > gpurun_out/r03/fuzz_r1_100000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r1_100000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r1_100000.txt
timeout -k 10 500 python3 scripts/fuzz_sweep.py 100000 100600 r2 > gpurun_out/r03/fuzz_r2_100000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r2_100000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r2_100000.txt
timeout -k 10 300 python3 scripts/fuzz_sweep.py 40000 40150 prbhet > gpurun_out/r03/fuzz_prbhet_40000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_prbhet_40000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_prbhet_40000.txt
