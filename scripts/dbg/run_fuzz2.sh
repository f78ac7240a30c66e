set -o pipefail
mkdir -p gpurun_out/r03
timeout -k 10 700 python3 scripts/fuzz_sweep.py 120000 122000 > gpurun_out/r03/fuzz_r1_120000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r1_120000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r1_120000.txt
timeout -k 10 700 python3 scripts/fuzz_sweep.py 120000 121500 r2 > gpurun_out/r03/fuzz_r2_120000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_r2_120000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_r2_120000.txt
timeout -k 10 400 python3 scripts/fuzz_shards.py 100000 100060 > gpurun_out/r03/fuzz_shards_100000.txt 2>&1 || { tail -5 gpurun_out/r03/fuzz_shards_100000.txt; exit 1; }
tail -n 1 gpurun_out/r03/fuzz_shards_100000.txt
timeout -k 10 400 python3 bench.py > gpurun_out/r03/bench_default_v6.json 2> gpurun_out/r03/bench_default_v6.err
tail -c 600 gpurun_out/r03/bench_default_v6.json
