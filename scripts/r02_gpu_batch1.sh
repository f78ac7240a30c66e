#!/bin/bash
# round 2, GPU batch 1: stream calibration (PMC), bench lines of every config, profile of C3 and C3-bio
set -x
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib_fetch -- scripts/dbg/t_stream_calib > $O/calib_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib_write -- scripts/dbg/t_stream_calib > $O/calib_write.log 2>&1
python3 scripts/calib_summary.py $O/calib_fetch $O/calib_write > $O/calib_summary.txt 2>&1
cat $O/calib_summary.txt
python3 bench.py --config c3 --steps 5 --warmup 1 > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 3000 $O/bench_c3.json
python3 bench.py --config c3bio --steps 5 --warmup 1 > $O/bench_c3bio.json 2> $O/bench_c3bio.err; tail -c 3000 $O/bench_c3bio.json
python3 bench.py --config c2 --steps 5 --warmup 1 > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 3000 $O/bench_c2.json
python3 bench.py --config c5 --steps 3 --warmup 1 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 3000 $O/bench_c5.json
python3 bench.py --config parenchyma --steps 3 --warmup 1 > $O/bench_parenchyma.json 2> $O/bench_parenchyma.err; tail -c 3000 $O/bench_parenchyma.json
tail -5 $O/*.err
