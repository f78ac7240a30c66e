#!/bin/bash
# Profiles bench.py with rocprofv3: (1) kernel trace + stats, (2..) PMC passes (separate runs, no tracing).
# Usage: scripts/profile_bench.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --main-only "$@" > $OUT/bench_trace.log 2>&1 || true
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
            "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SMEM" \
            "FETCH_SIZE" "WRITE_SIZE TCC_EA0_ATOMIC_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- python3 bench.py --main-only "$@" > $OUT/bench_pmc_$name.log 2>&1 || echo "pmc pass failed: $pass" >> $OUT/errors.log
done
python3 scripts/summarize_prof.py $OUT "$@" > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
