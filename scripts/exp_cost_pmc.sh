#!/bin/bash
# VALU instruction attribution: rocprofv3 counters for the experiment build (see scripts/exp_cost.py) per configuration.
export TMPDIR=/tmp
export LRT_LIBRARY=$PWD/scripts/dbg/libliverrt_exp.so
OUT=gpurun_out/exp_pmc; mkdir -p $OUT
for cfg in ${CFGS:-plain:0 free2x:256 query2x:512 surface2x:1024 fresh2x:2048 machinery:4096}; do
  name=${cfg%%:*}; export LRT_EXP=${cfg##*:}
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/$name -- python3 bench.py --no-cpu-baseline --spp 128 --steps 1 --warmup 0 > $OUT/$name.log 2>&1
  python3 - $OUT/$name $name <<'PY'
import sys, glob, csv, collections
tot = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]: tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(sys.argv[2], " ".join(f"{k}={v:.4g}" for k, v in sorted(tot.items())), flush=True)
PY
done
