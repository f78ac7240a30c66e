#!/bin/bash
# SQ counters of the render kernel for developer builds: scripts/dev_pmc.sh libA.so libB.so ...  -> gpurun_out/r03/pmc_<name>/
export TMPDIR=/tmp
CFG=${CFG:-c3}
for so in "$@"; do
  name=$(basename $so .so)
  O=gpurun_out/r03/pmc_$name
  rm -rf $O; mkdir -p $O
  LRT_LIBRARY=$PWD/$so rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/p1 -- python3 bench.py --config $CFG --no-cpu-baseline --steps 1 --warmup 0 > $O/p1.log 2>&1
  LRT_LIBRARY=$PWD/$so rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/p2 -- python3 bench.py --config $CFG --no-cpu-baseline --steps 1 --warmup 0 > $O/p2.log 2>&1
  python3 - $O $name <<'PY'
import csv, glob, sys, collections
O, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_render" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print(name, {k: "%.4g" % (v / max(1, n[k] and 1)) for k, v in sorted(acc.items())}, "dispatch rows", dict(n))
PY
done
