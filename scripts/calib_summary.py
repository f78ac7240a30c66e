"""Summarises the FETCH_SIZE / WRITE_SIZE calibration run of scripts/dbg/t_stream_calib (known bytes per launch: 88 B x
67 108 864 records).  Usage: python scripts/calib_summary.py gpurun_out/calib_fetch gpurun_out/calib_write"""
import csv, glob, os, sys, collections
KNOWN = 88.0 * (64 << 20)
for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            agg[(row["Kernel_Name"].split("(")[0], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (k, c), v in sorted(agg.items()):
        if "calib" not in k: continue
        v = v[-3:] if len(v) > 3 else v                       # the last three dispatches are the measured repetitions
        kib = sum(v) / len(v)
        expect = {"calib_load": (KNOWN, 0), "calib_store": (0, KNOWN), "calib_copy": (KNOWN, KNOWN)}[[n for n in ("calib_load", "calib_store", "calib_copy") if n in k][0]]
        known = expect[0] if c == "FETCH_SIZE" else expect[1]
        line = f"{k:24s} {c:11s} {kib * 1024 / 1e9:9.3f} GB per launch (mean of {len(v)})   algorithmic {known / 1e9:7.3f} GB"
        if known: line += f"   known / counter = {known / (kib * 1024):.4f}"
        print(line)
