"""Summarise a rocprofv3 output directory produced by scripts/profile_bench.sh.
Prints the kernel-trace stats table, the PMC counters of this library's kernels (per dispatch), derived figures for the
render kernel, and writes <out>/traffic.json (HBM bytes per launch of the render kernel; both counters are in KiB).
FETCH_SIZE is doubled (MI355X_MICROARCH.md "HBM": on gfx950 it reports 1/2 of the bytes of wide coalesced stream loads), which
profiles/r02_stream_calibration.txt reproduces on this kernel's own record streams with a known byte count (x2.000 / x1.980);
WRITE_SIZE is read as is (x1.000 in the same calibration)."""
FETCH_FACTOR = 2.0
import csv, glob, json, os, sys, collections
out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))
for f in find("*kernel_stats.csv"):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 10: print("  ", ", ".join(row))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in find("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "?")
        if "lrt::" not in name: continue
        k = name.split("(")[0][:70]; c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
        agg[k][c] += v; cnt[k][c] += 1
for k in agg:
    print("== counters, per dispatch (mean over dispatches):", k)
    for c in sorted(agg[k]): print(f"   {c:32s} {agg[k][c] / max(cnt[k][c], 1):.6g}   (dispatches {cnt[k][c]})")
for k in agg:
    if "k_render" not in k: continue
    a = {c: agg[k][c] / max(cnt[k][c], 1) for c in agg[k]}
    print("== derived,", k)
    if "GRBM_GUI_ACTIVE" in a and "SQ_ACTIVE_INST_VALU" in a:
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0                      # summed over the 8 XCDs
        # Two figures, neither a "fraction busy" in the strict sense (the old "VALU busy" printed 105 - 116 % on some
        # configurations).  (i) issue-slot occupancy: a wave64 VALU instruction holds its SIMD's issue port for 4 cycles
        # (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"; transcendentals 8, so this is a LOWER bound of the port's busy time):
        # SQ_INSTS_VALU x 4 / (cycles x 1024 SIMDs).  (ii) the sum over waves of quad-cycles with a VALU instruction in flight over the SIMD
        # quad-cycles: in-flight intervals of different waves overlap in the pipeline, so this one can exceed 100 % and is only a
        # relative measure between builds of the same kernel.
        if "SQ_INSTS_VALU" in a:
            print(f"   GPU-active cycles {cyc:.4g}; VALU issue-slot occupancy >= {a['SQ_INSTS_VALU'] * 4 / (cyc * 1024):.1%} (SQ_INSTS_VALU x 4 cycles / (cycles x 1024 SIMDs))")
        print(f"   VALU in-flight quad-cycles, summed over waves / SIMD quad-cycles: {a['SQ_ACTIVE_INST_VALU'] / (cyc * 1024 / 4):.2f} (not a fraction: overlapping waves; relative measure only)")
    if "SQ_THREAD_CYCLES_VALU" in a and "SQ_ACTIVE_INST_VALU" in a:
        print(f"   VALU lane utilisation {a['SQ_THREAD_CYCLES_VALU'] / (a['SQ_ACTIVE_INST_VALU'] * 64):.1%}")
    if "SQ_WAIT_ANY" in a and "SQ_WAVE_CYCLES" in a:
        print(f"   wave cycles waiting {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.1%}")
    if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
        rd, wr = a["FETCH_SIZE"] * 1024 * FETCH_FACTOR, a["WRITE_SIZE"] * 1024
        print(f"   HBM traffic per launch: read {rd / 1e9:.2f} GB (FETCH_SIZE x {FETCH_FACTOR}), write {wr / 1e9:.2f} GB, total {(rd + wr) / 1e9:.2f} GB")
        workload = None
        try:
            line = [l for l in open(os.path.join(out, "bench_trace.log")) if l.startswith("{")][-1]
            workload = json.loads(line)["config"]["workload"]
        except Exception:
            pass
        ksid = None
        try:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            import bench
            ksid = bench.kernel_source_id()
        except Exception:
            pass
        json.dump({"kernel": k, "workload": workload, "kernel_source_id": ksid, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                   "traffic_bytes_per_launch": rd + wr, "bench_args": sys.argv[2:],
                   "fetch_factor": FETCH_FACTOR,
                   "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB; FETCH_SIZE x fetch_factor (guide value, reproduced on the record streams: profiles/r02_stream_calibration.txt)"},
                  open(os.path.join(out, "traffic.json"), "w"), indent=1)
