"""Summarise a rocprofv3 output directory produced by scripts/profile_bench.sh."""
import csv, glob, os, sys, collections
out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, "**", pattern), recursive=True))
for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, out))
    for i, row in enumerate(csv.reader(open(f))):
        if i < 12: print("  ", ", ".join(row))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in find("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?").split("(")[0][:60]; c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
        agg[k][c] += v; cnt[k][c] += 1
for k in agg:
    print("== counters (sum over dispatches):", k)
    for c in sorted(agg[k]): print(f"   {c:32s} {agg[k][c]:.6g}   (dispatches {cnt[k][c]})")
