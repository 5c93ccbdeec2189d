"""Condense rocprofv3 csv output (kernel stats + counter_collection) into a short text summary."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
for f in glob.glob(os.path.join(out, "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats", f)
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get("Kernel_Name", "?")
        short = "pass_up" if "pass_up" in name else "pass_dw" if "pass_dw" in name else "naive" if "naive" in name else None
        if not short:
            continue
        agg[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in agg:
    print("== counters (mean per dispatch)", k)
    for c in sorted(agg[k]):
        vals = agg[k][c]
        print(f"  {c:36s} {sum(vals)/len(vals):.4g}  (n={len(vals)})")
