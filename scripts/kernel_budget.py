"""Kernel budget of a rocprofv3 --kernel-trace --stats run: usage kernel_budget.py <..._kernel_stats.csv> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 22
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time total %.3f s" % (tot / 1e9))
for r in rows[:n]:
    print("%-100s calls %6s  avg %9.3f ms  total %7.3f s  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e9, 100 * float(r["TotalDurationNs"]) / tot))
