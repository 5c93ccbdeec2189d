"""Condense rocprofv3 csv output (kernel stats + counter_collection) of the product kernels into a short text summary,
with the HBM-side bytes per launch: reads = TCC_EA0_RDREQ x 128 B (= FETCH_SIZE x 2 on gfx950), writes = WRITE_SIZE KiB."""
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
KEYS = ("pass_up", "pass_dw", "up_job", "dw_job", "naive", "nonlocal", "lz_", "tr_")
def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = short(row["Name"])
        if s:
            print(f"stats {s:10s} calls {row['Calls']:>4s} avg {float(row['AverageNs'])/1e6:.4f} ms  min {float(row['MinNs'])/1e6:.4f}  max {float(row['MaxNs'])/1e6:.4f}")
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        s = short(row.get("Kernel_Name", "?"))
        if s:
            agg[s][row["Counter_Name"]].append(float(row["Counter_Value"]))
tot = 0.0
for k in agg:
    c = {n: sum(v) / len(v) for n, v in agg[k].items()}
    rd = c.get("TCC_EA0_RDREQ_sum", 0) * 128.0
    wr = c.get("WRITE_SIZE", 0) * 1024.0
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print(f"traffic {k:10s} read {rd/1e9:.3f} GB (FETCH_SIZE x2 = {c.get('FETCH_SIZE',0)*2048/1e9:.3f})  write {wr/1e9:.3f} GB  L2 hit {hit:.4g} miss {miss:.4g} hit-rate {hit/max(hit+miss,1):.3f}")
    if k in ("pass_up", "pass_dw", "up_job", "dw_job"):
        tot += rd + wr
print(f"traffic product total {tot/1e9:.3f} GB")
