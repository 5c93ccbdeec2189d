"""profiles/traffic.json from a scripts/prof_traffic.sh summary (rocprofv3 --pmc, separate passes): HBM-side bytes per product =
sum over the product's two kernels of reads (TCC_EA0_RDREQ x 128 B = FETCH_SIZE x 2, the gfx950 correction of MI355X_MICROARCH.md)
+ writes (WRITE_SIZE x 1024).  usage: make_traffic.py <summary.txt> <kernels_stamp> [workload]"""
import json, re, sys
txt = open(sys.argv[1]).read()
detail, tot = {}, 0.0
for m in re.finditer(r"traffic (\w+)\s+read ([\d.]+) GB \(FETCH_SIZE x2 = ([\d.]+)\)\s+write ([\d.]+) GB\s+L2 hit ([\d.e+]+) miss ([\d.e+]+)", txt):
    k, rd, f2, wr, hit, miss = m.group(1), float(m.group(2)) * 1e9, float(m.group(3)) * 1e9, float(m.group(4)) * 1e9, float(m.group(5)), float(m.group(6))
    if k in ("pass_up", "pass_dw"):
        detail[k] = {"read_bytes": rd, "write_bytes": wr, "fetch_size_x2_bytes": f2, "l2_hit": hit, "l2_miss": miss}
        tot += rd + wr
ms = {m.group(1): float(m.group(2)) for m in re.finditer(r"stats (\w+)\s+calls\s+\d+ avg ([\d.]+) ms", txt)}
wl = sys.argv[3] if len(sys.argv) > 3 else "C3"
alg = {"C3": 5300380800, "C4": 5300380800, "C5": 75644940800}[wl]
json.dump({"workload": wl, "n_gpus": 1, "hbm_bytes_per_product": tot, "algorithmic_bytes": alg, "kernels": detail, "kernel_ms_during_collection": ms,
           "kernels_stamp": sys.argv[2],
           "method": "rocprofv3 --pmc, separate passes (scripts/prof_traffic.sh); reads = TCC_EA0_RDREQ_sum x 128 B (= FETCH_SIZE x 2, gfx950 correction), writes = WRITE_SIZE x 1024"},
          open("profiles/traffic.json" if wl == "C3" else f"profiles/traffic_{wl}.json", "w"), indent=1)
print(wl, json.dumps(detail), tot / 1e9, "GB")
