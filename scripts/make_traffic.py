"""Turn the PMC calibration output (scripts/prof_calib.sh) into profiles/traffic.json:
HBM bytes per product = sum over the product's kernels of TCC_EA0_RDREQ*128 B (all read requests are 128 B on
gfx950: TCC_EA0_RDREQ_128B == TCC_EA0_RDREQ, i.e. FETCH_SIZE*2 as MI355X_MICROARCH.md prescribes) + WRITE_SIZE."""
import json, re, sys
txt = open(sys.argv[1]).read()
out = {}
cur = None
for line in txt.splitlines():
    m = re.match(r"== (\w+)", line)
    if m:
        cur = m.group(1); out[cur] = {}; continue
    m = re.match(r"\s+(\w+)\s+([0-9.e+]+)", line)
    if m and cur:
        out[cur][m.group(1)] = float(m.group(2))
tot = 0.0
detail = {}
for k in ("pass_up", "pass_dw"):
    rd = out[k]["TCC_EA0_RDREQ_sum"] * 128.0
    wr = out[k]["WRITE_SIZE"] * 1024.0
    detail[k] = {"read_bytes": rd, "write_bytes": wr, "fetch_size_kb_raw": out[k]["FETCH_SIZE"], "l2_hit": out[k]["TCC_HIT_sum"], "l2_miss": out[k]["TCC_MISS_sum"]}
    tot += rd + wr
json.dump({"workload": "C3", "n_gpus": 1, "hbm_bytes_per_product": tot, "algorithmic_bytes": 5300380800, "kernels": detail,
           "method": "rocprofv3 --pmc, separate passes; reads = TCC_EA0_RDREQ_sum x 128 B (= FETCH_SIZE x 2, gfx950 correction), writes = WRITE_SIZE x 1024",
           "config": sys.argv[2] if len(sys.argv) > 2 else ""}, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(detail), tot / 1e9, "GB")
