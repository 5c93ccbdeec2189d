"""profiles/kernel_trace.json from a rocprofv3 --kernel-trace --stats CSV of `bench.py` (the *kernel_stats.csv): average duration of the product's
two kernels and the stamp of the kernel build it was collected on -- bench.py quotes it (roofline.rocprof_kernel_trace) only while its
KERNELS_STAMP is the same, like profiles/traffic.json.   usage: make_kernel_trace.py <kernel_stats.csv> <kernels_stamp> <name kept in profiles/> [workload]"""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1])))
avg = {}
for r in rows:
    name = r.get("Name") or r.get("KernelName") or ""
    ns = float(r.get("AverageNs") or r.get("Average") or 0.0)
    calls = int(float(r.get("Calls") or 0))
    for key in ("hxv_pass_up", "hxv_pass_dw", "hxv_up_job"):
        if key in name and calls >= 20:          # (the plain product's instantiations: the bench's timed loop, not the Lanczos legs' few calls)
            k = key + ("" if "HIP_vector_type" in name.split("(")[0] else "<real>")
            if k not in avg or calls > avg[k][1]:
                avg[k] = (ns * 1e-6, calls, name)
out = {"workload": sys.argv[4] if len(sys.argv) > 4 else "C3", "n_gpus": 1, "kernels_stamp": sys.argv[2], "file": "profiles/" + sys.argv[3],
       "avg_ms": {k: round(v[0], 4) for k, v in avg.items()}, "calls": {k: v[1] for k, v in avg.items()}, "kernel_names": {k: v[2] for k, v in avg.items()}}
if "hxv_pass_up" in avg and "hxv_pass_dw" in avg:
    out["product_ms"] = round(avg["hxv_pass_up"][0] + avg["hxv_pass_dw"][0], 4)
json.dump(out, open("profiles/kernel_trace.json", "w"), indent=1)
print(json.dumps(out, indent=1))
