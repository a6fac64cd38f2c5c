"""Summary of rocprofv3 passes over `tools/gemm_only.py n k` (the fp64 trailing update alone, C -= P P^T, lower tiles): kernel time and
per-launch counters of gemm_nt_dtv_kernel<double, 1>.   usage: python tools/f64_gemm_pmc.py <dir with kt/ sq/ lds/ mem_f/ mem_w/> n k <out.json>"""
import csv, glob, json, os, sys


def last(path, sub):
	per = {}
	for r in csv.DictReader(open(path)):
		if sub in r["Kernel_Name"]:
			per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
			per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
	return per[max(per)] if per else {}


def find(d, pat):
	g = glob.glob(os.path.join(d, "**", pat), recursive=True)
	return g[0] if g else None


d, n, k, outp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sub = "gemm_nt_dtv_kernel<double"
flops = float(n) * (n + 128) * k
e = {"_what": "rocprofv3 --kernel-trace --stats and --pmc passes (separate runs, no tracing options) over `python3 tools/gemm_only.py %d %d`: "
              "fp64 trailing update alone on the chip; counters of the last launch" % (n, k), "kernel": sub, "algorithmic_flop_per_launch": flops}
stats = find(os.path.join(d, "kt"), "*kernel_stats.csv")
if stats:
	for r in csv.DictReader(open(stats)):
		if sub in r["Name"]:
			e["launches"], e["kernel_ms_avg"], e["kernel_ms_min"] = int(r["Calls"]), round(float(r["AverageNs"]) * 1e-6, 4), round(float(r["MinNs"]) * 1e-6, 4)
			e["TFLOPs_at_min"] = round(flops / (float(r["MinNs"]) * 1e-9) / 1e12, 2)
			e["frac_of_78.6"] = round(e["TFLOPs_at_min"] / 78.6, 4)
for sd in ("sq", "lds", "mem_f", "mem_w"):
	f = find(os.path.join(d, sd), "*counter_collection.csv")
	if f:
		e.update(last(f, sub))
if "GRBM_GUI_ACTIVE" in e and "kernel_ms_avg" in e:
	e["clock_GHz_during_counter_pass"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / (e["kernel_ms_avg"] * 1e-3) / 1e9, 3)
if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
	# per-SIMD MFMA-busy cycles over the kernel's cycles (one XCD's GRBM count): the guide's reading of the counter pair
	e["mfma_busy_frac"] = round((e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0), 4)
for key in ("FETCH_SIZE", "WRITE_SIZE"):
	if key in e:
		e[key + "_bytes"] = e[key] * 1024.0
if "FETCH_SIZE_bytes" in e and "WRITE_SIZE_bytes" in e:
	e["hbm_bytes_per_launch_2xfetch_plus_write"] = 2 * e["FETCH_SIZE_bytes"] + e["WRITE_SIZE_bytes"]
	e["algorithmic_bytes_per_launch"] = float(n) * (n + 128) / 2 * 8 * 2 + float(n) * k * 8
json.dump(e, open(outp, "w"), indent=1)
print(json.dumps(e, indent=1))
