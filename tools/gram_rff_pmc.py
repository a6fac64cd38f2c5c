"""Summary of the rocprofv3 passes over tools/gram_rff_only.py (kernel stats + three --pmc passes: WRITE_SIZE; FETCH_SIZE; SQ counters)
for the two HBM-write-side kernels: per-dispatch counter values of the LAST launch of each kernel, derived readings.
usage: python tools/gram_rff_pmc.py <dir with gr_kt/ gr_write/ gr_fetch/ gr_sq/> <out.json>"""
import csv, json, os, sys

def last_dispatch(path, kernel_substr):
	"""{counter: value} of the last dispatch whose kernel name contains kernel_substr (summed over the rows of that dispatch)"""
	per = {}
	for r in csv.DictReader(open(path)):
		if kernel_substr in r["Kernel_Name"]:
			per.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
			per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
	if not per:
		return {}
	return per[max(per)]

def kernel_ms(stats_csv, substr):
	out = []
	for r in csv.DictReader(open(stats_csv)):
		if substr in r["Name"]:
			out.append({"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) * 1e-6, "min_ms": float(r["MinNs"]) * 1e-6})
	return out

def main():
	d, outp = sys.argv[1], sys.argv[2]
	res = {"_what": "rocprofv3 --kernel-trace --stats and three --pmc passes (WRITE_SIZE; FETCH_SIZE; SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES "
					"SQ_WAIT_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE -- separate runs, no tracing options) over `python3 tools/gram_rff_only.py`; counter values of the last "
					"launch of each kernel; WRITE_SIZE / FETCH_SIZE in KB as rocprofv3 prints them, converted to bytes here"}
	for tag, sub, alg_out in (("gram_fill", "gram_fill_f64_kernel", 131328 * 128 * 128 * 8.0),
							  ("rff_stream_bf16x3", "rff_stream_bf16x3_kernel", 262144.0 * 32768 * 4)):
		e = {"kernel": sub, "algorithmic_bytes_out": alg_out}
		ks = kernel_ms(os.path.join(d, "gr_kt", "gr_kernel_stats.csv"), sub)
		if ks:
			e["kernel_ms_avg"], e["kernel_ms_min"], e["launches"] = round(ks[0]["avg_ms"], 4), round(ks[0]["min_ms"], 4), ks[0]["calls"]
			e["TB_per_s_out_at_min"] = round(alg_out / (ks[0]["min_ms"] * 1e-3) / 1e12, 3)
		w = last_dispatch(os.path.join(d, "gr_write", "gr_counter_collection.csv"), sub)
		f = last_dispatch(os.path.join(d, "gr_fetch", "gr_counter_collection.csv"), sub)
		sq = last_dispatch(os.path.join(d, "gr_sq", "gr_counter_collection.csv"), sub)
		if w:
			e["WRITE_SIZE_bytes"] = w.get("WRITE_SIZE", 0.0) * 1024
		if f:
			e["FETCH_SIZE_bytes_raw"] = f.get("FETCH_SIZE", 0.0) * 1024
			e["FETCH_SIZE_bytes_x2"] = 2 * e["FETCH_SIZE_bytes_raw"]
		if sq:
			e["SQ"] = sq
			cyc = sq.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # summed over the 8 XCDs
			if cyc > 0:
				e["mfma_busy_frac"] = round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * 1024), 4)          # 1024 SIMDs
				if ks:
					e["clock_GHz_during_counter_pass"] = round(cyc / (ks[0]["avg_ms"] * 1e-3) / 1e9, 3)
			e["valu_wave_instructions"] = sq.get("SQ_INSTS_VALU")
		res[tag] = e
	json.dump(res, open(outp, "w"), indent=1)
	print(json.dumps(res, indent=1)[:3000])

if __name__ == "__main__":
	main()
