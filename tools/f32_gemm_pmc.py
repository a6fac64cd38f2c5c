"""Summary of the rocprofv3 passes over tools/f32_gemm_only.py: kernel time and per-launch counters of the bf16x3 kernel and of the
fp32-MFMA kernel on the same trailing update.   usage: python tools/f32_gemm_pmc.py <dir with kt/ sq/ lds/ mem/> <out.json>"""
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

def main():
	d, outp = sys.argv[1], sys.argv[2]
	n, k = 32768, 1024
	flops = float(n) * n * k          # lower-triangular update
	res = {"_what": "rocprofv3 --kernel-trace --stats and --pmc passes (separate runs, no tracing options) over `python3 tools/f32_gemm_only.py`: fp32 trailing update "
					"n = 32768 (lower), K = 1024, C -= A A^T; counters of the last launch of each kernel", "algorithmic_flop_per_launch": flops}
	stats = find(d, "*kernel_stats.csv")
	for tag, sub in (("bf16x3", "gemm_nt_bf3_kernel"), ("fp32_mfma", "gemm_nt_dtv_kernel<float")):
		e = {"kernel": sub}
		if stats:
			for r in csv.DictReader(open(stats)):
				if sub in r["Name"]:
					e["launches"], e["kernel_ms_avg"], e["kernel_ms_min"] = int(r["Calls"]), round(float(r["AverageNs"]) * 1e-6, 4), round(float(r["MinNs"]) * 1e-6, 4)
					e["TFLOPs_fp32_equivalent_at_min"] = round(flops / (float(r["MinNs"]) * 1e-9) / 1e12, 1)
		for sub_dir in ("sq", "lds", "mem_f", "mem_w"):
			f = find(os.path.join(d, sub_dir), "*counter_collection.csv")
			if f:
				e.update({k2: v for k2, v in last(f, sub).items()})
		if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"] > 0:
			# SQ_BUSY_CYCLES is summed over the 8 XCDs x shader engines; the guide's reading: MFMA-busy cycles per SIMD / kernel cycles
			e["mfma_busy_cycles_per_simd"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0)
		if "GRBM_GUI_ACTIVE" in e and "kernel_ms_avg" in e:
			e["clock_GHz_during_counter_pass"] = round(e["GRBM_GUI_ACTIVE"] / 8.0 / (e["kernel_ms_avg"] * 1e-3) / 1e9, 3)
			if "mfma_busy_cycles_per_simd" in e:
				e["mfma_pipe_busy_frac"] = round(e["mfma_busy_cycles_per_simd"] / (e["GRBM_GUI_ACTIVE"] / 8.0), 4)
		if "SQ_LDS_BANK_CONFLICT" in e and e.get("SQ_LDS_IDX_ACTIVE"):
			e["lds_conflict_frac_of_active"] = round(e["SQ_LDS_BANK_CONFLICT"] / e["SQ_LDS_IDX_ACTIVE"], 4)
		res[tag] = e
	json.dump(res, open(outp, "w"), indent=1)
	print(json.dumps(res, indent=1))

if __name__ == "__main__":
	main()
