"""Sums the counters of a `rocprofv3 --pmc ...` pass per kernel family (here: the two fp32-on-bf16 update kernels, gemm_nt_bf3_kernel =
operands split on the fly, gemm_bf3p_kernel = operands split once into planes) and prints the matrix-pipe busy fraction
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) where both counters are present.
usage: python tools/pmc_by_kernel.py <rocprofv3 output dir> [...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
	f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
	if not f:
		print(d, "no counter_collection.csv")
		continue
	acc = collections.defaultdict(collections.Counter)
	for r in csv.DictReader(open(f[0])):
		n = r["Kernel_Name"]
		if "bf3" not in n or "split" in n:
			continue
		acc["gemm_bf3p_kernel" if "bf3p" in n else "gemm_nt_bf3_kernel"][r["Counter_Name"]] += float(r["Counter_Value"])
	for k, c in sorted(acc.items()):
		line = {a: "%.4g" % b for a, b in sorted(c.items())}
		if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
			line["mfma_busy_frac"] = "%.3f" % (c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
		print(d.rstrip("/").split("/")[-1], k, line)
