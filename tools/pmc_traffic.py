"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py into the per-launch HBM
traffic of the GEMM kernel that bench.py reports as roofline.traffic.
usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
The record carries the version string of the library the passes ran with (stpy_version(): it contains a hash of the kernel
sources), so that bench.py pairs it only with runs of the same build."""
import collections, csv, ctypes, json, os, re, sys


def library_version():
	here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	lib = ctypes.CDLL(os.path.join(here, "stpy_amd", "libstpy_hip.so"))
	lib.stpy_version.restype = ctypes.c_char_p
	return lib.stpy_version().decode()

def collect(path, counter):
	per = collections.defaultdict(lambda: [0, 0.0])
	for r in csv.DictReader(open(path)):
		if r["Counter_Name"] != counter:
			continue
		name = r["Kernel_Name"]
		if not name.startswith("void stpy::") and "stpy::" not in name[:40]:
			continue
		name = re.sub(r"\(.*", "", name.replace("void ", ""))
		per[name][0] += 1
		per[name][1] += float(r["Counter_Value"])
	return {k: {"dispatches": v[0], "sum_KB": v[1]} for k, v in per.items()}

def main():
	fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
	g = lambda d: [(k, v) for k, v in d.items() if k.startswith("stpy::gemm_nt_kernel<double") or k.startswith("stpy::gemm_nt_dtv_kernel") or k.startswith("stpy::gemm_nt_k128_kernel") or k.startswith("stpy::gemm_nt_sliver_kernel") or k.startswith("stpy::trsm_strip_kernel")]
	launches = sum(v["dispatches"] for _, v in g(fetch))
	f_bytes = sum(v["sum_KB"] for _, v in g(fetch)) * 1024.0
	w_bytes = sum(v["sum_KB"] for _, v in g(write)) * 1024.0
	total = 2.0 * f_bytes + w_bytes
	out = {
		"_what": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes) over `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline` (N=65536, d=16, M=4096, fp64); all dispatches of stpy::gemm_nt_dtv_kernel<...> and stpy::gemm_nt_kernel<double,...> summed",
		"library_version": library_version(),
		"gemm_launches": launches, "FETCH_SIZE_bytes_raw": f_bytes, "WRITE_SIZE_bytes": w_bytes, "hbm_bytes_corrected": total,
		"correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B for wide coalesced streams (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact. The 8-byte-per-lane C-tile reads are not a calibrated access width, so 2x is an upper bound for them.",
		"per_launch_hbm_bytes": total / max(launches, 1),
		"per_kernel": {"FETCH_SIZE": fetch, "WRITE_SIZE": write},
	}
	json.dump(out, open(sys.argv[3], "w"), indent=1)
	print("gemm launches %d, fetch %.1f GB (raw), write %.1f GB, corrected total %.1f GB, per launch %.3f GB" % (launches, f_bytes / 1e9, w_bytes / 1e9, total / 1e9, total / max(launches, 1) / 1e9))

if __name__ == "__main__":
	main()
