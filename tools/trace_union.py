"""
Turns a `rocprofv3 --kernel-trace --output-format csv` file (one row per dispatch with start / end timestamps) into what
bench.py's `roofline` object reports, so that the number can be re-derived from a committed profile:

  * per kernel family: launches, summed duration, average duration, and the length of the UNION of their intervals
    (launches on the look-ahead stream overlap launches on the caller's stream; summing durations double counts, and two
    overlapping launches stretch each other);
  * for the MFMA GEMM family: algorithmic flop / union-busy time = achieved TFLOP/s, and its fraction of the fp64 MFMA peak.

The algorithmic flops of a step are not in the trace; they are given on the command line (--flops-per-step, default: the
GEMM share of the benchmarked step) together with the number of steps the traced command ran.

usage: python tools/trace_union.py trace.csv [--steps K] [--skip-first S] [--flops-per-step F] [--json out.json] [--timeline A:B]
"""
import argparse
import csv
import json
import re
import sys

PEAK_FP64 = 78.6e12

FAMILIES = [
	("gemm", re.compile(r"stpy::(gemm_nt_(dtv_|k128_|sliver_)?kernel|trsm_strip_kernel)")),
	("diag_block", re.compile(r"stpy::potf2_trtri")),
	("panel_fused", re.compile(r"stpy::panel_")),
	("trsv", re.compile(r"stpy::trsv_")),
	("predict", re.compile(r"stpy::predict_")),
	("gram_prep", re.compile(r"stpy::(prep_points|gram_)")),
	("rff", re.compile(r"stpy::rff_")),
	("splitk_reduce", re.compile(r"stpy::splitk_reduce")),
]


def family(name):
	for f, rx in FAMILIES:
		if rx.search(name):
			return f
	return "other"


def short(name):
	m = re.search(r"stpy::(\w+)(<[^(]*>)?", name)
	return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def union_len(iv):
	iv = sorted(iv)
	tot, cs, ce = 0, None, None
	for a, b in iv:
		if cs is None or a > ce:
			if cs is not None:
				tot += ce - cs
			cs, ce = a, b
		elif b > ce:
			ce = b
	if cs is not None:
		tot += ce - cs
	return tot


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("trace")
	ap.add_argument("--steps", type=int, default=1, help="benchmark steps inside the analysed window")
	ap.add_argument("--skip-first", type=float, default=0.0, help="fraction of the trace (by time) to skip: warm-up steps")
	ap.add_argument("--flops-per-step", type=float, default=None, help="algorithmic flops of the GEMM family per step")
	ap.add_argument("--n", type=int, default=65536)
	ap.add_argument("--m", type=int, default=4096)
	ap.add_argument("--json", default=None)
	ap.add_argument("--timeline", default=None, help="A:B -- print dispatches number A..B (after skipping) with stream and gaps")
	args = ap.parse_args()

	rows = []
	with open(args.trace, newline="") as fh:
		for r in csv.DictReader(fh):
			if r.get("Kind", "KERNEL_DISPATCH") != "KERNEL_DISPATCH":
				continue
			rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0"),
						 int(r.get("Grid_Size_X", 0) or 0), int(r.get("Workgroup_Size_X", 1) or 1), int(r.get("LDS_Block_Size", 0) or 0)))
	rows.sort()
	rows = [r for r in rows if "stpy::" in r[2]]
	if not rows:
		sys.exit("no stpy:: kernels in the trace")
	t0, t1 = rows[0][0], max(r[1] for r in rows)
	cut = t0 + (t1 - t0) * args.skip_first
	rows = [r for r in rows if r[0] >= cut]
	span = (max(r[1] for r in rows) - rows[0][0]) * 1e-6          # ms

	fam = {}
	per_kernel = {}
	for a, b, name, q, st, gx, wx, lds in rows:
		f = family(name)
		d = fam.setdefault(f, {"launches": 0, "sum_ms": 0.0, "iv": []})
		d["launches"] += 1
		d["sum_ms"] += (b - a) * 1e-6
		d["iv"].append((a, b))
		k = per_kernel.setdefault(short(name), {"launches": 0, "sum_ms": 0.0, "iv": []})
		k["launches"] += 1
		k["sum_ms"] += (b - a) * 1e-6
		k["iv"].append((a, b))
	out = {"trace": args.trace, "window_ms": round(span, 3), "steps": args.steps, "families": {}, "kernels": {}}
	for f, d in sorted(fam.items(), key=lambda kv: -kv[1]["sum_ms"]):
		out["families"][f] = {"launches": d["launches"], "sum_ms": round(d["sum_ms"], 3), "avg_ms": round(d["sum_ms"] / d["launches"], 5),
							  "union_busy_ms": round(union_len(d["iv"]) * 1e-6, 3)}
	for k, d in sorted(per_kernel.items(), key=lambda kv: -kv[1]["sum_ms"])[:24]:
		out["kernels"][k] = {"launches": d["launches"], "sum_ms": round(d["sum_ms"], 3), "avg_ms": round(d["sum_ms"] / d["launches"], 5),
							 "union_busy_ms": round(union_len(d["iv"]) * 1e-6, 3)}
	out["all_stpy_union_busy_ms"] = round(union_len([(a, b) for a, b, *_ in rows]) * 1e-6, 3)
	F = args.flops_per_step
	if F is None:
		n, m = args.n, args.m
		F = n ** 3 / 3.0 + float(n) * n * m          # potrf + block solve: what runs on the MFMA GEMM
	if "gemm" in out["families"]:
		g = out["families"]["gemm"]
		busy_s = g["union_busy_ms"] * 1e-3
		out["gemm_roofline"] = {"algorithmic_flops": F * args.steps, "union_busy_ms_per_step": round(g["union_busy_ms"] / args.steps, 3),
								"achieved_tflops": round(F * args.steps / busy_s / 1e12, 2), "peak_tflops": PEAK_FP64 / 1e12,
								"frac": round(F * args.steps / busy_s / PEAK_FP64, 4),
								"sum_of_durations_ms_per_step": round(g["sum_ms"] / args.steps, 3),
								"note": "sum of durations exceeds the union when look-ahead launches overlap the trailing update; the union is the time the MFMA GEMM family kept the chip"}
	print(json.dumps(out, indent=1))
	if args.json:
		with open(args.json, "w") as fh:
			json.dump(out, fh, indent=1)
	if args.timeline:
		a, b = [int(v) for v in args.timeline.split(":")]
		base = rows[a][0]
		prev_end = {}
		print("\n#  idx  queue  start_us  dur_us  gap_same_queue_us  grid_wgs  lds  kernel")
		for i in range(a, min(b, len(rows))):
			s, e, name, q, st, gx, wx, lds = rows[i]
			gap = (s - prev_end[q]) * 1e-3 if q in prev_end else 0.0
			prev_end[q] = e
			print("%5d  %4s  %10.1f  %8.1f  %8.1f  %6d  %6d  %s" % (i, q, (s - base) * 1e-3, (e - s) * 1e-3, gap, gx // max(wx, 1), lds, short(name)))


if __name__ == "__main__":
	main()
