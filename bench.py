"""
bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): "GP fit+mean_var wall-time (s) & fp64 MFMA util %, N=65 536 d=16".
One step = GaussianProcess.fit_gp(x, y) + GaussianProcess.mean_std(xtest) on synthetic data of that
shape (x ~ U(-1,1)^{N x d}, y = sin(sum x) + 0.1 N(0,1), M = 4096 test points,
SE kernel gamma = sqrt(d), s = 0.1, kappa = 1, fp64), inputs resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (block-cyclic path)

Rank 0 prints ONE JSON line.  `value` = seconds per step (max over ranks), lower is better.
`roofline` is for the dominant kernel, gemm_nt_kernel<double> (the fp64-MFMA contraction under
potrf and trsm): achieved = sum of algorithmic flops of its launches / sum of their durations,
measured live with HIP events recorded on the launch stream inside libstpy_hip (stpy_profile_*).
`cpu_baseline` times the CPU oracle (numpy/LAPACK restatement, kind "port") on a bounded sample.
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6       # MI355X fp64 matrix, vendor dense figure


def synth(n, d, m, device, seed=1234):
	gx = torch.Generator().manual_seed(seed)
	x = torch.rand(n, d, generator=gx, dtype=torch.float64) * 2 - 1
	gy = torch.Generator().manual_seed(seed + 1)
	y = torch.sin(x.sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=gy, dtype=torch.float64)
	gt = torch.Generator().manual_seed(seed + 2)
	xt = torch.rand(m, d, generator=gt, dtype=torch.float64) * 2 - 1
	return x.to(device), y.to(device), xt.to(device)


def flops_fit_predict(n, m):
	"""F = N^3/3 + 2N^2 + N^2 M + 4NM."""
	return n ** 3 / 3.0 + 2.0 * n * n + float(n) * n * m + 4.0 * n * m


def cpu_baseline(d, budget_n=16384, budget_m=2048):
	"""CPU oracle (port) on a bounded sample of the same workload; ~10-30 s of host work."""
	from oracle import gp_oracle as O           # checker/baseline only -- never on the product path
	n, m = budget_n, budget_m
	x, y, xt = synth(n, d, m, "cpu", seed=4321)
	spec = [("squared_exponential", {"gamma": math.sqrt(d), "kappa": 1.0}, "-")]
	t0 = time.perf_counter()
	L, alpha = O.fit(x.numpy(), y.numpy(), spec, 0.1)
	mu, std = O.mean_std(x.numpy(), L, alpha, xt.numpy(), spec)
	t = time.perf_counter() - t0
	try:
		cores = len(os.sched_getaffinity(0))
	except Exception:
		cores = os.cpu_count()
	return {"value": round(t, 4), "unit": "s", "cores": cores, "kind": "port",
			"sample": "oracle fit+mean_std (numpy/LAPACK Cholesky restatement) at N=%d, M=%d, d=%d fp64; "
					  "%.3e flop = 1/%.0f of the benchmarked step" % (n, m, d, flops_fit_predict(n, m), flops_fit_predict(65536, 4096) / flops_fit_predict(n, m)),
			"gflops": round(flops_fit_predict(n, m) / t / 1e9, 1)}


def main():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--n", "--train-points", dest="n", type=int, default=65536)
	ap.add_argument("--d", type=int, default=16)
	ap.add_argument("--m", type=int, default=4096)
	ap.add_argument("--nb", type=int, default=0)
	ap.add_argument("--no-cpu-baseline", action="store_true")
	args = ap.parse_args()

	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		if world == 1 and args.gpus > 1:
			raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
	# STPY_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box: the ranks share card 0 and the
	# collectives are staged through the host (functional check of this script only, not a measurement)
	backend = os.environ.get("STPY_BENCH_BACKEND", "nccl")
	if backend != "nccl":
		local_rank = local_rank % max(torch.cuda.device_count(), 1)
	torch.cuda.set_device(local_rank)
	dev = torch.device("cuda", local_rank)
	# STPY_BENCH_FORCE_DIST=1: take the block-cyclic code path (process group, sub-communicators, panel
	# pipeline) even with one rank -- a functional check of the RCCL set-up on a one-GPU box
	force_dist = world == 1 and os.environ.get("STPY_BENCH_FORCE_DIST", "0") == "1"
	dist_path = world > 1 or force_dist
	if force_dist:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
		os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
	if dist_path:
		import torch.distributed as dist
		if backend == "nccl":
			dist.init_process_group(backend="nccl", device_id=dev)
		else:
			dist.init_process_group(backend=backend)

	from stpy_amd import GaussianProcess, _lib
	lib = _lib.load()

	n, d, m = args.n, args.d, args.m
	gamma, s = math.sqrt(d), 0.1
	x, y, xt = synth(n, d, m, dev)

	if dist_path:
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		gp = DistributedGaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	else:
		gp = GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	gp.nb = args.nb

	def step():
		gp.fit_gp(x, y)
		return gp.mean_std(xt)

	def barrier():
		if dist_path:
			torch.distributed.barrier()
		torch.cuda.synchronize()

	for _ in range(args.warmup):
		step()
	barrier()
	lib.stpy_profile_enable(1)
	t0 = time.perf_counter()
	for _ in range(args.steps):
		mu, std = step()
	barrier()
	elapsed = time.perf_counter() - t0
	lib.stpy_profile_enable(0)
	if world > 1:
		tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
		torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
		elapsed = float(tt.item())
	sec_per_step = elapsed / args.steps

	# ---- live roofline of the dominant kernel from the event log (this rank)
	def prof(tag):
		ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
		_lib.check(lib.stpy_profile_read(tag, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt)), "stpy_profile_read")
		return ms.value, fl.value, cnt.value
	tags = {"syrk": 0, "panel_gemm": 1, "trsm_gemm": 2, "potf2": 3, "gemm_api": 4}
	pr = {k: prof(v) for k, v in tags.items()}
	# the block-cyclic path issues its trailing updates through stpy_gemm_nt_bc (tag 4)
	gemm_tags = ["syrk", "panel_gemm", "trsm_gemm"] + (["gemm_api"] if dist_path else [])
	g_ms = sum(pr[t][0] for t in gemm_tags)
	g_fl = sum(pr[t][1] for t in gemm_tags)
	g_cnt = sum(pr[t][2] for t in gemm_tags)
	# launches on the look-ahead stream overlap the trailing update: time them as the union of intervals
	ub, uf, uc = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
	_lib.check(lib.stpy_profile_read_union(0b10111 if dist_path else 0b0111, ctypes.byref(ub), ctypes.byref(uf), ctypes.byref(uc)), "stpy_profile_read_union")
	achieved = uf.value / (ub.value * 1e-3) / 1e12 if ub.value > 0 else 0.0

	# HBM traffic of the dominant kernel: PMC counters cannot be collected inside this process, so the
	# value comes from the committed separate-pass rocprofv3 --pmc runs of this same command
	# (profiles/*_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), per launch.
	traffic = None
	try:
		import glob
		cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
		if cands and n == 65536 and d == 16 and m == 4096 and not dist_path:
			with open(cands[-1]) as fh:
				traffic = round(json.load(fh)["per_launch_hbm_bytes"])
	except Exception:
		traffic = None

	if rank == 0:
		F = flops_fit_predict(n, m)
		out = {
			"metric": "GP fit+mean_var wall-time (s), N=%d d=%d fp64" % (n, d),
			"value": round(sec_per_step, 4), "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(sec_per_step * 1e3, 2), "higher_is_better": False,
			"scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (REHEARSAL: %s backend, ranks share one GPU)" % backend,
			"config": {"workload": "GaussianProcess.fit_gp + mean_std, SE kernel gamma=sqrt(d), s=0.1, N=%d train, M=%d test, d=%d, fp64" % (n, m, d),
					   "n": n, "m": m, "d": d, "nb": args.nb or ("distribution block %d" % gp.NB if dist_path else "potrf panels 1024, recursive block solve (library defaults)"),
					   "parallelism": "2-D block-cyclic over %d GPU%s" % (world, "s" if world > 1 else " (forced: functional check of the distributed code path)") if dist_path else "single GPU"},
			"step_tflops": round(F / sec_per_step / 1e12, 2),
			"step_frac_of_fp64_mfma_peak": round(F / sec_per_step / 1e12 / (PEAK_FP64_MFMA_TFLOPS * world), 4),
			"roofline": {"bound": "mfma", "kernel": "stpy::gemm_nt_dtv_kernel / gemm_nt_kernel<double>" if dist_path else "stpy::gemm_nt_dtv_kernel<SUB> (+ gemm_nt_kernel<double,...> for the ragged / fused-epilogue launches)", "achieved": round(achieved, 2), "peak": PEAK_FP64_MFMA_TFLOPS,
						 "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP64_MFMA_TFLOPS, 4), "traffic": traffic,
						 "traffic_note": "HBM bytes per launch of this kernel from separate rocprofv3 --pmc passes of this command (profiles/), 2*FETCH_SIZE + WRITE_SIZE",
						 "launches": int(g_cnt), "avg_launch_ms": round(g_ms / max(g_cnt, 1), 4), "busy_ms_per_step": round(ub.value / args.steps, 2),
						 "algorithmic_gflop_per_launch": round(g_fl / max(g_cnt, 1) / 1e9, 3)},
			"breakdown_ms_per_step": {k: round(v[0] / args.steps, 2) for k, v in pr.items()},
			"result_check": {"mu_norm": float(torch.norm(mu)), "std_mean": float(std.mean()), "nan": bool(torch.isnan(std).any())},
		}
		if world == 1 and not args.no_cpu_baseline:
			out["cpu_baseline"] = cpu_baseline(d)
		print(json.dumps(out), flush=True)
	if dist_path:
		torch.distributed.destroy_process_group()


if __name__ == "__main__":
	try:
		main()
	except Exception as exc:                         # noqa: BLE001
		# a multi-GPU run that dies must still say why on the one line the driver reads (value null = not measured)
		if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0:
			import traceback
			print(json.dumps({"metric": "GP fit+mean_var wall-time (s), N=65536 d=16 fp64", "value": None, "unit": "s",
							  "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "higher_is_better": False, "scaling": "strong",
							  "error": "%s: %s" % (type(exc).__name__, exc), "traceback_tail": traceback.format_exc()[-1500:]}), flush=True)
		raise
