"""
bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): "GP fit+mean_var wall-time (s) & fp64 MFMA util %, N=65 536 d=16".
One step = GaussianProcess.fit_gp(x, y) + GaussianProcess.mean_std(xtest) on synthetic data of that
shape (x ~ U(-1,1)^{N x d}, y = sin(sum x) + 0.1 N(0,1), M = 4096 test points,
SE kernel gamma = sqrt(d), s = 0.1, kappa = 1, fp64), inputs resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (block-cyclic path)

Rank 0 prints ONE JSON line.  `value` = seconds per step (max over ranks), lower is better.
`roofline` is for the dominant kernel family, the fp64-MFMA contraction gemm_nt_dtv_kernel<double,SUB> (+ the sliver / K = 128 /
tile forms of the panel chain) under potrf and trsm: achieved = sum of the algorithmic flops of its launches / the length of
the UNION of their intervals (look-ahead launches overlap the trailing update), measured live with HIP events recorded on the
launch stream inside libstpy_hip (stpy_profile_*); tools/trace_union.py derives the same figure from a rocprofv3 kernel trace
(profiles/r02_*_union.json).  `extra_configs` holds BASELINE configs 2, 3 and 5 with their own roofline fraction and parity.
`cpu_baseline` times the CPU oracle (numpy/LAPACK restatement, kind "port") on a bounded sample, checks the HIP path against
it, and adds the reference-shaped op sequence over an N sweep with a power-law extrapolation.
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
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X fp32 matrix (MI355X_MICROARCH.md: 155 measured)
PEAK_BF16X3_TFLOPS = 2500.0 / 6.0  # fp32-equivalent ceiling of the exact three-way bf16 split (six bf16 MFMA products per fp32 product; dense bf16 peak 2.5 PFLOP/s)
BF16X3_NOTE = ("fp32 in / fp32 out; the large aligned products run on the bf16 matrix cores from an EXACT three-way split of both operands (six products, "
			   "two-level fp32 accumulation; gemm_bf3p_kernel from operands split once into bf16 planes where an operand is reused by many tiles -- the factorisation's panels, the feature slabs -- and gemm_nt_bf3_kernel, which splits on the fly, elsewhere; the two are bit-identical) -- `peak` / `frac` are those of the pipe that bounds the kernel: the dense bf16 MFMA "
			   "peak / 6 products = 416.7 TFLOP/s fp32-equivalent; `ratio_to_fp32_mfma_peak` relates the same rate to the 157.3 TFLOP/s fp32-MFMA "
			   "pipe the config was priced against in rounds 1-2 (a comparison, not a roofline fraction)")


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


def _cores():
	try:
		return len(os.sched_getaffinity(0))
	except Exception:
		return os.cpu_count()


def cpu_baseline(d, gp_factory, dev, budget_n=16384, budget_m=2048, long_sweep=False):
	"""CPU baseline beside the GPU number, all on a BOUNDED sample (about 20-30 s of host work):
	  * the Cholesky restatement of the oracle ("port": what estimator.py:35-37 would cost) at N = 16 384, M = 2048 on all
	    host cores -- the headline `value`; the HIP path runs the same sample and `parity_rel_err` reports mu / sigma against it;
	  * the reference's OWN operation sequence (dense Sigma^T Sigma, lstsq with N right-hand sides, the per-point loop:
	    gauss_procc.py:151-163, :347, :376-378) at N in {1024, 2048, 4096} on all cores and at {1024, 2048} on one thread,
	    a power law fitted through the all-core points and evaluated at N = 65 536 (labelled extrapolated: the reference
	    cannot allocate that size -- five dense N x N matrices).
	"""
	from oracle import gp_oracle as O           # checker/baseline only -- never on the product path
	try:
		from threadpoolctl import threadpool_limits
	except Exception:                               # noqa: BLE001
		threadpool_limits = None
	n, m = budget_n, budget_m
	x, y, xt = synth(n, d, m, "cpu", seed=4321)
	spec = [("squared_exponential", {"gamma": math.sqrt(d), "kappa": 1.0}, "-")]
	t0 = time.perf_counter()
	L, alpha = O.fit(x.numpy(), y.numpy(), spec, 0.1)
	mu, std = O.mean_std(x.numpy(), L, alpha, xt.numpy(), spec)
	t = time.perf_counter() - t0
	del L
	# the same sample through the HIP path: the CPU leg doubles as a parity check at BASELINE config 2's size
	gp = gp_factory()
	gp.fit_gp(x.to(dev), y.to(dev))
	mu_g, std_g = gp.mean_std(xt.to(dev))
	rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
	parity = {"mu": rel(mu_g.cpu().numpy(), mu), "sigma": rel(std_g.cpu().numpy(), std)}
	del gp
	torch.cuda.empty_cache()
	# the same restatement on ONE thread (SURVEY section 8d asks for a one-thread figure), on a smaller bounded sample: N = 8192, M = 1024
	one_thread = None
	if threadpool_limits is not None:
		n1, m1 = 8192, 1024
		x1, y1, xt1 = synth(n1, d, m1, "cpu", seed=4323)
		with threadpool_limits(limits=1):
			t1s = time.perf_counter()
			L1, a1 = O.fit(x1.numpy(), y1.numpy(), spec, 0.1)
			O.mean_std(x1.numpy(), L1, a1, xt1.numpy(), spec)
			t1 = time.perf_counter() - t1s
		del L1
		one_thread = {"seconds": round(t1, 3), "cores": 1, "sample": "same restatement at N=%d, M=%d" % (n1, m1), "gflops": round(flops_fit_predict(n1, m1) / t1 / 1e9, 1)}
	out = {"value": round(t, 4), "unit": "s", "cores": _cores(), "kind": "port", "one_thread": one_thread,
		   "sample": "oracle fit+mean_std (numpy/LAPACK Cholesky restatement) at N=%d, M=%d, d=%d fp64; "
					 "%.3e flop = 1/%.0f of the benchmarked step" % (n, m, d, flops_fit_predict(n, m), flops_fit_predict(65536, 4096) / flops_fit_predict(n, m)),
		   "gflops": round(flops_fit_predict(n, m) / t / 1e9, 1),
		   "parity_rel_err": {k: float("%.3e" % v) for k, v in parity.items()},
		   "parity_note": "HIP path vs this CPU result on the same sample (tolerance 1e-8)"}

	# ---- the reference's own op sequence, N sweep + power law.  LAPACK's pivoted QR (gelsy) is BLAS-2 bound and gets SLOWER
	# with hundreds of threads (measured on the box: N = 2048 6.3 s on 256 threads, 2.5 s on one), so the sweep runs on the
	# thread count a one-GPU job is entitled to here (16) and on one thread; one all-thread point is kept for the record.
	def ref_shaped(nn, threads):
		xs, ys, xts = synth(nn, d, 256, "cpu", seed=4322)
		t1 = time.perf_counter()
		if threads and threadpool_limits is not None:
			with threadpool_limits(limits=threads):
				O.fit_predict_reference_shaped(xs.numpy(), ys.numpy(), xts.numpy(), spec, 0.1)
		else:
			O.fit_predict_reference_shaped(xs.numpy(), ys.numpy(), xts.numpy(), spec, 0.1)
		return time.perf_counter() - t1
	nthr = min(16, _cores())
	ref_shaped(256, nthr)          # untimed: first-call costs of the LAPACK drivers
	sweep, one = {}, {}
	spent = 0.0
	for nn in (1024, 2048, 4096) + ((8192,) if long_sweep else ()):
		if spent > 7.0 and not (long_sweep and nn == 8192):          # bounded: a slow host stops after the sizes that fit the budget
			break
		sweep[nn] = ref_shaped(nn, nthr)
		spent += sweep[nn]
	for nn in (1024, 2048):
		if threadpool_limits is None or spent > 16.0:
			break
		one[nn] = ref_shaped(nn, 1)
		spent += one[nn]
	ref = {"kind": "port of the reference's op sequence (gauss_procc.py:151-163,:347,:376-378: dense Sigma^T Sigma, gelsy lstsq with N right-hand sides, per-point loop)",
		   "threads": nthr, "seconds_by_N": {str(k): round(v, 3) for k, v in sweep.items()}, "one_thread_seconds_by_N": {str(k): round(v, 3) for k, v in one.items()}, "m_test": 256}
	if spent < 20.0:
		ref["all_%d_threads_seconds_at_N1024" % _cores()] = round(ref_shaped(1024, 0), 3)
	if len(sweep) >= 2:
		ks = sorted(sweep)[-2:] if len(sweep) > 2 else sorted(sweep)          # the two largest sizes: the small one is overhead-dominated
		b, a = np.polyfit(np.log([float(k) for k in ks]), np.log([sweep[k] for k in ks]), 1)
		ref["power_law"] = {"exponent": round(float(b), 3), "fit_points": ks, "fit": "two-point fit (the two largest sizes timed)"}
		ref["extrapolated_s_at_N65536"] = round(float(math.exp(a) * 65536.0 ** b), 1)
		ref["extrapolation_note"] = ("TWO-POINT power-law extrapolation over a factor of %d in N, not a measurement (SURVEY section 8d asked for sizes up to 16 384: "
									 "those take minutes per point on this host; --cpu-baseline-long adds N = 8192): at N = 65 536 the sequence needs >= 5 dense "
									 "N x N fp64 matrices (172 GB)" % (65536 // ks[-1]))
	out["reference_shaped"] = ref
	return out


def extra_configs(dev, lib):
	"""The other single-GPU BASELINE.json configurations (parity-test cases, not the bench line), each timed over a few
	repetitions with inputs resident, with the roofline that bounds it and a parity figure measured in the same run."""
	from stpy_amd import GaussianProcess, RFFEmbedding
	out = {}

	def timed(fn, reps=3):
		fn()
		torch.cuda.synchronize()
		ts = []
		for _ in range(reps):
			t0 = time.perf_counter()
			r = fn()
			torch.cuda.synchronize()
			ts.append(time.perf_counter() - t0)
		return min(ts), r

	# C2: N = 16 384, d = 8, SE, fp64
	n, d, m = 16384, 8, 4096
	x, y, xt = synth(n, d, m, dev)
	gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)

	def c2():
		gp.fit_gp(x, y)
		return gp.mean_std(xt)
	t, (mu, std) = timed(c2, reps=5)
	idx = torch.arange(0, n, 8, device=dev)
	mu_tr, _ = gp.mean_std(x[idx])
	expect = y[idx] - 0.01 * gp.A.reshape(-1, 1).to(dev)[idx]
	F = flops_fit_predict(n, m)
	out["C2"] = {"workload": "N=16384 d=8 SE fp64 fit_gp+mean_std, M=4096", "seconds": round(t, 5), "bound": "mfma", "achieved": round(F / t / 1e12, 2),
				 "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(F / t / 1e12 / PEAK_FP64_MFMA_TFLOPS, 4),
				 "parity": {"training_point_identity_rel_err": float("%.2e" % float(torch.norm(mu_tr - expect) / torch.norm(expect))),
							"note": "mean(x_i) = y_i - s^2 alpha_i through the prediction path; vs-oracle parity of this size: cpu_baseline.parity_rel_err and tests/test_gpu_configs.py"}}
	del gp
	torch.cuda.empty_cache()

	# GRAM: the Gram fill of the headline fit (N = 65 536, d = 16, SE, fp64) at the C ABI: lower-only (what fit_gp writes: the tiles the
	# Cholesky reads) and full, against the 8 TB/s spec AND against a plain device fill of the same buffer measured in this process
	from stpy_amd import _lib as L
	n, d = 65536, 16
	xg, _, _ = synth(n, d, 8, dev)
	ilg = torch.full((d,), 1.0 / math.sqrt(d), dtype=torch.float64, device=dev)
	Kg = torch.empty((n, n), dtype=torch.float64, device=dev)
	wsg = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
	gram = {}
	t_fillg, _ = timed(lambda: Kg.fill_(1.0), reps=3)
	for lower, tag in ((1, "lower_only"), (0, "full")):
		tg, _ = timed(lambda: L.check(lib.stpy_gram(L.K_SE, L.F64, L.ptr(xg), n, d, L.ptr(xg), n, d, d, None, L.ptr(ilg), 1.0, 0.0, 0.01, lower, 0,
														  L.ptr(Kg), n, L.ptr(wsg), wsg.numel(), L.stream_ptr()), "gram"), reps=5)
		tiles = (n // 128) * (n // 128 + 1) // 2 if lower else (n // 128) ** 2
		by = tiles * 128 * 128 * 8
		gram[tag] = {"seconds": round(tg, 5), "bytes": by, "achieved": round(by / tg / 1e12, 2), "frac": round(by / tg / 8e12, 4),
					 "frac_of_store_ceiling": round((by / tg) / (n * n * 8 / t_fillg), 4)}
	# parity in the run: 64 sampled rows of the full fill against the oracle's kernel
	from oracle import gp_oracle as O          # checker only
	rows = torch.arange(0, n, n // 64, device=dev)
	refK = O.kernel(xg.cpu().numpy(), xg[rows].cpu().numpy(), [("squared_exponential", {"gamma": math.sqrt(d), "kappa": 1.0}, "-")])
	refK[np.arange(rows.numel()), rows.cpu().numpy()] += 0.01
	gerr = float(np.abs(Kg[rows].cpu().numpy() - refK).max())
	out["GRAM"] = {"workload": "stpy_gram SE N=65536 d=16 fp64 (+ s^2 on the diagonal): the Gram fill of the headline fit", "bound": "hbm", "peak": 8.0, "unit": "TB/s",
				   "seconds": gram["lower_only"]["seconds"], "achieved": gram["lower_only"]["achieved"], "frac": gram["lower_only"]["frac"], "cases": gram,
				   "store_ceiling": {"what": "device fill of the same N x N fp64 buffer, same process", "seconds": round(t_fillg, 5), "TB/s": round(n * n * 8 / t_fillg / 1e12, 2)},
				   "parity": {"max_abs_err_vs_oracle_64_rows": float("%.2e" % gerr), "tolerance": 1e-12}}
	del Kg, wsg, xg
	torch.cuda.empty_cache()

	# C3: N = 65 536, d = 16, Matern-5/2, fp32 + log_marginal; parity against the fp64 HIP path on the same inputs
	n, d, m = 65536, 16, 4096
	x, y, xt = synth(n, d, m, dev)
	x32, y32, xt32 = x.float(), y.float(), xt.float()
	kw = dict(gamma=math.sqrt(d), s=0.3, kernel_name="matern", nu=2.5, d=d)
	g64 = GaussianProcess(**kw)
	g64.fit_gp(x32.double(), y32.double())
	mu64, sd64 = g64.mean_std(xt32.double())
	lm64 = float(g64.log_marginal(g64.kernel_object, {}, 1.0))
	del g64
	torch.cuda.empty_cache()
	g32 = GaussianProcess(**kw)

	def c3():
		g32.fit_gp(x32, y32)
		mu, sd = g32.mean_std(xt32)
		return mu, sd, g32.log_marginal(g32.kernel_object, {}, 1.0)
	t, (mu32, sd32, lm32) = timed(c3, reps=2)
	rel = lambda a, b: float(torch.norm(a.double() - b) / torch.norm(b))
	out["C3"] = {"workload": "N=65536 d=16 Matern-5/2 fp32 fit_gp+mean_std+log_marginal, M=4096, s=0.3", "seconds": round(t, 4), "bound": "mfma",
				 "achieved": round(F_fp(n, m) / t / 1e12, 2), "peak": round(PEAK_BF16X3_TFLOPS, 1), "bound_pipe": "bf16 mfma x6 (exact 3-way split)", "unit": "TFLOP/s",
				 "frac": round(F_fp(n, m) / t / 1e12 / PEAK_BF16X3_TFLOPS, 4),
				 "ratio_to_fp32_mfma_peak": round(F_fp(n, m) / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "arithmetic": BF16X3_NOTE,
				 "parity": {"vs_fp64_hip_rel_err": {"mu": float("%.2e" % rel(mu32, mu64)), "sigma": float("%.2e" % rel(sd32, sd64)),
													"lml": float("%.2e" % (abs(float(lm32) - lm64) / abs(lm64)))}, "tolerance": 1e-3}}
	del g32, mu64, sd64
	torch.cuda.empty_cache()

	# C5: RFF embed N = 262 144, d = 64, m = 32 768, fp32 at the C ABI (output buffer allocated once)
	n, d, m = 262144, 64, 32768
	gen = torch.Generator().manual_seed(1237)
	xr = torch.rand(n, d, generator=gen, dtype=torch.float32).to(dev)
	np.random.seed(1237)
	emb = RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
	W = emb.W.float().to(dev)
	z = torch.empty((n, m), dtype=torch.float32, device=dev)
	from stpy_amd import _lib as L

	wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m))          # the split W of the bf16-matrix-core route (see csrc/rff.hip)
	work = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)

	def c5(ws=True):
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(xr), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(z), m, 0,
								   L.ptr(work) if ws and wb else None, wb if ws else 0, L.stream_ptr()), "rff")
	rows = torch.cat([torch.arange(0, 64), torch.arange(n // 2, n // 2 + 64), torch.arange(n - 64, n)]).to(dev)
	from oracle import gp_oracle as O          # checker only
	ref = O.rff_embed(xr[rows].double().cpu().numpy(), W.double().cpu().numpy(), m)
	t32, _ = timed(lambda: c5(False), reps=4)          # the fp32-MFMA kernel (no workspace), for the record
	err32 = float(np.abs(z[rows].cpu().numpy() - ref).max() / math.sqrt(2.0 / m))
	t, _ = timed(c5, reps=5)
	err = float(np.abs(z[rows].cpu().numpy() - ref).max() / math.sqrt(2.0 / m))
	bytes_ = n * m * 4 + n * d * 4 + m * d * 4
	flops = 2.0 * n * d * m
	# the store-only ceiling of THIS buffer on THIS box: a plain device fill of the same 34.4 GB (the best case of a write stream:
	# contiguous, no arithmetic, no operand reads), timed the same way
	t_fill, _ = timed(lambda: z.fill_(1.0), reps=5)
	out["C5"] = {"workload": "RFF embed N=262144 d=64 m=32768 fp32 (stpy_rff_embed with its workspace, output resident)", "seconds": round(t, 5),
				 "bound": "hbm", "achieved": round(bytes_ / t / 1e12, 2), "peak": 8.0, "unit": "TB/s", "frac": round(bytes_ / t / 8e12, 4),
				 "store_ceiling": {"what": "device fill of the same output buffer (n x m fp32), same process", "seconds": round(t_fill, 5),
								   "TB/s": round(n * m * 4 / t_fill / 1e12, 2), "frac_of_it": round(t_fill / t, 4)},
				 "arithmetic": "fp32 in / fp32 out; contraction = six bf16 MFMA products of an exact 8+8+8-bit split of both operands, fp32 accumulation "
							   "(dropped terms < 2^-23 |x||w|); the fp32 MFMA shares the SIMD's ALUs with the trig work, the bf16 matrix pipe does not",
				 "contraction_tflops_fp32_equivalent": round(flops / t / 1e12, 1),
				 "fp32_mfma_kernel": {"seconds": round(t32, 5), "frac_of_fp32_mfma_peak": round(flops / t32 / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
									  "max_abs_err_over_amplitude": float("%.2e" % err32)},
				 "parity": {"max_abs_err_over_amplitude": float("%.2e" % err), "rows_checked": int(rows.numel()), "tolerance": 2e-5}}
	del work
	del z
	torch.cuda.empty_cache()

	# KF: KernelizedFeatures at BASELINE config 5's data shape with m = 8192 features, fp32: the normal
	# equations V = Phi^T Phi + s^2 lam I accumulated over row slabs (Phi never materialised: it would be 8.6 GB here, 34 GB at
	# m = 32 768), Cholesky of V, prediction at M = 4096 points.  Bound: fp32 MFMA on N m^2 (lower-triangular SYRK) + m^3/3 + M m^2.
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	n, d, m, M = 262144, 64, 8192, 4096
	gen = torch.Generator().manual_seed(1238)
	xk = torch.rand(n, d, generator=gen, dtype=torch.float32).to(dev)
	yk = torch.sin(xk[:, :4].sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=torch.Generator().manual_seed(1239), dtype=torch.float32).to(dev)
	xtk = torch.rand(M, d, generator=torch.Generator().manual_seed(1240), dtype=torch.float32).to(dev)
	np.random.seed(1238)
	embk = RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
	embk.W = embk.W.float()
	# (noise level 1: with 262 144 rows summed in fp32 the entries of Phi^T Phi carry ~1e-3 of absolute error, a perturbation of norm
	# ~0.1 of the m x m matrix -- s^2 lam = 0.09 sits at that edge, whichever kernel accumulates)
	s_kf = 1.0
	kf = KernelizedFeatures(embedding=embk, m=m, s=s_kf, lam=1.0, d=d)

	def kfit():
		kf.fit_gp(xk, yk)
		return kf.mean_std(xtk)
	peak0 = torch.cuda.max_memory_allocated()
	torch.cuda.reset_peak_memory_stats()
	t, (muk, sdk) = timed(kfit, reps=3)
	peak = torch.cuda.max_memory_allocated()
	# parity on a sub-problem the oracle finishes in a second: 8192 rows, 1024 features (the oracle's pseudo-inverse of an m x m
	# matrix is what costs: minutes at m = 8192), three slabs, against the one-shot normal equations
	ns, ms_ = 8192, 1024
	np.random.seed(1241)
	embs = RFFEmbedding(gamma=math.sqrt(d), m=ms_, d=d)
	embs.W = embs.W.float()
	Wk = embs.W.double().cpu().numpy()
	Q = O.rff_embed(xk[:ns].double().cpu().numpy(), Wk, ms_)
	_, invV, theta = O.kernelized_features_fit(Q, yk[:ns].double().cpu().numpy(), s_kf, 1.0)
	mu_o, sd_o = O.kernelized_features_mean_std(O.rff_embed(xtk[:256].double().cpu().numpy(), Wk, ms_), invV, theta, s_kf)
	kf2 = KernelizedFeatures(embedding=embs, m=ms_, s=s_kf, lam=1.0, d=d)
	kf2.slab_bytes = ms_ * 4 * 3072          # three slabs
	kf2.fit_gp(xk[:ns], yk[:ns])
	mu2, sd2 = kf2.mean_std(xtk[:256])
	relk = lambda a, b: float(np.linalg.norm(a.double().cpu().numpy() - b) / np.linalg.norm(b))
	Fk = float(n) * m * m + m ** 3 / 3.0 + float(M) * m * m + 2.0 * n * d * m
	out["KF"] = {"workload": "KernelizedFeatures.fit_gp + mean_std on RFF features: N=262144 d=64 m=8192 fp32, s=1, M=4096, Phi streamed in <= 2 GiB row slabs",
				 "seconds": round(t, 4), "bound": "mfma", "achieved": round(Fk / t / 1e12, 2), "peak": round(PEAK_BF16X3_TFLOPS, 1), "bound_pipe": "bf16 mfma x6 (exact 3-way split)", "unit": "TFLOP/s",
				 "frac": round(Fk / t / 1e12 / PEAK_BF16X3_TFLOPS, 4), "ratio_to_fp32_mfma_peak": round(Fk / t / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4), "arithmetic": BF16X3_NOTE,
				 "algorithmic_flop": "N m^2 (lower-triangular Phi^T Phi) + m^3/3 + M m^2 + 2 N d m (embed) = %.3e" % Fk,
				 "peak_device_bytes": int(peak), "phi_bytes_if_materialised": int(n) * m * 4,
				 "parity": {"sub_problem": "N=8192, m=1024 (three slabs) vs the oracle's one-shot normal equations, 256 test points",
							"mu": float("%.2e" % relk(mu2, mu_o)), "sigma": float("%.2e" % relk(sd2, sd_o)), "tolerance": 2e-3}}
	del kf, kf2, xk, yk
	torch.cuda.empty_cache()

	# GRAD: the evidence and its gradient (what Estimator.optimize_params_general evaluates per L-BFGS step,
	# estimator.py:156-190): log_marginal value + backward at N = 32 768, d = 16, fp64, for an isotropic SE lengthscale and for an
	# ARD kernel (16 lengthscales).  Algorithmic flop: N^3/3 (Cholesky) + 2 N^3/3 (K^-1 from the factor: triangular-aware solve on the
	# identity + SYRK) = N^3, + 2 N^2 d' for H [Xs | 1]; the Gram fills and the elementwise H are HBM work beside it.
	n, d = 32768, 16
	x, y, _ = synth(n, d, 8, dev)
	Fg = float(n) ** 3 + 2.0 * n * n * (d + 1)
	grad = {}
	for tag, make, leaf, key in (
			("SE", lambda: GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d), lambda: torch.tensor(4.0, dtype=torch.float64, requires_grad=True), "gamma"),
			("ARD", lambda: GaussianProcess(s=0.1, kernel=__import__("stpy_amd").KernelFunction(kernel_name="ard", ard_gamma=torch.full((d,), 4.0, dtype=torch.float64), d=d), d=d),
			 lambda: (torch.linspace(3.0, 5.0, d, dtype=torch.float64)).requires_grad_(True), "ard_gamma")):
		gpg = make()
		gpg.load_data((x, y))
		holder = {}

		def one():
			g = leaf()
			f = gpg.log_marginal(gpg.kernel_object, {'0': {key: g}}, 1.0)
			f.backward()
			holder["f"], holder["g"] = float(f.detach()), g.grad.detach().clone()
		t, _ = timed(one, reps=3)
		# parity in the run: central difference of the HIP value itself along the gradient direction (the pinned check against
		# autograd through the reference is golden G14 in tests/)
		g0 = leaf().detach()
		dirn = holder["g"].cpu() / holder["g"].cpu().norm()
		h = 1e-4
		fp = float(gpg.log_marginal(gpg.kernel_object, {'0': {key: (g0 + h * dirn.reshape(g0.shape))}}, 1.0))
		fm = float(gpg.log_marginal(gpg.kernel_object, {'0': {key: (g0 - h * dirn.reshape(g0.shape))}}, 1.0))
		fd = (fp - fm) / (2 * h)
		an = float((holder["g"].cpu().reshape(-1) * dirn.reshape(-1)).sum())
		grad[tag] = {"seconds": round(t, 4), "achieved": round(Fg / t / 1e12, 2), "frac": round(Fg / t / 1e12 / PEAK_FP64_MFMA_TFLOPS, 4),
					 "value": holder["f"], "directional_derivative": {"analytic": float("%.8e" % an), "central_difference": float("%.8e" % fd),
																		"rel_err": float("%.2e" % (abs(an - fd) / max(abs(fd), 1e-300)))}}
		del gpg
		torch.cuda.empty_cache()
	out["GRAD"] = {"workload": "GaussianProcess.log_marginal value + backward (analytic evidence gradient), N=32768 d=16 fp64: SE (1 lengthscale) and ARD (16)",
				   "bound": "mfma", "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "algorithmic_flop": "N^3/3 (potrf) + 2N^3/3 (potri) + 2 N^2 (d+1) = %.3e" % Fg,
				   "seconds": grad["SE"]["seconds"], "achieved": grad["SE"]["achieved"], "frac": grad["SE"]["frac"], "cases": grad,
				   "parity_note": "in-run: analytic directional derivative vs a central difference of the HIP value (h = 1e-4); pinned check vs autograd through the reference: tests/golden/G14"}
	return out


def F_fp(n, m):
	return flops_fit_predict(n, m)

METRIC_FMT = "GP fit+mean_var wall-time (s), N=%d d=%d fp64"
HEADLINE_FALLBACK_TAG = "STPY_BENCH_HEADLINE_BEFORE_EXTRA "
C4_ONE_GPU_SECONDS = 11.47      # BASELINE config 4's shape on ONE MI355X (tools/c4_single_gpu.py, end of round 2; re-measured by `--gpus 1` runs: extra_configs.C4.seconds)
C4_ONE_GPU_MEASURED_WITH = "round-2/3 builds on other boxes of the pool (11.35-11.60 s)"   # a multi-GPU run cannot re-measure it: the ratio below is labelled accordingly
WORKLOADS = {"headline": (65536, 16, 4096), "c4": (131072, 32, 4096), "c2": (16384, 8, 4096)}


def _error_line(msg, n_gpus, n=65536, d=16, tb=None, **extra):
	"""Every failure path ends in ONE JSON line with value null (not measured) and the reason."""
	out = {"metric": METRIC_FMT % (n, d), "value": None, "unit": "s", "n_gpus": n_gpus, "higher_is_better": False, "scaling": "strong",
		   "vs_baseline": None, "dtype": "f64", "data": "synthetic", "error": msg}
	if tb:
		out["traceback_tail"] = tb[-1500:]
	out.update(extra)
	return json.dumps(out)


def parse_args(argv=None):
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--config", choices=sorted(WORKLOADS), default="headline",
					help="headline: N=65536 d=16 (BASELINE metric); c4: N=131072 d=32 (BASELINE config 4's shape); c2: N=16384 d=8")
	ap.add_argument("--n", "--train-points", dest="n", type=int, default=0)
	ap.add_argument("--d", "--dims", dest="d", type=int, default=0)
	ap.add_argument("--m", "--test-points", dest="m", type=int, default=0)
	ap.add_argument("--nb", type=int, default=0)
	ap.add_argument("--nb-dist", type=int, default=0, help="distribution block of the block-cyclic path (0: the class default for the size)")
	ap.add_argument("--transport", choices=["collective", "fanout", "auto"], default=os.environ.get("STPY_DIST_TRANSPORT", "collective"),
					help="panel broadcasts of the block-cyclic path: RCCL broadcast (default) or point-to-point fan-out to every peer")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--cpu-baseline-long", action="store_true", help="extend the reference-shaped CPU sweep to N = 8192 (about a minute more of host time)")
	ap.add_argument("--no-extra-configs", action="store_true")
	args = ap.parse_args(argv)
	n0, d0, m0 = WORKLOADS[args.config]
	args.n, args.d, args.m = args.n or n0, args.d or d0, args.m or m0
	return args


def self_launch(args, argv):
	"""`python bench.py --gpus N` started plainly (WORLD_SIZE unset): start N fresh rank processes through torch.distributed.run,
	relay rank 0's JSON line and exit with the children's return code.  This process has made NO GPU call (importing torch does
	not initialise the runtime), and it never replaces itself: the ranks are children."""
	import signal
	import socket
	import subprocess
	import tempfile
	with socket.socket() as sk:
		sk.bind(("127.0.0.1", 0))
		port = sk.getsockname()[1]
	# (torch.distributed.run's argparse classifies EVERY argument up front, also those behind the script name: `--n` / `--m` / `--d`
	# would be rejected as ambiguous abbreviations of its own options -- the ranks get the long spellings)
	long_names = {"--n": "--train-points", "--m": "--test-points", "--d": "--dims"}
	argv = [long_names.get(a, a) for a in argv]
	argv = [long_names.get(a.split("=", 1)[0], a.split("=", 1)[0]) + "=" + a.split("=", 1)[1] if "=" in a and a.split("=", 1)[0] in long_names else a for a in argv]
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
		   "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
	env = dict(os.environ)
	env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this pool (RCCL across processes)
	env["STPY_BENCH_LAUNCHED_BY"] = "self"
	limit = float(os.environ.get("STPY_BENCH_TIMEOUT_S", "1700"))
	errf = tempfile.TemporaryFile(mode="w+")
	proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=errf, env=env, text=True, start_new_session=True)
	timed_out = False
	try:
		out, _ = proc.communicate(timeout=limit)
	except subprocess.TimeoutExpired:
		timed_out = True
		try:
			os.killpg(proc.pid, signal.SIGKILL)          # exactly the process group started above
		except ProcessLookupError:
			pass
		out, _ = proc.communicate()
	errf.seek(0)
	err = errf.read()
	sys.stderr.write(err[-20000:])
	line = None
	for ln in (out or "").splitlines():
		ln = ln.strip()
		if ln.startswith("{") and '"metric"' in ln:
			line = ln
	rc = proc.returncode if not timed_out else 124
	if line is None:
		# the ranks died after the headline was measured (inside the extra C4 workload): rank 0 left the finished line on stderr
		for ln in err.splitlines():
			if ln.startswith(HEADLINE_FALLBACK_TAG):
				try:
					rec = json.loads(ln[len(HEADLINE_FALLBACK_TAG):])
					rec.setdefault("extra_configs", {})["C4"] = {"workload": "N=131072 d=32", "seconds": None,
																 "error": "the rank processes ended (rc %s%s) during this extra workload, after the headline had been measured" % (rc, ", timed out" if timed_out else "")}
					line = json.dumps(rec)
					rc = 0
				except ValueError:
					pass
	if line is None:
		why = "timed out after %.0f s" % limit if timed_out else "rank processes ended with rc %s and printed no result line" % rc
		line = _error_line("self-launch of %d ranks: %s" % (args.gpus, why), args.gpus, args.n, args.d, tb=err, launcher="bench.py self-launch (torch.distributed.run)")
		rc = rc or 1
	print(line, flush=True)
	return rc


def run_steps(gp, x, y, xt, warmup, steps, dist_path, lib=None):
	"""`warmup` untimed steps, then exactly `steps` timed ones between barrier + synchronize; returns the elapsed seconds of this rank."""
	def barrier():
		if dist_path:
			torch.distributed.barrier()
		torch.cuda.synchronize()
	out = None
	for _ in range(warmup):
		gp.fit_gp(x, y)
		out = gp.mean_std(xt)
	barrier()
	if lib is not None:
		lib.stpy_profile_enable(1)
	t0 = time.perf_counter()
	for _ in range(steps):
		gp.fit_gp(x, y)
		out = gp.mean_std(xt)
	barrier()
	elapsed = time.perf_counter() - t0
	if lib is not None:
		lib.stpy_profile_enable(0)
	return elapsed, out


def max_over_ranks(elapsed, world, backend, dev):
	if world > 1:
		tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
		torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
		elapsed = float(tt.item())
	return elapsed


def traffic_from_profiles(lib_version, launches_per_step):
	"""HBM traffic of the dominant kernel family: PMC counters cannot be collected inside this process, so the value comes from
	the committed separate-pass rocprofv3 --pmc runs of this same command (profiles/*_pmc_traffic.json: FETCH_SIZE doubled per the
	gfx950 correction + WRITE_SIZE), per launch -- and ONLY when that file was recorded with this very build of the library
	(its `library_version`) and the same number of launches per step; otherwise null (stale numbers are not paired with this run)."""
	import glob
	info = {"traffic": None, "traffic_source": None}
	cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
	if not cands:
		info["traffic_note"] = "no profiles/*_pmc_traffic.json"
		return info
	src = cands[-1]
	try:
		with open(src) as fh:
			rec = json.load(fh)
	except Exception as exc:          # noqa: BLE001
		info["traffic_note"] = "unreadable %s: %s" % (os.path.basename(src), exc)
		return info
	info["traffic_source"] = "profiles/" + os.path.basename(src)
	info["traffic_source_gemm_launches_per_step"] = rec.get("gemm_launches")
	info["traffic_source_library_version"] = rec.get("library_version")
	if rec.get("library_version") != lib_version:
		info["traffic_note"] = "null: the profile was recorded with library build %r, this run is %r (re-run tools/profile_round.sh)" % (rec.get("library_version"), lib_version)
		return info
	if rec.get("gemm_launches") and abs(rec["gemm_launches"] - launches_per_step) > 0.02 * launches_per_step:
		info["traffic_note"] = "null: launches per step differ (profile %s, this run %.0f)" % (rec.get("gemm_launches"), launches_per_step)
		return info
	info["traffic"] = round(rec["per_launch_hbm_bytes"])
	info["traffic_note"] = "HBM bytes per launch of this kernel family from separate rocprofv3 --pmc passes of this command with this build, 2*FETCH_SIZE + WRITE_SIZE"
	return info


def main(args):
	world = int(os.environ.get("WORLD_SIZE", "1"))
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	if world != args.gpus:
		raise RuntimeError("--gpus %d but WORLD_SIZE=%d: start `python bench.py --gpus N` plainly (it launches its ranks itself) or through "
						   "torch.distributed.run --nproc-per-node N" % (args.gpus, world))
	# STPY_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box: the ranks share card 0 and the
	# collectives are staged through the host (functional check of this script only, not a measurement)
	backend = os.environ.get("STPY_BENCH_BACKEND", "nccl")
	if not torch.cuda.is_available():
		raise RuntimeError("no GPU visible to this process (the HIP path has no CPU fallback)")
	if backend != "nccl":
		local_rank = local_rank % max(torch.cuda.device_count(), 1)
	elif world > torch.cuda.device_count():
		raise RuntimeError("%d ranks on the nccl (RCCL) backend need %d GPUs, %d visible" % (world, world, torch.cuda.device_count()))
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
		import datetime
		# a collective that never completes must end the run with an error line, not sit in the driver's clock: 4-minute watchdog
		if backend == "nccl":
			dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(minutes=4))
		else:
			dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=4))

	from stpy_amd import GaussianProcess, _lib
	lib = _lib.load()
	lib_version = lib.stpy_version().decode()

	n, d, m = args.n, args.d, args.m
	s = 0.1

	def make_gp(dd, nn):
		if dist_path:
			from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
			kw = {"nb_dist": args.nb_dist} if args.nb_dist else {}
			kw["transport"] = args.transport
			g = DistributedGaussianProcess(gamma=math.sqrt(dd), s=s, kappa=1.0, kernel_name="squared_exponential", d=dd, force_path=force_dist, **kw)
		else:
			g = GaussianProcess(gamma=math.sqrt(dd), s=s, kappa=1.0, kernel_name="squared_exponential", d=dd)
		g.nb = args.nb
		return g

	# Multi-GPU: before anything is timed, the distributed path is checked on a small problem against the single-GPU class on this
	# very GPU (every rank holds the replicated result; a wrong collective order or a stale panel shows up here, not as a fast wrong number)
	selfcheck = None
	if dist_path:
		ns, ms = 4096 + 384, 256
		xs, ys, xts = synth(ns, d, ms, dev, seed=99)
		g1 = GaussianProcess(gamma=math.sqrt(d), s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
		g1.fit_gp(xs, ys)
		mu1, sd1 = g1.mean_std(xts)
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		gd = DistributedGaussianProcess(gamma=math.sqrt(d), s=s, kappa=1.0, kernel_name="squared_exponential", d=d, force_path=force_dist, nb_dist=256, transport=args.transport)
		gd.fit_gp(xs, ys)
		mud, sdd = gd.mean_std(xts)
		err = torch.stack([torch.norm(mud - mu1) / torch.norm(mu1), torch.norm(sdd - sd1) / torch.norm(sd1)])
		if world > 1:
			errh = err.cpu() if backend != "nccl" else err
			torch.distributed.all_reduce(errh, op=torch.distributed.ReduceOp.MAX)
			err = errh
		selfcheck = {"n": ns, "m": ms, "nb_dist": 256, "mu_rel_err_vs_single_gpu_class": float("%.2e" % float(err[0])), "sigma_rel_err": float("%.2e" % float(err[1])), "tolerance": 1e-8}
		if not (float(err[0]) < 1e-8 and float(err[1]) < 1e-8):
			raise RuntimeError("distributed self-check failed: %s" % json.dumps(selfcheck))
		del g1, gd, xs, ys, xts
		torch.cuda.empty_cache()

	x, y, xt = synth(n, d, m, dev)
	gp = make_gp(d, n)
	elapsed, (mu, std) = run_steps(gp, x, y, xt, args.warmup, args.steps, dist_path, lib)
	sec_per_step = max_over_ranks(elapsed, world, backend, dev) / args.steps

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

	tinfo = {"traffic": None, "traffic_source": None, "traffic_note": "null: only recorded for the single-GPU headline workload"}
	if (n, d, m) == WORKLOADS["headline"] and not dist_path:
		tinfo = traffic_from_profiles(lib_version, g_cnt / max(args.steps, 1))

	# ---- multi-GPU: what moved and how busy every rank's MFMA GEMM was (gathered; one small host object per rank)
	dist_info = None
	if dist_path:
		mine = {"rank": rank, "gemm_busy_ms_per_step": round(ub.value / args.steps, 2),
				"bcast_MB_per_step": round(gp.stats["bcast_bytes"] / 1e6 / (args.steps + args.warmup), 1),
				"reduce_MB_per_step": round(gp.stats["reduce_bytes"] / 1e6 / (args.steps + args.warmup), 1),
				"collectives_per_step": gp.stats["collectives"] // max(args.steps + args.warmup, 1)}
		gathered = [None] * world
		torch.distributed.all_gather_object(gathered, mine)
		dist_info = {"rccl_ranks": world if backend == "nccl" else 0, "backend": backend, "grid": "%dx%d" % (gp.Pr, gp.Pc), "nb_dist": gp.NB,
					 "transport": getattr(gp, "transport", "collective"), "col_exchange": getattr(gp, "col_exchange", None), "selfcheck": selfcheck, "per_rank": gathered}
	gp_nb = gp.NB if dist_path else None
	result_check = {"mu_norm": float(torch.norm(mu)), "std_mean": float(std.mean()), "nan": bool(torch.isnan(std).any())}
	del gp, mu, std, x, y, xt
	torch.cuda.empty_cache()

	# ---- the headline line is complete here (rank 0); what follows only ADDS to it
	out = None
	if rank == 0:
		F = flops_fit_predict(n, m)
		out = {
			"metric": METRIC_FMT % (n, d),
			"value": round(sec_per_step, 4), "unit": "s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(sec_per_step * 1e3, 2), "higher_is_better": False,
			"scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if backend == "nccl" or world == 1 else "synthetic (REHEARSAL: %s backend, ranks share one GPU)" % backend,
			"config": {"workload": "GaussianProcess.fit_gp + mean_std, SE kernel gamma=sqrt(d), s=0.1, N=%d train, M=%d test, d=%d, fp64" % (n, m, d),
					   "n": n, "m": m, "d": d, "nb": args.nb or ("distribution block %d" % gp_nb if dist_path else "potrf panels by size (library defaults), recursive block solve"),
					   "parallelism": "2-D block-cyclic over %d GPU%s" % (world, "s" if world > 1 else " (forced: functional check of the distributed code path)") if dist_path else "single GPU"},
			"step_tflops": round(F / sec_per_step / 1e12, 2),
			"step_frac_of_fp64_mfma_peak": round(F / sec_per_step / 1e12 / (PEAK_FP64_MFMA_TFLOPS * world), 4),
			"library": lib_version,
			"roofline": dict({"bound": "mfma", "kernel": "stpy::gemm_nt_dtv_kernel / gemm_nt_kernel<double>" if dist_path else "stpy::gemm_nt_dtv_kernel<SUB> (+ gemm_nt_kernel<double,...> for the ragged / fused-epilogue launches)", "achieved": round(achieved, 2), "peak": PEAK_FP64_MFMA_TFLOPS,
						 "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP64_MFMA_TFLOPS, 4)}, **tinfo,
						 **{"launches": int(g_cnt), "avg_launch_ms": round(g_ms / max(g_cnt, 1), 4), "busy_ms_per_step": round(ub.value / args.steps, 2),
						 "algorithmic_gflop_per_launch": round(g_fl / max(g_cnt, 1) / 1e9, 3)}),
			"breakdown_ms_per_step": {k: round(v[0] / args.steps, 2) for k, v in pr.items()},
			"result_check": result_check,
		}
		if dist_info is not None:
			out["multi_gpu"] = dist_info
		if world > 1:
			# the measured headline must survive whatever the extra workload below does to the process group (a rank that dies inside
			# a collective takes the others down with the watchdog): it is put on stderr now, where bench.py's own launcher -- and
			# anyone reading the log -- finds it if no line reaches stdout
			sys.stderr.write(HEADLINE_FALLBACK_TAG + json.dumps(out) + "\n")
			sys.stderr.flush()

	# ---- BASELINE config 4's shape (N = 131 072, d = 32): the strong-scaling target of north_star.  One GPU: the reference time
	# ---- of the curve; several: seconds, speed-up over one GPU and the fraction of the aggregate fp64 MFMA peak.
	c4 = None
	if args.config == "headline" and not args.no_extra_configs and (n, d, m) == WORKLOADS["headline"] and os.environ.get("STPY_BENCH_SKIP_C4", "0") != "1":
		n4, d4, m4 = WORKLOADS["c4"]
		x4 = y4 = xt4 = g4 = None
		ok, why = 1, ""
		try:          # phase A, no collective: inputs and the estimator object
			x4, y4, xt4 = synth(n4, d4, m4, dev)
			g4 = make_gp(d4, n4)
		except Exception as exc:          # noqa: BLE001
			ok, why = 0, "%s: %s" % (type(exc).__name__, exc)
		if world > 1:          # every rank must enter phase B or none: agree first (one tiny all-reduce, bounded by the group's watchdog)
			flag = torch.tensor([ok], dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
			torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
			if int(flag.item()) == 0 and ok:
				ok, why = 0, "another rank could not set the workload up"
		if ok:
			try:
				e4, (mu4, sd4) = run_steps(g4, x4, y4, xt4, 1, 1 if world == 1 else 2, dist_path)
				t4 = max_over_ranks(e4, world, backend, dev) / (1 if world == 1 else 2)
				F4 = flops_fit_predict(n4, m4)
				c4 = {"workload": "N=131072 d=32 SE fp64 fit_gp+mean_std, M=4096 (BASELINE config 4's shape)", "seconds": round(t4, 4), "n_gpus": world,
					  "bound": "mfma", "achieved": round(F4 / t4 / 1e12, 2), "peak": PEAK_FP64_MFMA_TFLOPS * world, "unit": "TFLOP/s",
					  "frac": round(F4 / t4 / 1e12 / (PEAK_FP64_MFMA_TFLOPS * world), 4),
					  "nb_dist": g4.NB if dist_path else None,
					  "result_check": {"mu_norm": float(torch.norm(mu4)), "std_mean": float(sd4.mean()), "nan": bool(torch.isnan(sd4).any())}}
				if world > 1:
					# the one-GPU time of the curve cannot be re-measured inside a multi-GPU run: the ratio is against a figure from ANOTHER
					# build and box and says so (the driver computes scaling itself from its own N = 1 run)
					c4["one_gpu_reference"] = {"seconds": C4_ONE_GPU_SECONDS, "measured_with": C4_ONE_GPU_MEASURED_WITH, "same_build_and_box": False}
					c4["speedup_vs_that_reference"] = round(C4_ONE_GPU_SECONDS / t4, 3)
				del mu4, sd4
			except Exception as exc:          # noqa: BLE001  (the headline line must survive a failure of the extra workload: it is measured already)
				c4 = {"workload": "N=131072 d=32", "seconds": None, "error": "%s: %s" % (type(exc).__name__, exc)}
		else:
			c4 = {"workload": "N=131072 d=32", "seconds": None, "error": why}
		del g4, x4, y4, xt4
		torch.cuda.empty_cache()

	if rank == 0:
		extra = {}
		if c4 is not None:
			extra["C4"] = c4
		if world == 1 and not args.no_extra_configs and not dist_path:
			extra.update(extra_configs(dev, lib))
		if extra:
			out["extra_configs"] = extra
		if world == 1 and not args.no_cpu_baseline and not dist_path:
			out["cpu_baseline"] = cpu_baseline(d, lambda: GaussianProcess(gamma=math.sqrt(d), s=s, kappa=1.0, kernel_name="squared_exponential", d=d), dev,
											   long_sweep=args.cpu_baseline_long)
		print(json.dumps(out), flush=True)
	if dist_path:
		torch.distributed.destroy_process_group()


if __name__ == "__main__":
	_args = parse_args()
	if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		sys.exit(self_launch(_args, sys.argv[1:]))
	try:
		main(_args)
	except BaseException as exc:                         # noqa: BLE001
		# a run that dies must still say why on the one line the driver reads (value null = not measured) -- any world size
		if isinstance(exc, SystemExit) and exc.code in (0, None):
			raise
		if int(os.environ.get("RANK", "0")) == 0:
			import traceback
			print(_error_line("%s: %s" % (type(exc).__name__, exc), int(os.environ.get("WORLD_SIZE", "1")), _args.n, _args.d, tb=traceback.format_exc()), flush=True)
		if isinstance(exc, KeyboardInterrupt):
			raise
		sys.exit(1)
