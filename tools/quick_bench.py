"""Scratch micro-benchmarks of the C-ABI entry points (GPU box only)."""
import collections
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")


def timeit(fn, reps=3, warm=1):
	for _ in range(warm):
		fn()
	torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		fn()
		e1.record()
		torch.cuda.synchronize()
		ts.append(e0.elapsed_time(e1) * 1e-3)
	return min(ts), float(np.median(ts))


def bench_gemm(n, k, tri):
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, tri, L.stream_ptr()), "gemm")
	t, tm = timeit(f)
	flops = 2.0 * n * n * k * (0.5 + 64.0 / n if tri else 1.0)
	print("gemm_nt n=%d k=%d tri=%d: %.3f ms  %.1f TF/s (min)  median %.3f ms" % (n, k, tri, t * 1e3, flops / t / 1e12, tm * 1e3), flush=True)


def bench_potrf(n, nb):
	x = torch.rand(n, 16, dtype=torch.float64, device=dev) * 2 - 1
	il = torch.full((16,), 0.25, dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
	work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, nb)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, 16)), dtype=torch.uint8, device=dev)
	gram0 = lambda: L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 16, L.ptr(x), n, 16, 16, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, None, 0, L.stream_ptr()), "gram")
	gram = lambda: L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 16, L.ptr(x), n, 16, 16, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel() * ws.element_size(), L.stream_ptr()), "gram")
	t0, _ = timeit(gram0)
	t, _ = timeit(gram)
	print("gram lower n=%d: tile kernel %.3f ms %.2f TB/s | MFMA+epilogue %.3f ms  %.2f TB/s" % (n, t0 * 1e3, n * n * 8 * 0.5 / t0 / 1e12, t * 1e3, n * n * 8 * 0.5 / t / 1e12), flush=True)
	def f():
		gram()
		L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel() * work.element_size(), nb, 0, L.ptr(info), L.stream_ptr()), "potrf")
	t2, _ = timeit(f, reps=2)
	tp = t2 - t
	print("potrf n=%d nb=%d: %.3f ms  %.1f TF/s  info=%d" % (n, nb, tp * 1e3, n ** 3 / 3.0 / tp / 1e12, int(info.item())), flush=True)


def ab_gemm():
	"""interleaved A/B of the first-round stagger (one process, same buffers)"""
	for n, k in ((32768, 512), (32768, 256), (32768, 1024), (16384, 512), (49152, 512)):
		P = torch.randn(n, k, dtype=torch.float64, device=dev)
		C = torch.randn(n, n, dtype=torch.float64, device=dev)
		f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
		res = {0: [], 1: []}
		for rnd in range(4):
			for v in (0, 1):
				lib.stpy_tune(0, v)
				res[v].append(timeit(f, reps=2, warm=1)[0])
		flops = float(n) * n * k
		print("n=%d k=%d  stagger off: %.3f ms %.1f TF | on: %.3f ms %.1f TF" % (n, k, min(res[0]) * 1e3, flops / min(res[0]) / 1e12, min(res[1]) * 1e3, flops / min(res[1]) / 1e12), flush=True)
		del P, C


def exp_gemm():
	"""timing-only ablations of the GEMM main loop (results are wrong for exp != 0)"""
	n, k = 32768, 1024
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
	names = collections.defaultdict(str)
	names.update({5: "no loads after tile 0, no barrier", 8: "no buffer flip (reads one LDS buffer)", 13: "no loads, no barrier, no flip"})
	names.update({0: "baseline", 1: "no global loads after tile 0", 3: "no loads, no LDS stores", 7: "no loads/stores/barrier", 15: "same + no buffer flip",
			 2: "loads kept, no LDS stores", 6: "loads kept, no stores, no barrier", 4: "no barrier only (racy)"})
	for rnd in range(2):
		for e in (0, 1, 4, 5):
			lib.stpy_tune(1, e)
			t = timeit(f, reps=2, warm=1)[0]
			print("exp=%2d %-36s %.3f ms  %.1f TF" % (e, names[e], t * 1e3, float(n) * n * k / t / 1e12), flush=True)
	lib.stpy_tune(1, 0)


def ab_fit(key, values, sizes=((16384, 8), (65536, 16))):
	"""interleaved A/B of a tune knob on the whole fit+mean_std step (one process)"""
	sys.path.insert(0, ".")
	from bench import synth
	from stpy_amd import GaussianProcess
	import math
	for n, d in sizes:
		x, y, xt = synth(n, d, 4096, dev)
		gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)
		def step():
			gp.fit_gp(x, y)
			return gp.mean_std(xt)
		res = {v: [] for v in values}
		for rnd in range(3):
			for v in values:
				lib.stpy_tune(key, v)
				res[v].append(timeit(step, reps=1, warm=1 if rnd == 0 else 0)[0])
		print("n=%d tune(%d): " % (n, key) + "  ".join("%d -> %.4f s" % (v, min(res[v])) for v in values), flush=True)
		del gp, x, y, xt
		torch.cuda.empty_cache()


def small_k():
	n = 32768
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	for k in (16, 32, 64, 128, 256, 512):
		P = torch.randn(n, k, dtype=torch.float64, device=dev)
		for mode in (1, 0):
			f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, mode, 1, L.stream_ptr()), "gemm")
			t = timeit(f, reps=3, warm=1)[0]
			print("n=%d k=%4d mode=%d: %.3f ms  (%.2f us per round of 512 tiles)  C-bytes/time = %.2f TB/s" % (n, k, mode, t * 1e3, t * 1e6 / (32896 / 512.0), (2 if mode else 1) * n * n * 4 / t / 1e12), flush=True)


def fixed_cost():
	"""t(k) per round of 512 tiles for the trailing-update shape: separates the per-tile fixed cost from the K slope"""
	n = 32768
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	rounds = (n // 128) * (n // 128 + 1) / 2 / 512.0
	for mode in (1, 0):
		pts = []
		for k in (256, 512, 1024, 2048, 4096):
			P = torch.randn(n, k, dtype=torch.float64, device=dev)
			f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, mode, 1, L.stream_ptr()), "gemm")
			t = timeit(f, reps=3, warm=1)[0]
			pts.append((k, t * 1e6 / rounds))
			print("mode=%d k=%4d: %.3f ms  %.1f us/round  %.1f TF" % (mode, k, t * 1e3, t * 1e6 / rounds, float(n) * (n + 128) * k / t / 1e12), flush=True)
			del P
		(k0, t0), (k1, t1) = pts[2], pts[4]
		b = (t1 - t0) / (k1 - k0)
		print("mode=%d: slope %.4f us per k per round (= %.1f TF), fixed %.1f us per round" % (mode, b, 512 * 2.0 * 128 * 128 / b / 1e6, t0 - b * k0), flush=True)


def rect():
	"""the shapes of the block solve: C (m x n) -= A (m x k) B^T, m = 4096"""
	for m, n, k in ((4096, 65536, 512), (4096, 32768, 512), (4096, 65536, 1024), (8192, 32768, 512), (4096, 65536, 256)):
		A = torch.randn(m, k, dtype=torch.float64, device=dev)
		B = torch.randn(n, k, dtype=torch.float64, device=dev)
		C = torch.randn(m, n, dtype=torch.float64, device=dev)
		f = lambda: L.check(lib.stpy_gemm_nt(L.F64, m, n, k, L.ptr(A), k, L.ptr(B), k, L.ptr(C), n, 1, 0, L.stream_ptr()), "gemm")
		t = timeit(f, reps=3, warm=1)[0]
		print("rect m=%d n=%d k=%d: %.3f ms  %.1f TF" % (m, n, k, t * 1e3, 2.0 * m * n * k / t / 1e12), flush=True)
		del A, B, C


def leftlook():
	"""the shapes a LEFT-looking factorisation / solve would give: few column tiles, long K"""
	for m, n, k, lda in ((57344, 1024, 8192, 65536), (32768, 1024, 32768, 65536), (16384, 1024, 49152, 65536), (8192, 1024, 57344, 65536),
						 (4096, 1024, 61440, 65536), (4096, 1024, 32768, 65536), (4096, 512, 32768, 65536), (32768, 512, 32768, 65536),
						 (32768, 2048, 32768, 65536)):
		A = torch.randn(m, lda, dtype=torch.float64, device=dev)
		B = torch.randn(n, lda, dtype=torch.float64, device=dev)
		C = torch.randn(m, n, dtype=torch.float64, device=dev)
		f = lambda: L.check(lib.stpy_gemm_nt(L.F64, m, n, k, L.ptr(A), lda, L.ptr(B), lda, L.ptr(C), n, 1, 0, L.stream_ptr()), "gemm")
		t = timeit(f, reps=3, warm=1)[0]
		msg = "left m=%d n=%d k=%d: plain %.3f ms %.1f TF" % (m, n, k, t * 1e3, 2.0 * m * n * k / t / 1e12)
		tiles = (m // 128) * (n // 128)
		for passes in (2, 4):
			if tiles * passes > 1024 or tiles >= 512:
				continue
			W = torch.empty(passes * m * n, dtype=torch.float64, device=dev)
			f2 = lambda: L.check(lib.stpy_gemm_nt_splitk(L.F64, m, n, k, L.ptr(A), lda, L.ptr(B), lda, L.ptr(C), n, 1, passes, L.ptr(W), W.numel() * W.element_size(), L.stream_ptr()), "gemm")
			t2 = timeit(f2, reps=3, warm=1)[0]
			msg += " | %d passes %.3f ms %.1f TF" % (passes, t2 * 1e3, 2.0 * m * n * k / t2 / 1e12)
			del W
		print(msg, flush=True)
		del A, B, C


if __name__ == "__main__":
	which = sys.argv[1] if len(sys.argv) > 1 else "all"
	if which == "dtv":
		# A/B of the direct-to-VGPR GEMM (stpy_tune key 6) on GEMM shapes, then on the whole step
		n = 32768
		C0 = torch.randn(n, n, dtype=torch.float64, device=dev)
		for k in (256, 1024, 4096):
			P = torch.randn(n, k, dtype=torch.float64, device=dev)
			outs = {}
			for rnd in range(2):
				for v in (0, 1):
					lib.stpy_tune(6, v)
					C = C0.clone()
					f = lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
					f(); torch.cuda.synchronize()
					outs[v] = C[:2048, :2048].clone()
					t = timeit(f, reps=3, warm=0)[0]
					print("tri update n=%d k=%d dtv=%d: %.3f ms  %.2f TF" % (n, k, v, t * 1e3, float(n) * (n + 128) * k / t / 1e12), flush=True)
			print("  max |diff| %.2e" % float((outs[0] - outs[1]).abs().max()), flush=True)
			del P
		del C0
		torch.cuda.empty_cache()
		ab_fit(6, [0, 1, 1024, 8192])
		lib.stpy_tune(6, 1024)
	if which == "diagfirst":
		for n in (8192, 16384, 32768, 65536):
			for thr in (1, 1 << 30, 1, 1 << 30):
				lib.stpy_tune(7, thr)
				print("diag-first below %d:" % thr, end=" ")
				bench_potrf(n, 0)
		lib.stpy_tune(7, 8192)
	if which == "fixed":
		fixed_cost()
	if which == "leftlook":
		leftlook()
	if which == "rect":
		rect()
	if which == "smallk":
		small_k()
	if which == "abfit":
		ab_fit(int(sys.argv[2]), [int(v) for v in sys.argv[3].split(",")])
	if which == "exp":
		exp_gemm()
	if which == "ab":
		ab_gemm()
	if which in ("all", "gemm"):
		bench_gemm(8192, 512, 0)
		bench_gemm(16384, 512, 1)
		bench_gemm(32768, 512, 1)
		bench_gemm(32768, 256, 1)
		bench_gemm(32768, 1024, 1)
	if which in ("all", "potrf"):
		for n, nb in ((8192, 512), (16384, 512), (32768, 512), (32768, 256), (32768, 1024)):
			bench_potrf(n, nb)
	if which == "big":
		for nb in (0, 1024, 0, 1024):
			bench_potrf(65536, nb)
		for nb in (0, 512, 0, 512):
			bench_potrf(32768, nb)
