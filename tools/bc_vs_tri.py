"""The trailing update C -= P P^T (lower part, K = 1024) through the two launch forms: stpy_gemm_nt(lower_only=1) -- the single-GPU
factorisation -- and stpy_gemm_nt_bc with a 1 x 1 grid and 1024-wide distribution blocks -- the block-cyclic path on one rank.
usage: python tools/bc_vs_tri.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
k = 1024
for n in (8192, 16384, 32768, 49152):
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.zeros(n, n, dtype=torch.float64, device=dev)
	def tri():
		L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
	def bc():
		L.check(lib.stpy_gemm_nt_bc(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1024, 1, 1, 0, 0, 0, 0, L.stream_ptr()), "gemm_bc")
	res = {}
	for name, f in (("tri", tri), ("bc", bc), ("tri", tri), ("bc", bc)):
		f(); torch.cuda.synchronize()
		ts = []
		for _ in range(3):
			torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
		res.setdefault(name, []).append(min(ts))
	fl = float(n) * n * k          # lower triangle: n^2 k
	print("n=%6d  lower_only %.3f ms (%.1f TF)   block-cyclic 1x1 %.3f ms (%.1f TF)   ratio %.3f" % (n, min(res["tri"]) * 1e3, fl / min(res["tri"]) / 1e12, min(res["bc"]) * 1e3, fl / min(res["bc"]) / 1e12, min(res["bc"]) / min(res["tri"])), flush=True)
	del P, C
	torch.cuda.empty_cache()
