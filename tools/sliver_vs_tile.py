"""fp64 lower-triangular updates C -= P P^T (n x n, K = panel width) through stpy_gemm_nt: the 32 x 128 sliver kernel (route key 30
large) against the 128 x 128 tile kernels (30 = 0), one process.   usage: python tools/sliver_vs_tile.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")

def timed(fn, reps=5):
	fn(); torch.cuda.synchronize()
	best = 1e9
	for _ in range(reps):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
	return best

for n in (1024, 2048, 4096, 8192, 12288, 16384, 24576):
	for k in (128, 256, 512, 1024, 2048):
		P = torch.randn(n, k, dtype=torch.float64, device=dev)
		C = torch.randn(n, n, dtype=torch.float64, device=dev)
		res = []
		for route in (1 << 30, 0):
			lib.stpy_tune(30, route)
			res.append(timed(lambda: L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")))
		tiles = (n // 128) * (n // 128 + 1) // 2
		fl = float(n) * n * k * (1 + 128.0 / n)
		print("n=%6d K=%5d tiles=%6d  sliver %9.3f ms %5.1f TF | tile %9.3f ms %5.1f TF | tile/sliver %.2f" % (n, k, tiles, res[0] * 1e3, fl / res[0] / 1e12, res[1] * 1e3, fl / res[1] / 1e12, res[1] / res[0]), flush=True)
		del P, C
lib.stpy_tune(30, 3200)
