"""fp32 NT products: the bf16-matrix-core kernel (exact three-way split, route key 26 > 0) against the fp32-MFMA kernels (26 = 0), one
process.  Lower-triangular trailing updates (n x n, K = 1024 / 2048), the block solve's rectangles, the feature-space SYRK.
usage: python tools/f32_gemm_bench.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")

def timed(fn, reps=4):
	fn(); torch.cuda.synchronize()
	best = 1e9
	for _ in range(reps):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
	return best

for (m, n, k, lower, mode, what) in [(32768, 32768, 1024, 1, 1, "trailing update n=32768 K=1024"), (32768, 32768, 2048, 1, 1, "trailing update n=32768 K=2048"),
									 (16384, 16384, 512, 1, 1, "trailing update n=16384 K=512"), (4096, 32768, 32768, 0, 1, "block solve M=4096"),
									 (8192, 8192, 61440, 1, 2, "Phi^T Phi m=8192 slab 61440"), (8192, 8192, 1024, 0, 0, "square 8192 K=1024")]:
	A = torch.randn(m, k, dtype=torch.float32, device=dev)
	B = A if m == n else torch.randn(n, k, dtype=torch.float32, device=dev)
	C = torch.randn(m, n, dtype=torch.float32, device=dev)
	fl = (float(m) * n * k) if lower else 2.0 * m * n * k
	res = []
	for route in (64, 0):
		lib.stpy_tune(26, route)
		t = timed(lambda: L.check(lib.stpy_gemm_nt(L.F32, m, n, k, L.ptr(A), k, L.ptr(B), k, L.ptr(C), n, mode, lower, L.stream_ptr()), "gemm"))
		res.append((t, fl / t / 1e12))
	lib.stpy_tune(26, 64)
	print("%-34s bf16x3 %8.3f ms %6.1f TF | fp32-MFMA %8.3f ms %6.1f TF | x%.2f" % (what, res[0][0] * 1e3, res[0][1], res[1][0] * 1e3, res[1][1], res[1][0] / res[0][0]), flush=True)
	del A, B, C
