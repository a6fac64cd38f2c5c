"""Reference point only: the vendor library's fp64 GEMM (torch.matmul -> rocBLAS / hipBLASLt) on the same
box, next to stpy_gemm_nt on the same shapes.  Not used by the product."""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")

def timeit(f, reps=5):
	f(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record(); f(); e1.record(); torch.cuda.synchronize()
		ts.append(e0.elapsed_time(e1) * 1e-3)
	return min(ts)

for (m, n, k) in ((8192, 8192, 8192), (16384, 16384, 4096), (32768, 32768, 1024), (16384, 16384, 16384)):
	A = torch.randn(m, k, dtype=torch.float64, device=dev)
	B = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.zeros(m, n, dtype=torch.float64, device=dev)
	t_v = timeit(lambda: torch.matmul(A, B.t(), out=C))
	t_s = timeit(lambda: L.check(lib.stpy_gemm_nt(L.F64, m, n, k, L.ptr(A), k, L.ptr(B), k, L.ptr(C), n, 0, 0, L.stream_ptr()), "gemm"))
	fl = 2.0 * m * n * k
	print("m=%d n=%d k=%d: vendor %.2f ms %.1f TF | stpy_gemm_nt %.2f ms %.1f TF" % (m, n, k, t_v * 1e3, fl / t_v / 1e12, t_s * 1e3, fl / t_s / 1e12), flush=True)
	del A, B, C
