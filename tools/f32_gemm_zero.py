"""Is the bf16x3 fp32 GEMM held back by stalls or by the clock the chip sustains under its load?  The same trailing update
(n = 32768 lower, K = 1024 and 2048) with all-zero operands (no toggling in the multipliers: the clock stays up), constant operands and
random operands; the fp32-MFMA kernel beside it.   usage: python tools/f32_gemm_zero.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")

def timed(fn, reps=6):
	fn(); torch.cuda.synchronize()
	best = 1e9
	for _ in range(reps):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
	return best

n = 32768
C = torch.zeros(n, n, dtype=torch.float32, device=dev)
for k in (1024, 2048):
	for what, make in (("zeros", lambda: torch.zeros(n, k, dtype=torch.float32, device=dev)), ("ones", lambda: torch.ones(n, k, dtype=torch.float32, device=dev)),
					   ("random", lambda: torch.randn(n, k, dtype=torch.float32, device=dev))):
		A = make()
		res = []
		for route in (64, 0):
			lib.stpy_tune(26, route)
			t = timed(lambda: L.check(lib.stpy_gemm_nt(L.F32, n, n, k, L.ptr(A), k, L.ptr(A), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm"))
			res.append(float(n) * n * k / t / 1e12)
		lib.stpy_tune(26, 64)
		print("K=%d %-7s bf16x3 %6.1f TF | fp32-MFMA %6.1f TF" % (k, what, res[0], res[1]), flush=True)
		del A
