"""Trailing-update GEMM (n = 32768 lower) at several K for first-round stagger values of the direct-to-VGPR kernel
(stpy_tune key 0: 0 off, > 1 = cycles)."""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
n = 32768
C = torch.randn(n, n, dtype=torch.float64, device=dev)
vals = [int(v) for v in sys.argv[1:]] or [0, 20000, 40000, 80000, 160000]
for k in (256, 512, 1024, 2048):
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	res = {}
	for rnd in range(3):
		for v in vals:
			lib.stpy_tune(0, v)
			f = lambda: lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr())
			f(); torch.cuda.synchronize()
			ts = []
			for _ in range(3):
				torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
			res.setdefault(v, []).append(min(ts))
	fl = 2.0 * n * n * k * (0.5 + 64.0 / n)
	print("k %5d: " % k + "  ".join("[%d] %.3f ms %.1f TF" % (v, min(r) * 1e3, fl / min(r) / 1e12) for v, r in res.items()), flush=True)
lib.stpy_tune(0, 40000)
