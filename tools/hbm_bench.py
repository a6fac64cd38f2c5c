"""The HBM-bound kernels of the path at the headline shapes, one process: ms and TB/s of ALGORITHMIC bytes.
  predict  M = 4096 rows of X (N = 65 536)         M N w bytes read
  trsv     forward / backward, n = 65 536          n^2 w / 2 bytes read
  gram     lower-only and full SE fill, N = 65 536  bytes written
  rff      BASELINE config 5 (with workspace)      n m 4 bytes written
usage: python tools/hbm_bench.py [predict] [trsv] [gram] [rff]"""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
what = set(sys.argv[1:]) or {"predict", "trsv", "gram", "rff"}


def timed(fn, reps=5):
	fn(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
	return min(ts)


if "predict" in what:
	for dt, w in ((torch.float64, 8), (torch.float32, 4)):
		M, N = 4096, 65536
		X = torch.randn(M, N, dtype=dt, device=dev); z = torch.randn(N, dtype=dt, device=dev); kd = torch.full((M,), 1e9, dtype=dt, device=dev)
		mu = torch.empty(M, dtype=dt, device=dev); sg = torch.empty(M, dtype=dt, device=dev)
		code = L.dtype_code(dt)
		t = timed(lambda: L.check(lib.stpy_predict(code, M, N, L.ptr(X), N, L.ptr(z), L.ptr(kd), L.ptr(mu), L.ptr(sg), 0, L.stream_ptr()), "predict"))
		ref = (X.double() @ z.double())
		print("predict %s M=%d N=%d: %.3f ms = %.2f TB/s   (mu rel err %.1e)" % (str(dt)[6:], M, N, t * 1e3, M * N * w / t / 1e12, float((mu.double() - ref).norm() / ref.norm())), flush=True)
		del X

if "trsv" in what:
	n = 65536
	Lm = torch.randn(n, n, dtype=torch.float64, device=dev) * 1e-3
	Lm.diagonal().fill_(1.0)
	winv = torch.zeros(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
	winv.view(-1, 128, 128)[:, torch.arange(128), torch.arange(128)] = 1.0          # (not the true inverse blocks: timing only)
	y = torch.randn(n, dtype=torch.float64, device=dev); out = torch.empty_like(y)
	for trans in (0, 1):
		def run():
			yy = y.clone()
			L.check(lib.stpy_trsv(L.F64, n, L.ptr(Lm), n, L.ptr(winv), winv.numel(), L.ptr(yy), L.ptr(out), trans, L.stream_ptr()), "trsv")
		t = timed(run)
		print("trsv %s n=%d: %.3f ms = %.2f TB/s" % ("bwd" if trans else "fwd", n, t * 1e3, n * n * 8 / 2 / t / 1e12), flush=True)
	del Lm

if "gram" in what:
	n, d = 65536, 16
	x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
	il = torch.full((d,), 0.25, dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
	for kind, name in ((0, "SE"), (3, "Matern52")):
		for lower in (1, 0):
			res = []
			for route in (1, 0):          # route key 28: the dedicated fill kernel / the GEMM epilogue
				lib.stpy_tune(28, route)
				t = timed(lambda: L.check(lib.stpy_gram(kind, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, lower, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram"))
				res.append(t)
			lib.stpy_tune(28, 1)
			tiles = (n // 128) * (n // 128 + 1) // 2 if lower else (n // 128) ** 2
			by = tiles * 128 * 128 * 8
			print("gram %s %s N=%d: dedicated %.3f ms = %.2f TB/s | GEMM epilogue %.3f ms = %.2f TB/s" % (name, "lower" if lower else "full ", n, res[0] * 1e3, by / res[0] / 1e12, res[1] * 1e3, by / res[1] / 1e12), flush=True)
	del K

if "rff" in what:
	n, d, m = 262144, 64, 32768
	xr = torch.rand(n, d, dtype=torch.float32, device=dev)
	W = (torch.randn(m, d, dtype=torch.float32, device=dev) / 8.0)
	z = torch.empty((n, m), dtype=torch.float32, device=dev)
	wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m))
	work = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
	t = timed(lambda: L.check(lib.stpy_rff_embed(L.F32, L.ptr(xr), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(z), m, 0, L.ptr(work), wb, L.stream_ptr()), "rff"))
	print("rff C5: %.3f ms = %.2f TB/s" % (t * 1e3, n * m * 4 / t / 1e12), flush=True)
