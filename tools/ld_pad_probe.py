"""Do the write-bound kernels lose bandwidth to the power-of-two row stride of their output?  Same launches with the output's
leading dimension padded by a few hundred bytes (the kernels take ldo): RFF config 5 (row stride 128 KiB) and the fp64 Gram fill
at N = 65 536 (row stride 512 KiB).
usage: python tools/ld_pad_probe.py"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")


def timed(fn, reps=5):
	fn(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
	return min(ts)


n, d, m = 262144, 64, 32768
xr = torch.rand(n, d, dtype=torch.float32, device=dev)
W = (torch.randn(m, d, dtype=torch.float32, device=dev) / 8.0)
wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m))
work = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
for pad in (0, 32, 64, 128, 256, 1024, 4096 + 64):
	z = torch.empty((n, m + pad), dtype=torch.float32, device=dev)
	t = timed(lambda: L.check(lib.stpy_rff_embed(L.F32, L.ptr(xr), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(z), m + pad, 0, L.ptr(work), wb, L.stream_ptr()), "rff"))
	print("rff C5 ldo = m + %4d floats: %.3f ms = %.2f TB/s" % (pad, t * 1e3, n * m * 4 / t / 1e12), flush=True)
	del z
del xr, W

n, d = 65536, 16
x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
il = torch.full((d,), 0.25, dtype=torch.float64, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
for pad in (0, 16, 32, 64, 128, 512, 2048 + 32):
	K = torch.empty(n, n + pad, dtype=torch.float64, device=dev)
	for lower in (1, 0):
		t = timed(lambda: L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, lower, 0, L.ptr(K), n + pad, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram"))
		tiles = (n // 128) * (n // 128 + 1) // 2 if lower else (n // 128) ** 2
		by = tiles * 128 * 128 * 8
		print("gram SE %s ldo = n + %4d doubles: %.3f ms = %.2f TB/s" % ("lower" if lower else "full ", pad, t * 1e3, by / t / 1e12), flush=True)
	del K
