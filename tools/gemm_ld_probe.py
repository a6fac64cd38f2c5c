"""Trailing-update-shaped products (C -= P P^T, lower tiles) with the operand panel's leading dimension padded: does the power-of-two
row stride of the panel workspace (nb * 8 bytes) cost the direct-to-VGPR kernel anything?
usage: python tools/gemm_ld_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")

for (n, k) in ((32768, 1024), (32768, 2048), (61440, 2048), (16384, 512)):
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	for pad in (0, 16, 32, 48, 80, 272):
		Pb = torch.randn(n, k + pad, dtype=torch.float64, device=dev) * 0.01
		P = Pb[:, :k]
		ts = []
		for rnd in range(4):
			torch.cuda.synchronize(); t0 = time.perf_counter()
			L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), P.stride(0), L.ptr(P), P.stride(0), L.ptr(C), C.stride(0), 1, 1, L.stream_ptr()), "gemm_nt")
			torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
		t = min(ts[1:])
		print("update n=%d K=%d ld = K + %3d: %.3f ms = %.2f TFLOP/s" % (n, k, pad, t * 1e3, n * (n + 128) * k / t / 1e12), flush=True)
		del Pb, P
	del C
