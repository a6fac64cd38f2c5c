"""Gram fill alone (stpy_gram, fp64): time and HBM write rate of the full and the lower-only fill at N = 32 768 / 65 536, d = 16,
per kernel family, and the largest relative difference from torch's own exp on a 2048 x 2048 corner (SE).
usage: python tools/gram_bench.py"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
KINDS = {"SE": 0, "Matern-3/2": 2, "Matern-5/2": 3}
if len(sys.argv) > 1:
	lib.stpy_tune(1, int(sys.argv[1]))          # timing ablations of the GEMM kernel (16: the fill without its stores)
for n in (32768, 65536):
	d = 16
	x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
	il = torch.full((d,), 0.25, dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
	for name, kind in KINDS.items():
		for lower in (0, 1):
			ts = []
			for _ in range(4):
				torch.cuda.synchronize(); t0 = time.perf_counter()
				L.check(lib.stpy_gram(kind, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, lower, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
				torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
			by = n * n * 8.0 * (0.5 + 64.0 / n if lower else 1.0)
			print("N=%d %-10s %s: %.3f ms  %.2f TB/s of tile bytes" % (n, name, "lower" if lower else "full ", min(ts) * 1e3, by / min(ts) / 1e12), flush=True)
		if name == "SE":
			xs = x[:2048] * il
			ref = torch.exp(-0.5 * torch.cdist(xs, xs).pow(2))
			sq = (xs * xs).sum(1)
			ref2 = torch.exp(-0.5 * (sq[:, None] + sq[None, :] - 2 * xs @ xs.T))
			got = K[:2048, :2048].clone(); got.diagonal().sub_(0.01)
			print("   SE corner vs torch.exp of the same norm expansion: max rel diff %.2e (cdist form %.2e)" % (float(((got - ref2).abs() / ref2).max()), float(((got - ref).abs() / ref).max())))
	del K, x
	torch.cuda.empty_cache()
