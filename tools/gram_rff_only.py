"""Three launches each of the two HBM-write-side kernels of the path -- the Gram fill (N = 65 536, d = 16, SE, lower triangle,
fp64: the MFMA contraction with the fused kernel-function epilogue) and the fp32 RFF embed of BASELINE config 5
(N = 262 144, d = 64, m = 32 768: rff_stream_bf16x3_kernel, the contraction on the bf16 matrix cores) -- as a target for `rocprofv3 --pmc` / `--kernel-trace`.
usage: python tools/gram_rff_only.py [gram] [rff]"""
import math
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
which = sys.argv[1:] or ["gram", "rff"]
if "gram" in which:
	n, d = 65536, 16
	x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
	il = torch.full((d,), 0.25, dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
	for _ in range(3):
		L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
	torch.cuda.synchronize()
	del K
	torch.cuda.empty_cache()
if "rff" in which:
	n, d, m = 262144, 64, 32768
	g = torch.Generator().manual_seed(1237)
	xr = torch.rand(n, d, generator=g, dtype=torch.float32).to(dev)
	W = (torch.randn(m, d, generator=g, dtype=torch.float32) / 8.0).to(dev)
	z = torch.empty((n, m), dtype=torch.float32, device=dev)
	wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m))          # with the workspace: rff_split_w_kernel + rff_stream_bf16x3_kernel
	work = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
	for _ in range(3):
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(xr), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(z), m, 0, L.ptr(work), wb, L.stream_ptr()), "rff")
	torch.cuda.synchronize()
print("done")
