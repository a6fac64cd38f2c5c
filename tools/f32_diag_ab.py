"""fp32 factorisation with the diagonal blocks through the flow-form kernel (fp64 arithmetic inside the block; lab knob 29 = 1, the default)
against the fp32 form of the older kernel (29 = 2): error of L and of L L^T against an fp64 factorisation of the same fp32 matrix, for
several sizes and condition numbers.   usage: STPY_HIP_LIB=lab python tools/f32_diag_ab.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
for n, s in ((257, 0.3), (257, 0.03), (1000, 0.3), (1000, 0.05), (3000, 0.3), (3000, 0.05), (8192, 0.1)):
	rng = np.random.RandomState(n)
	x = rng.uniform(-1, 1, size=(n, 8))
	G = (x @ x.T + 1.0) ** 2 + s * s * np.eye(n)          # polynomial kernel: wide dynamic range
	G32 = G.astype(np.float32)
	ref = np.linalg.cholesky(G32.astype(np.float64))
	cond = np.linalg.cond(G32.astype(np.float64)) if n <= 3000 else float("nan")
	out = {}
	for knob in (2, 1):
		lib.stpy_tune(29, knob)
		K = torch.from_numpy(G32).to(dev)
		winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float32, device=dev)
		work = torch.empty(max(int(lib.stpy_potrf_workspace_bytes(L.F32, n, 0)), 16), dtype=torch.uint8, device=dev)
		info = torch.zeros(1, dtype=torch.int32, device=dev)
		L.check(lib.stpy_potrf(L.F32, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
		torch.cuda.synchronize()
		Lg = np.tril(K.cpu().numpy().astype(np.float64))
		out[knob] = (int(info.item()), np.linalg.norm(Lg - ref) / np.linalg.norm(ref), np.linalg.norm(Lg @ Lg.T - G32) / np.linalg.norm(G32))
	lib.stpy_tune(29, 1)
	print("n=%5d s=%.2f cond %.1e   fp32 block kernel: info %d  |L-Lref|/|Lref| %.2e  |LL^T-K|/|K| %.2e     fp64-inside flow kernel: info %d  %.2e  %.2e"
	      % (n, s, cond, out[2][0], out[2][1], out[2][2], out[1][0], out[1][1], out[1][2]), flush=True)
