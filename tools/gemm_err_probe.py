"""Error distribution of stpy_gemm_nt (fp64) on the shape one fuzz run flagged (m = n = 640, k = 2048, lower-only, padded leading
dimensions), against a float128 CPU product -- separates the kernel's rounding from the reference's (torch fp64 matmul on the GPU).
usage: python tools/gemm_err_probe.py [repeats]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m = n = 640; k = 2048
for same in (False, True):
	for rep in range(reps):
		torch.manual_seed(100 * rep + same)
		pa, pb, pc = 36, 2, 36
		A = torch.full((m, k + pa), float("nan"), dtype=torch.float64, device=dev); A[:, :k] = torch.randn(m, k, dtype=torch.float64, device=dev)
		if same:
			B = A; pb = pa
		else:
			B = torch.full((n, k + pb), float("nan"), dtype=torch.float64, device=dev); B[:, :k] = torch.randn(n, k, dtype=torch.float64, device=dev)
		C = torch.full((m, n + pc), 7.0, dtype=torch.float64, device=dev)
		rc = lib.stpy_gemm_nt(L.F64, m, n, k, L.ptr(A), k + pa, L.ptr(B), k + pb, L.ptr(C), n + pc, 0, 1, L.stream_ptr())
		assert rc == 0
		torch.cuda.synchronize()
		a = A[:, :k].cpu().numpy(); b = B[:, :k].cpu().numpy()
		rows = np.arange(0, m, 5)
		exact = (a[rows].astype(np.longdouble) @ b.astype(np.longdouble).T)
		scale = np.abs(a[rows]) @ np.abs(b).T
		got = C[:, :n].cpu().numpy()[rows]
		gpu_ref = (A[:, :k] @ B[:, :k].T).cpu().numpy()[rows]
		ti = rows // 128
		msk = ti[:, None] >= (np.arange(n) // 128)[None, :]
		e_k = np.abs((got - exact).astype(np.float64) / scale)[msk]
		e_r = np.abs((gpu_ref - exact).astype(np.float64) / scale)[msk]
		e_kr = np.abs((got - gpu_ref) / scale)[msk]
		print("same=%d rep %d: kernel vs exact max %.2e mean %.2e | torch matmul vs exact max %.2e | kernel vs torch max %.2e" % (same, rep, e_k.max(), e_k.mean(), e_r.max(), e_kr.max()), flush=True)
