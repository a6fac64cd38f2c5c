"""Three launches each of the fp32 trailing update (n = 32 768 lower, K = 1024) on the bf16 matrix cores (exact three-way split) and on the
fp32 MFMA: target for `rocprofv3 --kernel-trace --stats` and `--pmc` passes (profiles/r03_*_f32_gemm_pmc.json).
usage: python tools/f32_gemm_only.py"""
import sys
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
n, k = 32768, 1024
A = torch.randn(n, k, dtype=torch.float32, device=dev)
C = torch.randn(n, n, dtype=torch.float32, device=dev)
for route in (64, 0):
	lib.stpy_tune(26, route)
	for _ in range(3):
		L.check(lib.stpy_gemm_nt(L.F32, n, n, k, L.ptr(A), k, L.ptr(A), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
	torch.cuda.synchronize()
lib.stpy_tune(26, 64)
print("done")
