"""One GEMM shape, a few launches -- target for rocprofv3 --pmc runs."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from stpy_amd import _lib as L
lib = L.load()
n, k = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
P = torch.randn(n, k, dtype=torch.float64, device=dev)
C = torch.randn(n, n, dtype=torch.float64, device=dev)
for _ in range(3):
	L.check(lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm")
torch.cuda.synchronize()
