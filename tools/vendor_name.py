"""Prints nothing useful by itself: run under rocprofv3 --kernel-trace to learn which vendor kernel serves fp64 NT GEMMs."""
import torch
dev = torch.device("cuda:0")
for (m, n, k) in ((8192, 8192, 8192), (32768, 32768, 1024)):
	A = torch.randn(m, k, dtype=torch.float64, device=dev)
	B = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.zeros(m, n, dtype=torch.float64, device=dev)
	for _ in range(2):
		torch.matmul(A, B.t(), out=C)
	torch.cuda.synchronize()
	del A, B, C
