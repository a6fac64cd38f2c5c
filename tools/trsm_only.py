"""One fit, then the block solve X = K* L^-T three times -- target for rocprofv3 --kernel-trace (timeline analysis)."""
import sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from stpy_amd import GaussianProcess, _lib as L
lib = L.load()
n, m, d = 65536, 4096, 16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.sin(x.sum(1, keepdim=True))
xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
gp = GaussianProcess(gamma=d ** 0.5, s=0.1, kernel_name="squared_exponential", d=d)
gp.fit_gp(x, y)
X = torch.empty((m, n), dtype=torch.float64, device=dev)
wk = torch.empty(int(lib.stpy_trsm_workspace_bytes(L.F64, m, n, 0)), dtype=torch.uint8, device=dev)
for it in range(3):
	gp.kernel_object._kernel_into(gp._xd, xt, X)
	torch.cuda.synchronize()
	L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(gp._L), gp._L.stride(0), L.ptr(gp._winv), gp._winv.numel(), L.ptr(X), X.stride(0), 0, 0, L.ptr(wk), wk.numel() * wk.element_size(), L.stream_ptr()), "trsm")
	torch.cuda.synchronize()
