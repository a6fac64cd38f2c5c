"""GaussianProcess.fit_gp + mean_std at a small size a few times -- target for rocprofv3 --kernel-trace (what a small fit consists of).
usage: python tools/fit_small_trace.py [n]"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(n)
x = (torch.rand(n, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.sin(3 * x.sum(1, keepdim=True))
xt = (torch.rand(256, 2, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
gp = GaussianProcess(gamma=0.5, s=0.05, kernel_name="squared_exponential", d=2)
for _ in range(5):
	gp.fit_gp(x, y); gp.mean_std(xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
	gp.fit_gp(x, y)
torch.cuda.synchronize()
print("fit_gp n=%d: %.3f ms per call (20 calls back to back)" % (n, (time.perf_counter() - t0) / 20 * 1e3))
