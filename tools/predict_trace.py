"""GaussianProcess.mean_std on a fitted model a few times -- target for rocprofv3 --kernel-trace.   usage: python tools/predict_trace.py n m"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(n)
x = (torch.rand(n, 8, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.sin(3 * x.sum(1, keepdim=True))
xt = (torch.rand(m, 8, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
gp = GaussianProcess(gamma=2.8, s=0.1, kernel_name="squared_exponential", d=8)
gp.fit_gp(x, y)
for _ in range(3):
	gp.mean_std(xt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
	gp.mean_std(xt)
torch.cuda.synchronize()
print("mean_std n=%d m=%d: %.3f ms per call" % (n, m, (time.perf_counter() - t0) / 10 * 1e3))
