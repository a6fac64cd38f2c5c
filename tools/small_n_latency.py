"""fit_gp + mean_std wall time at the sizes the reference's own scripts use (N = 256 ... 4096, M = 256, d = 2, fp64), through the
estimator class, with a host synchronisation per step as a caller would see it.   usage: python tools/small_n_latency.py"""
import sys, time, math
import torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess
dev = torch.device("cuda:0")
for n in (256, 512, 1024, 2048, 4096):
	d, m = 2, 256
	g = torch.Generator().manual_seed(n)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = torch.sin(3 * x.sum(1, keepdim=True))
	xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	gp = GaussianProcess(gamma=0.5, s=0.05, kernel_name="squared_exponential", d=d)
	for _ in range(3):
		gp.fit_gp(x, y); mu, sd = gp.mean_std(xt)
	torch.cuda.synchronize()
	tf, tp = [], []
	for _ in range(10):
		t0 = time.perf_counter(); gp.fit_gp(x, y); torch.cuda.synchronize(); t1 = time.perf_counter()
		mu, sd = gp.mean_std(xt); torch.cuda.synchronize(); t2 = time.perf_counter()
		tf.append(t1 - t0); tp.append(t2 - t1)
	print("N=%5d  fit %.3f ms  mean_std(M=256) %.3f ms" % (n, min(tf) * 1e3, min(tp) * 1e3), flush=True)
