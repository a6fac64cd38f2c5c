import sys, time, math
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from bench import synth
from stpy_amd import GaussianProcess, _lib as L
lib = L.load()
dev = torch.device("cuda:0")
n, d = 32768, 16
x, y, xt = synth(n, d, 128, dev)
gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)
gp.fit_gp(x, y)
for rnd in range(2):
	for alg in (0,):
		g = torch.tensor(4.0, dtype=torch.float64, requires_grad=True)
		torch.cuda.synchronize(); t0 = time.perf_counter()
		f = gp.log_marginal(gp.kernel_object, {'0': {'gamma': g}}, 1.0)
		f.backward()
		torch.cuda.synchronize(); t = time.perf_counter() - t0
		print("alg %d: value+gradient %.3f s  f %.6f  grad %.8f" % (alg, t, float(f.detach()), float(g.grad)), flush=True)
lib.stpy_tune(5, 0)
