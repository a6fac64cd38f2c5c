"""fit_gp + mean_std for tile-aligned and ragged N (how much the guarded kernels cost a user whose N is arbitrary).
usage: python tools/ragged_bench.py [n ...]"""
import os, sys, time, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stpy_amd import GaussianProcess

def main():
	ns = [int(v) for v in sys.argv[1:]] or [32768, 32700, 32001]
	d, m = 16, 4096
	dev = torch.device("cuda", 0)
	for n in ns:
		g = torch.Generator().manual_seed(1234)
		x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
		y = torch.sin(x.sum(1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=g, dtype=torch.float64).to(dev)
		xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
		gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)
		best = 1e9
		for it in range(3):
			torch.cuda.synchronize(); t0 = time.perf_counter()
			gp.fit_gp(x, y); mu, std = gp.mean_std(xt)
			torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
		F = n ** 3 / 3 + 2 * n * n + n * n * m + 4 * n * m
		print("n %6d  m %d: %.4f s  %.1f TFLOP/s  |mu| %.6f" % (n, m, best, F / best / 1e12, float(mu.norm())), flush=True)
		del gp, x, y, xt
		torch.cuda.empty_cache()

if __name__ == "__main__":
	main()
