"""BASELINE config 4's shape (N = 131 072, d = 32, fp64: 137 GB in place) on ONE MI355X: timing and the size-independent
identity mean(x_i) = y_i - s^2 alpha_i on 4096 training points, 0 <= sigma <= 1, finite evidence."""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth
from stpy_amd import GaussianProcess

def main():
	n, d, m = 131072, 32, 4096
	dev = torch.device("cuda:0")
	x, y, xt = synth(n, d, m, dev)
	s = 0.1
	gp = GaussianProcess(gamma=math.sqrt(d), s=s, kernel_name="squared_exponential", d=d)
	for it in range(2):
		torch.cuda.synchronize(); t0 = time.perf_counter()
		gp.fit_gp(x, y)
		torch.cuda.synchronize(); t1 = time.perf_counter()
		mu, std = gp.mean_std(xt)
		torch.cuda.synchronize(); t2 = time.perf_counter()
		F = n ** 3 / 3 + 2 * n * n + float(n) * n * m + 4 * n * m
		print("it %d: fit %.2f s  predict %.2f s  total %.2f s = %.1f TFLOP/s" % (it, t1 - t0, t2 - t1, t2 - t0, F / (t2 - t0) / 1e12), flush=True)
	idx = torch.arange(0, n, n // 4096, device=dev)[:4096]
	mu_tr, std_tr = gp.mean_std(x[idx])
	expect = y[idx] - s * s * gp.A.reshape(-1, 1)[idx]
	print("identity rel err %.2e   std range [%.3e, %.3e]   nan %s" % (float(torch.norm(mu_tr - expect) / torch.norm(expect)), float(std.min()), float(std.max()), bool(torch.isnan(std).any() or torch.isnan(mu).any())))
	lm = gp.log_marginal(gp.kernel_object, {}, 1.0)
	print("log_marginal %.6f  peak memory %.1f GB" % (float(lm), torch.cuda.max_memory_allocated() / 1e9))

if __name__ == "__main__":
	main()
