"""BASELINE config 3 at full size (N = 65 536, d = 16, Matern-5/2, fp32 + log_marginal): error of the fp32 path against the
fp64 path on the same (fp32-representable) inputs -- mu, sigma, log-marginal, alpha, residuals -- for s in argv (default 0.3 0.1)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from bench import synth
from stpy_amd import GaussianProcess

dev = torch.device("cuda:0")


def rel(a, b):
	return float(torch.norm(a.double() - b.double()) / torch.norm(b.double()))


def run(n, d, m, s, kernel_name="matern", nu=2.5):
	x, y, xt = synth(n, d, m, dev)
	x32, y32, xt32 = x.float(), y.float(), xt.float()
	x64, y64, xt64 = x32.double(), y32.double(), xt32.double()
	kw = dict(nu=nu) if kernel_name == "matern" else {}
	g64 = GaussianProcess(gamma=math.sqrt(d), s=s, kernel_name=kernel_name, d=d, **kw)
	g64.fit_gp(x64, y64)
	mu64, sd64 = g64.mean_std(xt64)
	lm64 = float(g64.log_marginal(g64.kernel_object, {}, 1.0))
	a64 = g64.A.clone()
	z64 = g64._z.clone()
	del g64
	torch.cuda.empty_cache()
	g32 = GaussianProcess(gamma=math.sqrt(d), s=s, kernel_name=kernel_name, d=d, **kw)
	g32.fit_gp(x32, y32)
	mu32, sd32 = g32.mean_std(xt32)
	lm32 = float(g32.log_marginal(g32.kernel_object, {}, 1.0))
	print("n=%d d=%d %s s=%.2f : rel err fp32 vs fp64  mu %.2e  sigma %.2e  lml %.2e  alpha %.2e  z %.2e   (lml64 %.4f lml32 %.4f, |mu| %.3f, mean sigma %.4f, nan32=%s)"
		  % (n, d, kernel_name, s, rel(mu32, mu64), rel(sd32, sd64), abs(lm32 - lm64) / abs(lm64), rel(g32.A, a64), rel(g32._z[:n], z64[:n]), lm64, lm32,
			 float(torch.norm(mu64)), float(sd64.mean()), bool(torch.isnan(sd32).any())), flush=True)
	# max abs errors too (sigma is small where data is dense)
	print("      max abs err mu %.2e sigma %.2e ; max|mu| %.3f" % (float((mu32.double() - mu64).abs().max()), float((sd32.double() - sd64).abs().max()), float(mu64.abs().max())), flush=True)
	del g32
	torch.cuda.empty_cache()


if __name__ == "__main__":
	ss = [float(v) for v in sys.argv[1:]] or [0.3, 0.1]
	for s in ss:
		run(65536, 16, 4096, s)
	run(16384, 16, 4096, 0.3)
	run(65536, 16, 4096, 0.1, kernel_name="squared_exponential")
