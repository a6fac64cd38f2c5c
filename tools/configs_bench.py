"""Timings of the other BASELINE.json configs (parity-test cases, not the bench line): C2, C3, C5."""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import synth
from stpy_amd import GaussianProcess, RFFEmbedding

dev = torch.device("cuda:0")


def timed(fn, reps=2):
	fn()
	torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter()
		out = fn()
		torch.cuda.synchronize()
		ts.append(time.perf_counter() - t0)
	return min(ts), out


def c2():
	n, d, m = 16384, 8, 4096
	x, y, xt = synth(n, d, m, dev)
	gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)
	def step():
		gp.fit_gp(x, y)
		return gp.mean_std(xt)
	t, _ = timed(step)
	F = n ** 3 / 3 + 2 * n * n + n * n * m + 4 * n * m
	print("C2  N=16384 d=8 SE fp64 fit+mean_std: %.4f s  %.1f TFLOP/s" % (t, F / t / 1e12), flush=True)


def c3():
	n, d, m = 65536, 16, 4096
	x, y, xt = synth(n, d, m, dev)
	x, y, xt = x.float(), y.float(), xt.float()
	gp = GaussianProcess(gamma=math.sqrt(d), s=0.3, kernel_name="matern", nu=2.5, d=d)
	def step():
		gp.fit_gp(x, y)
		mu, std = gp.mean_std(xt)
		return mu, std, gp.log_marginal(gp.kernel_object, {}, 1.0)
	t, (mu, std, lml) = timed(step)
	F = n ** 3 / 3 + 2 * n * n + n * n * m + 4 * n * m
	print("C3  N=65536 d=16 Matern-5/2 fp32 fit+mean_std+log_marginal: %.4f s  %.1f TFLOP/s  (lml %.4f, nan=%s)" % (t, F / t / 1e12, float(lml), bool(torch.isnan(std).any())), flush=True)
	# residual check of the fp32 factorisation on a probe vector: || K alpha - y || / || y ||
	A = gp.A
	del gp._L
	torch.cuda.empty_cache()
	K = gp.K
	r = K @ A - y
	print("    fp32 residual ||K alpha - y|| / ||y|| = %.2e" % float(torch.norm(r) / torch.norm(y)), flush=True)


def c5():
	n, d, m = 262144, 64, 32768
	g = torch.Generator().manual_seed(1237)
	x = torch.rand(n, d, generator=g, dtype=torch.float32).to(dev)
	np.random.seed(1237)
	emb = RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
	emb.W = emb.W.float().to(dev)
	t, z = timed(lambda: emb.embed(x))
	bytes_ = n * m * 4 + n * d * 4 + m * d * 4
	print("C5  RFF N=262144 d=64 m=32768 fp32: %.4f s  %.2f TB/s algorithmic (%.1f GB)  %.1f TFLOP/s" % (t, bytes_ / t / 1e12, bytes_ / 1e9, 2.0 * n * d * m / t / 1e12), flush=True)
	# sanity against a torch fp64 evaluation of a slice
	zs = z[:256, :].double().cpu()
	q = (emb.W.double().cpu() @ x[:256].double().cpu().T)
	ref = torch.cat([torch.cos(q[:m // 2]), torch.sin(q[m // 2:])]).T * math.sqrt(2.0 / m)
	print("    max abs err vs fp64 on a 256-row slice: %.2e (|z| <= %.2e)" % (float((zs - ref).abs().max()), math.sqrt(2.0 / m)), flush=True)


def grad():
	"""one evidence + gradient evaluation (the unit of work of optimize_params), SE, d = 16"""
	for n in (16384, 32768):
		d = 16
		x, y, _ = synth(n, d, 16, dev)
		gp = GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)
		gp.load_data((x, y))
		def step():
			g = torch.tensor([math.sqrt(d)], dtype=torch.float64, requires_grad=True)
			f = gp.log_marginal(gp.kernel_object, {'0': {'gamma': g}}, 1.0)
			f.backward()
			return float(f.detach()), float(g.grad)
		def fwd():
			return float(gp.log_marginal(gp.kernel_object, {'0': {'gamma': torch.tensor(math.sqrt(d)).double()}}, 1.0))
		tf, _ = timed(fwd)
		t, (f, g) = timed(step)
		F = n ** 3 / 3.0
		print("grad N=%d d=16 SE: value only %.4f s (%.1f TF/s of n^3/3); value+gradient %.4f s = %.2fx  (3 n^3/3 flop -> %.1f TF/s)  f=%.6f df/dgamma=%.6f"
			  % (n, tf, F / tf / 1e12, t, t / tf, 3 * F / t / 1e12, f, g), flush=True)
		del gp
		torch.cuda.empty_cache()


if __name__ == "__main__":
	for name in (sys.argv[1:] or ["c2", "c3", "c5"]):
		globals()[name]()
