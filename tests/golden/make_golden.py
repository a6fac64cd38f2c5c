"""
Golden-vector generator: runs the REAL reference (Mojusko/stpy, read-only at /root/reference) in the
authoring container and stores inputs + the reference's outputs as small .npz fixtures.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The fixtures (data only: inputs and expected outputs) are committed; the reference itself never is
and never travels to the GPU box.  Nothing at test/bench/smoke time reads /root/reference.

Import notes:
* ``stpy.kernels`` and ``stpy.embeddings.embedding`` import as they are.  Cases K*, R* below are
  produced from those modules alone.
* ``stpy.continuous_processes.gauss_procc`` and ``stpy.estimator`` import optional solver packages
  at module top (cvxpy, cvxpylayers, pymanopt, torchmin, autograd_minimize, mosek) that are not
  installed here and that the squared-loss path (fit_gp / mean_std / execute / log_marginal) never
  calls.  ``sys.modules`` is pre-seeded with inert placeholders for those names so the module body
  can execute; every number stored below is computed by the reference's own code
  (torch.linalg.lstsq / lu / slogdet / solve on CPU).
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

for name in ["cvxpy", "cvxpylayers", "cvxpylayers.torch", "pymanopt", "pymanopt.manifolds",
			 "pymanopt.optimizers", "pymanopt.function", "torchmin", "autograd_minimize", "mosek"]:
	if name not in sys.modules:
		sys.modules[name] = mock.MagicMock()

from stpy.kernels import KernelFunction                                   # noqa: E402
from stpy.embeddings.embedding import RFFEmbedding                        # noqa: E402
from stpy.continuous_processes.gauss_procc import GaussianProcess         # noqa: E402
from stpy.estimator import Estimator                                      # noqa: E402
import stpy.helpers.helper as helper                                      # noqa: E402


def simple_1d(X):
	# synthetic target of BASELINE config 1 (formula of test_functions/benchmarks.py:478-482; that module
	# needs h5py to import, and y is an *input* of the fixtures, so the formula is evaluated here)
	z = (X + 0.5) * 1.2
	return -(1.4 - 3 * z) * np.sin(18 * z)


def T(a):
	return torch.from_numpy(np.ascontiguousarray(a)).double()


def N(t):
	return t.detach().numpy().copy()


def save(name, **arrays):
	path = os.path.join(HERE, name + ".npz")
	if os.path.exists(path) and "--force" not in sys.argv:
		print("%-28s kept (exists; --force regenerates)" % (name + ".npz"))
		return
	np.savez_compressed(path, **arrays)
	print("%-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def gp_outputs(GP, xtest, n_head=16):
	mu, std = GP.mean_std(xtest)
	K = GP.get_kernel()
	return dict(mu=N(mu), std=N(std), K_head=N(K[:8, :8]), K_trace=np.array(float(torch.trace(K))),
				K_fro=np.array(float(torch.norm(K))), A_head=N(GP.A[:n_head]), A_norm=np.array(float(torch.norm(GP.A))))


def lml(GP, X=None, weight=1.0):
	return N(GP.log_marginal(GP.kernel_object, {} if X is None else X, weight))


def main():
	rng = np.random.RandomState(20241101)

	# ---------------------------------------------------------------- K1: bare kernel matrices
	a = rng.uniform(-1, 1, size=(5, 3))
	b = rng.uniform(-1, 1, size=(7, 3))
	out = {}
	out["se"] = N(KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3).kernel(T(a), T(b)))
	out["se_group"] = N(KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3, group=[0, 2]).kernel(T(a), T(b)))
	out["ard"] = N(KernelFunction(kernel_name="ard", ard_gamma=torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64), kappa=0.9, d=3).kernel(T(a), T(b)))
	for nu in (0.5, 1.5, 2.5):
		out["matern_%s" % str(nu).replace(".", "")] = N(KernelFunction(kernel_name="matern", gamma=1.7, nu=nu, kappa=1.1, d=3).kernel(T(a), T(b)))
		out["ard_matern_%s" % str(nu).replace(".", "")] = N(KernelFunction(kernel_name="ard_matern", ard_gamma=torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64), nu=nu, kappa=1.1, d=3).kernel(T(a), T(b)))
	out["linear"] = N(KernelFunction(kernel_name="linear", kappa=2.0, d=3, offset=0.25).kernel(T(a), T(b)))
	k1 = KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3)
	k2 = KernelFunction(kernel_name="matern", gamma=1.7, nu=2.5, kappa=0.5, d=3)
	out["sum"] = N((k1 + k2).kernel(T(a), T(b)))
	k1 = KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3)
	k2 = KernelFunction(kernel_name="ard", ard_gamma=torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64), kappa=0.9, d=3)
	out["prod"] = N((k1 * k2).kernel(T(a), T(b)))
	# kwargs override protocol (kernels.py:138-157)
	kse = KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3)
	for i, g in enumerate((0.3, 1.0, 2.5)):
		out["se_override_%d" % i] = N(kse.kernel(T(a), T(b), **{'0': {'gamma': torch.tensor(g, dtype=torch.float64)}}))
	out["se_override_gammas"] = np.array([0.3, 1.0, 2.5])
	# self-kernel symmetry / diagonal
	x8 = rng.uniform(-1, 1, size=(9, 3))
	out["se_self"] = N(KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3).kernel(T(x8), T(x8)))
	save("K1_kernels", a=a, b=b, x8=x8, **out)

	# ---------------------------------------------------------------- G1: config-1 shaped (tutorial values)
	Ntr, M = 512, 256
	x = rng.uniform(-0.5, 0.0, size=(Ntr, 1))
	y = simple_1d(x) + 0.01 * rng.normal(size=(Ntr, 1))
	xtest = helper.interval(M, 1, L_infinity_ball=0.5)
	for tag, s in (("G1_c1_s001", 0.01), ("G1_c1_s01", 0.1)):
		GP = GaussianProcess(gamma=0.1, s=s, kappa=1.0, kernel_name="squared_exponential", d=1)
		GP.fit_gp(T(x), T(y))
		o = gp_outputs(GP, T(xtest))
		save(tag, x=x, y=y, xtest=xtest, gamma=np.array(0.1), s=np.array(s), kappa=np.array(1.0),
			 lml=lml(GP), lml_w05=lml(GP, weight=0.5), **o)

	# ---------------------------------------------------------------- G2: d=8 SE, kappa != 1, and a column group
	Ntr, M, d = 256, 64, 8
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	GP = GaussianProcess(gamma=1.0, s=0.1, kappa=1.7, kernel_name="squared_exponential", d=d)
	GP.fit_gp(T(x), T(y))
	o = gp_outputs(GP, T(xtest))
	kg = KernelFunction(kernel_name="squared_exponential", gamma=1.0, kappa=1.7, d=d, group=[0, 2, 5])
	GPg = GaussianProcess(s=0.1, kernel=kg)
	GPg.fit_gp(T(x), T(y))
	og = gp_outputs(GPg, T(xtest))
	save("G2_se_d8", x=x, y=y, xtest=xtest, gamma=np.array(1.0), s=np.array(0.1), kappa=np.array(1.7),
		 lml=lml(GP), group=np.array([0, 2, 5]), lml_group=lml(GPg),
		 **o, **{"group_" + k: v for k, v in og.items()})

	# ---------------------------------------------------------------- G3: Matern nu in {1/2, 3/2, 5/2}, d=16
	Ntr, M, d = 256, 64, 16
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	out = {}
	for nu in (0.5, 1.5, 2.5):
		GP = GaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="matern", nu=nu, d=d)
		GP.fit_gp(T(x), T(y))
		tag = "nu%s_" % str(nu).replace(".", "")
		out.update({tag + k: v for k, v in gp_outputs(GP, T(xtest)).items()})
		out[tag + "lml"] = lml(GP)
	save("G3_matern_d16", x=x, y=y, xtest=xtest, gamma=np.array(2.0), s=np.array(0.1), kappa=np.array(1.0), **out)

	# ---------------------------------------------------------------- G4: ARD and ARD-Matern, d=4
	Ntr, M, d = 200, 50, 4
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.cos(x @ np.array([[1.0], [0.5], [2.0], [0.1]])) + 0.05 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	ag = np.array([0.5, 1.0, 2.0, 4.0])
	out = {}
	kk = KernelFunction(kernel_name="ard", ard_gamma=T(ag), kappa=1.2, d=d)
	GP = GaussianProcess(s=0.2, kernel=kk)
	GP.fit_gp(T(x), T(y))
	out.update({"ard_" + k: v for k, v in gp_outputs(GP, T(xtest)).items()})
	out["ard_lml"] = lml(GP)
	out["ard_lml_override"] = lml(GP, {'0': {'ard_gamma': T(ag * 1.5)}})
	for nu in (1.5, 2.5):
		kk = KernelFunction(kernel_name="ard_matern", ard_gamma=T(ag), nu=nu, kappa=1.2, d=d)
		GP = GaussianProcess(s=0.2, kernel=kk)
		GP.fit_gp(T(x), T(y))
		tag = "ardm%s_" % str(nu).replace(".", "")
		out.update({tag + k: v for k, v in gp_outputs(GP, T(xtest)).items()})
		out[tag + "lml"] = lml(GP)
	save("G4_ard_d4", x=x, y=y, xtest=xtest, ard_gamma=ag, s=np.array(0.2), kappa=np.array(1.2), **out)

	# ---------------------------------------------------------------- G5: composite kernels through the GP
	Ntr, M, d = 128, 32, 3
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(2 * x[:, :1]) + x[:, 1:2] * x[:, 2:3] + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	out = {}
	k1 = KernelFunction(kernel_name="squared_exponential", gamma=0.8, kappa=1.0, d=d)
	k2 = KernelFunction(kernel_name="matern", gamma=1.5, nu=2.5, kappa=0.5, d=d)
	GP = GaussianProcess(s=0.1, kernel=k1 + k2)
	GP.fit_gp(T(x), T(y))
	out.update({"sum_" + k: v for k, v in gp_outputs(GP, T(xtest)).items()})
	out["sum_lml"] = lml(GP)
	k1 = KernelFunction(kernel_name="squared_exponential", gamma=0.8, kappa=1.0, d=d)
	k2 = KernelFunction(kernel_name="squared_exponential", gamma=2.0, kappa=0.7, d=d, group=[1, 2])
	GP = GaussianProcess(s=0.1, kernel=k1 * k2)
	GP.fit_gp(T(x), T(y))
	out.update({"prod_" + k: v for k, v in gp_outputs(GP, T(xtest)).items()})
	out["prod_lml"] = lml(GP)
	save("G5_composite", x=x, y=y, xtest=xtest, s=np.array(0.1), **out)

	# ---------------------------------------------------------------- G6: kwargs override + log_marginal(kernel, X, weight)
	Ntr, d = 192, 2
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(3 * x[:, :1]) * np.cos(2 * x[:, 1:2]) + 0.1 * rng.normal(size=(Ntr, 1))
	GP = GaussianProcess(gamma=0.5, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(T(x), T(y))
	gam = np.array([0.2, 0.5, 1.3])
	lmls = np.zeros((3, 2))
	for i, g in enumerate(gam):
		for j, w in enumerate((1.0, 0.5)):
			lmls[i, j] = float(GP.log_marginal(GP.kernel_object, {'0': {'gamma': torch.tensor(g, dtype=torch.float64)}}, w))
	# Estimator.log_marginal (explicit Cholesky, estimator.py:32-40) on the same object
	lml_est = N(Estimator.log_marginal(GP, GP.kernel_object, {}, 1.0))
	save("G6_override_lml", x=x, y=y, gamma=np.array(0.5), s=np.array(0.1), gammas=gam, weights=np.array([1.0, 0.5]),
		 lmls=lmls, lml_default=lml(GP), lml_estimator=lml_est)

	# ---------------------------------------------------------------- G7: full=True covariance, prior branch, execute
	Ntr, M, d = 96, 32, 2
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(3 * x[:, :1]) + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	GP = GaussianProcess(gamma=0.6, s=0.1, kappa=1.4, kernel_name="squared_exponential", d=d)
	# unfitted prior branch: full=False raises TypeError in this snapshot (gauss_procc.py:346 evaluates
	# kernel(self.x=None, xtest) before the fitted check); only full=True reaches :349-363
	try:
		GP.mean_std(T(xtest))
		prior_full_false_raises = 0
	except TypeError:
		prior_full_false_raises = 1
	mu0f, cov0f = GP.mean_std(T(xtest), full=True)
	ks0, kss0 = GP.execute(T(xtest))
	assert ks0 is None
	GP.fit_gp(T(x), T(y))
	muf, cov = GP.mean_std(T(xtest), full=True)
	ks, kss = GP.execute(T(xtest))
	mean_only = GP.mean(T(xtest))
	save("G7_full_prior", x=x, y=y, xtest=xtest, gamma=np.array(0.6), s=np.array(0.1), kappa=np.array(1.4),
		 prior_full_false_raises=np.array(prior_full_false_raises), prior_full_mu=N(mu0f), prior_full_cov=N(cov0f), prior_kss=N(kss0),
		 full_mu=N(muf), full_cov=N(cov), exec_ks=N(ks), exec_kss=N(kss), mean=N(mean_only),
		 ucb=N(GP.ucb(T(xtest))), lcb=N(GP.lcb(T(xtest))))

	# ---------------------------------------------------------------- G8: chunked prediction (max_size patched to 64, M=200)
	Ntr, M, d = 128, 200, 2
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(3 * x[:, :1]) + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	GP = GaussianProcess(gamma=0.6, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(T(x), T(y))
	GP.max_size = 64
	mu, std = GP.mean_std(T(xtest))
	save("G8_chunked", x=x, y=y, xtest=xtest, gamma=np.array(0.6), s=np.array(0.1), max_size=np.array(64), mu=N(mu), std=N(std))

	# ---------------------------------------------------------------- G9: add_data_point x3 then mean_std
	d = 2
	xs = [rng.uniform(-1, 1, size=(n, d)) for n in (40, 1, 7)]
	ys = [np.sin(3 * xx[:, :1]) + 0.1 * rng.normal(size=(xx.shape[0], 1)) for xx in xs]
	xtest = rng.uniform(-1, 1, size=(33, d))
	GP = GaussianProcess(gamma=0.6, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	for xx, yy in zip(xs, ys):
		GP.add_data_point(T(xx), T(yy))
	o = gp_outputs(GP, T(xtest))
	save("G9_add_data_point", x0=xs[0], x1=xs[1], x2=xs[2], y0=ys[0], y1=ys[1], y2=ys[2], xtest=xtest,
		 gamma=np.array(0.6), s=np.array(0.1), n=np.array(GP.n), **o)

	# ---------------------------------------------------------------- G10: RFF embed
	m, d, n = 64, 5, 33
	W = rng.normal(size=(m, d)) / 0.7
	bvec = 2 * np.pi * rng.uniform(size=(m,))
	xr = rng.uniform(0, 1, size=(n, d))
	emb = RFFEmbedding(gamma=0.7, m=m, d=d, kappa=1.0)
	emb.W = T(W)
	z = N(emb.embed(T(xr)))
	emb2 = RFFEmbedding(gamma=0.7, m=m, d=d, kappa=2.5)
	emb2.W = T(W)
	z_k = N(emb2.embed(T(xr)))
	embb = RFFEmbedding(gamma=0.7, m=m, d=d, kappa=2.5, biased=True)
	embb.W = T(W)
	embb.b = T(bvec)
	z_b = N(embb.embed(T(xr)))
	# x with fewer columns than d: embed uses W[:, 0:d_x]   (embedding.py:230,234)
	z_sub = N(emb.embed(T(xr[:, :3])))
	# sampler pin: np.random.seed(0) -> W = N(0,1)/gamma
	np.random.seed(0)
	embs = RFFEmbedding(gamma=0.7, m=8, d=3)
	W_seed0 = N(embs.W)
	np.random.seed(0)
	embsb = RFFEmbedding(gamma=0.7, m=8, d=3, biased=True)
	save("G10_rff", W=W, b=bvec, x=xr, gamma=np.array(0.7), m=np.array(m), z=z, z_kappa25=z_k, z_biased_kappa25=z_b,
		 z_sub3=z_sub, W_seed0=W_seed0, W_seed0_biased=N(embsb.W), b_seed0_biased=N(embsb.b))

	# ---------------------------------------------------------------- G11: back_prop=False (LU) branch equals default
	Ntr, M, d = 128, 40, 3
	x = rng.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(Ntr, 1))
	xtest = rng.uniform(-1, 1, size=(M, d))
	GP = GaussianProcess(gamma=1.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.back_prop = False
	GP.fit_gp(T(x), T(y))
	mu, std = GP.mean_std(T(xtest))
	GP2 = GaussianProcess(gamma=1.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP2.fit_gp(T(x), T(y))
	mu2, std2 = GP2.mean_std(T(xtest))
	save("G11_lu_branch", x=x, y=y, xtest=xtest, gamma=np.array(1.0), s=np.array(0.1), mu_lu=N(mu), std_lu=N(std),
		 mu=N(mu2), std=N(std2))

	# ---------------------------------------------------------------- G12: KernelizedFeatures (primal ridge on RFF features)
	from stpy.continuous_processes.kernelized_features import KernelizedFeatures
	rng12 = np.random.RandomState(121)
	m, d, Ntr, M = 64, 3, 300, 50
	W = rng12.normal(size=(m, d)) / 0.8
	x = rng12.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(2 * x[:, :1]) + x[:, 1:2] * x[:, 2:3] + 0.1 * rng12.normal(size=(Ntr, 1))
	xtest = rng12.uniform(-1, 1, size=(M, d))
	emb = RFFEmbedding(gamma=0.8, m=m, d=d, kappa=1.5)
	emb.W = T(W)
	KF = KernelizedFeatures(embedding=emb, m=m, s=0.2, lam=1.3, d=d)
	KF.fit_gp(T(x), T(y))
	mu, std = KF.mean_std(T(xtest))
	theta, Z = KF.theta_mean(var=True)
	save("G12_kernelized_features", W=W, x=x, y=y, xtest=xtest, gamma=np.array(0.8), kappa=np.array(1.5), s=np.array(0.2), lam=np.array(1.3),
		 mu=N(mu), std=N(std), theta=N(theta), Z_head=N(Z[:8, :8]), V_head=N(KF.V[:8, :8]), kernel_head=N(KF.kernel(T(x[:5]), T(x[:7]))))

	# ---------------------------------------------------------------- H1: helpers
	save("H1_helpers", interval_5_2=helper.interval(5, 2), interval_4_1_half=helper.interval(4, 1, L_infinity_ball=0.5),
		 cartesian_2x3=helper.cartesian([np.array([1., 2.]), np.array([3., 4., 5.])]))

	# ---------------------------------------------------------------- K2: additive-group, full-covariance and polynomial kernels
	# (own RNG stream so the fixtures above stay reproducible)
	rng2 = np.random.RandomState(20241102)
	a = rng2.uniform(-1, 1, size=(6, 5))
	b = rng2.uniform(-1, 1, size=(9, 5))
	x7 = rng2.uniform(-1, 1, size=(7, 5))
	out = {}
	groups = [[0, 1], [2], [3, 4]]
	ag = torch.tensor([0.5, 1.0, 2.0, 0.8, 1.5], dtype=torch.float64)
	out["ard_additive"] = N(KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups).kernel(T(a), T(b)))
	gpg = [0.6, 1.1, 2.0]
	out["se_per_group"] = N(KernelFunction(kernel_name="squared_exponential_per_group", kappa=1.3, d=5, groups=groups,
										   params={'gamma_per_group': gpg}).kernel(T(a), T(b)))
	apg = torch.tensor([0.5, 1.0, 2.0, 0.8, 1.5], dtype=torch.float64)
	out["ard_per_group"] = N(KernelFunction(kernel_name="ard_per_group", kappa=1.3, d=5, groups=groups,
											params={'ard_per_group': apg}).kernel(T(a), T(b)))
	cov = rng2.normal(size=(5, 5)) * 0.6
	out["cov"] = cov
	out["fullcov_se"] = N(KernelFunction(kernel_name="full_covariance_se", cov=T(cov), kappa=1.2, d=5).kernel(T(a), T(b)))
	cov3 = rng2.normal(size=(3, 3)) * 0.8
	out["cov3"] = cov3
	out["fullcov_se_group"] = N(KernelFunction(kernel_name="full_covariance_se", cov=T(cov3), kappa=1.2, d=5, group=[0, 2, 4]).kernel(T(a), T(b)))
	for nu in (0.5, 1.5, 2.5):
		out["fullcov_matern_%s" % str(nu).replace(".", "")] = N(KernelFunction(kernel_name="full_covariance_matern", cov=T(cov), nu=nu, kappa=0.7, d=5).kernel(T(a), T(b)))
	for p in (1, 2, 3, 5):
		out["poly_%d" % p] = N(KernelFunction(kernel_name="polynomial", power=p, kappa=1.4, d=5).kernel(T(a), T(b)))
	out["poly_3_group"] = N(KernelFunction(kernel_name="polynomial", power=3, kappa=1.4, d=5, group=[1, 3]).kernel(T(a), T(b)))
	# additive polynomial: the reference indexes the already-subset columns with `group` again
	try:
		KernelFunction(kernel_name="polynomial", power=2, kappa=1.0, d=5, groups=groups).kernel(T(a), T(b))
		out["poly_additive_raises"] = np.array(0)
	except Exception as e:                                          # noqa: BLE001
		out["poly_additive_raises"] = np.array(1)
	# kernel algebra on top of the new families, and the self-kernel of a multi-term item
	k1 = KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups)
	k2 = KernelFunction(kernel_name="polynomial", power=2, kappa=0.3, d=5)
	out["sum_additive_poly"] = N((k1 + k2).kernel(T(a), T(b)))
	k1 = KernelFunction(kernel_name="squared_exponential", gamma=0.9, kappa=1.1, d=5)
	k2 = KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups)
	out["prod_se_additive"] = N((k1 * k2).kernel(T(a), T(b)))
	out["ard_additive_self"] = N(KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups).kernel(T(x7), T(x7)))
	# a GP on an additive kernel end to end
	x = rng2.uniform(-1, 1, size=(96, 5)); xtest = rng2.uniform(-1, 1, size=(20, 5))
	y = np.sin(x[:, :1] + x[:, 2:3]) + 0.05 * rng2.normal(size=(96, 1))
	GP = GaussianProcess(kernel=KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups), s=0.1, d=5)
	GP.fit_gp(T(x), T(y))
	mu, std = GP.mean_std(T(xtest))
	save("K2_more_kernels", a=a, b=b, x7=x7, groups_flat=np.array([0, 1, -1, 2, -1, 3, 4]), p_ard_gamma=N(ag), p_gamma_per_group=np.array(gpg),
		 p_ard_per_group=N(apg), gp_x=x, gp_y=y, gp_xtest=xtest, gp_mu=N(mu), gp_std=N(std), gp_lml=lml(GP), **out)

	# ---------------------------------------------------------------- Q1: quadrature / Hermite embeddings (embedding.py:250-707)
	import stpy.embeddings.embedding as E
	rngq = np.random.RandomState(20241104)
	out = {}
	x1 = rngq.uniform(-1, 1, size=(37, 1))
	x2 = rngq.uniform(-1, 1, size=(29, 2))
	x3 = rngq.uniform(-1, 1, size=(11, 3))
	cases = [
		("hermite_d1", E.HermiteEmbedding, dict(gamma=0.4, m=32, d=1, kappa=1.0), x1),
		("hermite_d2", E.HermiteEmbedding, dict(gamma=0.7, m=128, d=2, kappa=2.5), x2),
		("hermite_d3", E.HermiteEmbedding, dict(gamma=1.1, m=2 * 4 ** 3, d=3, kappa=1.0), x3),
		("hermite_ones", E.HermiteEmbedding, dict(gamma=0.5, m=16, d=1, ones=True), x1),
		("hermite_cosarg", E.HermiteEmbedding, dict(gamma=0.5, m=16, d=1, cosine=True), x1),      # keyword is reset by the base class
		("quad_d1", E.QuadratureEmbedding, dict(gamma=0.3, m=40, d=1, kappa=1.3), x1),
		("quad_d2_scale", E.QuadratureEmbedding, dict(gamma=0.6, m=72, d=2, scale=2.0), x2),
		("quad_cos_d1", E.QuadratureEmbedding, dict(gamma=0.3, m=22, d=1, cosine=True), x1),
		("quad_cos_d2", E.QuadratureEmbedding, dict(gamma=0.6, m=26, d=2, cosine=True), x2),       # q = 5: an ODD number of features
		("trapezoidal_d1", E.TrapezoidalEmbedding, dict(gamma=0.8, m=24, d=1), x1),
		("clenshaw_d1", E.ClenshawCurtisEmbedding, dict(gamma=0.8, m=24, d=1), x1),
		("overcomplete_d1", E.OverCompleteHermiteEmbedding, dict(gamma=0.4, m=20, d=1), x1),
		("lattice_d2", E.LatticeEmbedding, dict(gamma=0.9, m=32, d=2), x2),
		("matern_laplace_d1", E.MaternEmbedding, dict(gamma=0.5, m=20, d=1, kernel="laplace"), x1),
		("matern_nu2_d1", E.MaternEmbedding, dict(gamma=0.5, m=20, d=1, kernel="modified_matern", nu=2), x1),
	]
	names = []
	for tag, cls, kw, xx in cases:
		emb = cls(**kw)
		out[tag + "_W"] = N(emb.W)
		out[tag + "_weights"] = N(emb.weights)
		out[tag + "_m"] = np.array(emb.get_m())
		out[tag + "_z"] = N(emb.embed(T(xx)))
		names.append(tag)
	# (in one dimension Phi Phi^T converges to the SE kernel; the positive-orthant tensor grids for d > 1 do not --
	# they are what the reference computes, and that is what is pinned)
	out["se_d1_gram_gamma04"] = N(KernelFunction(kernel_name="squared_exponential", gamma=0.4, d=1).kernel(T(x1), T(x1)))
	save("Q1_quadrature", x1=x1, x2=x2, x3=x3, **out)

	# ---------------------------------------------------------------- G13: KernelizedFeatures on Hermite features (tutorial: exact GP vs QFF)
	rng13 = np.random.RandomState(131)
	d, Ntr, M = 2, 250, 60
	x = rng13.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(3 * x[:, :1]) * np.cos(2 * x[:, 1:2]) + 0.1 * rng13.normal(size=(Ntr, 1))
	xtest = rng13.uniform(-1, 1, size=(M, d))
	emb = E.HermiteEmbedding(gamma=0.5, m=2 * 10 ** 2, d=d, kappa=1.2)
	KF = KernelizedFeatures(embedding=emb, m=emb.get_m(), s=0.15, lam=1.0, d=d)
	KF.fit_gp(T(x), T(y))
	mu, std = KF.mean_std(T(xtest))
	GP = GaussianProcess(gamma=0.5, s=0.15, kappa=1.2, kernel_name="squared_exponential", d=d)
	GP.fit_gp(T(x), T(y))
	mu_gp, std_gp = GP.mean_std(T(xtest))
	save("G13_hermite_features", x=x, y=y, xtest=xtest, gamma=np.array(0.5), kappa=np.array(1.2), s=np.array(0.15), lam=np.array(1.0), m=np.array(emb.get_m()),
		 mu=N(mu), std=N(std), mu_exact_gp=N(mu_gp), std_exact_gp=N(std_gp))

	# ---------------------------------------------------------------- G14: gradients of log_marginal
	# autograd THROUGH THE REFERENCE's own GaussianProcess.log_marginal (gauss_procc.py:497-504 -> :631-638: Gram by the kernel's
	# torch ops, slogdet + solve), the function estimator.py:156-190 hands to its optimisers.  Differentiable there: SE gamma, ARD
	# ard_gamma (also with additive groups), sums of such items, and the noise level when self.s is a tensor.  The Matern kernels
	# go through NumPy (kernels.py:840-859) and carry no gradient in the reference.
	rng14 = np.random.RandomState(20241114)
	Ntr, d = 200, 3
	x = rng14.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(2 * x[:, :1]) + 0.5 * x[:, 1:2] * x[:, 2:3] + 0.1 * rng14.normal(size=(Ntr, 1))
	s14 = 0.15
	ag = np.array([0.6, 1.1, 1.7])
	groups14 = [[0], [1, 2]]
	out = {}

	def grad_case(tag, make_kernel, leaves, X_of, s_leaf=False):
		for w in (1.0, 0.5):
			GP = GaussianProcess(kernel=make_kernel(), s=s14, d=d)
			GP.fit_gp(T(x), T(y))
			lv = [torch.tensor(np.atleast_1d(v), dtype=torch.float64, requires_grad=True) for v in leaves]
			if s_leaf:
				sv = torch.tensor(s14, dtype=torch.float64, requires_grad=True)
				GP.s = sv
			f = GP.log_marginal(GP.kernel_object, X_of(lv), w)
			f.backward()
			sfx = "_w%s" % str(w).replace(".", "")
			out[tag + sfx + "_value"] = N(f)
			for i, v in enumerate(lv):
				out[tag + sfx + "_grad%d" % i] = N(v.grad)
			if s_leaf:
				out[tag + sfx + "_grad_s"] = N(sv.grad)
		for i, v in enumerate(leaves):
			out[tag + "_leaf%d" % i] = np.atleast_1d(np.asarray(v, dtype=np.float64))

	grad_case("se", lambda: KernelFunction(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d), [0.7],
			  lambda v: {'0': {'gamma': v[0]}})
	grad_case("se_noise", lambda: KernelFunction(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d), [0.7],
			  lambda v: {'0': {'gamma': v[0]}}, s_leaf=True)
	grad_case("ard", lambda: KernelFunction(kernel_name="ard", ard_gamma=T(ag), kappa=0.8, d=d), [ag * 1.1],
			  lambda v: {'0': {'ard_gamma': v[0]}})
	grad_case("ard_groups", lambda: KernelFunction(kernel_name="ard", ard_gamma=T(ag), kappa=1.1, d=d, groups=groups14), [ag * 0.9],
			  lambda v: {'0': {'ard_gamma': v[0]}})
	grad_case("sum", lambda: KernelFunction(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d) + KernelFunction(kernel_name="ard", ard_gamma=T(ag), kappa=0.8, d=d),
			  [0.7, ag * 1.2], lambda v: {'0': {'gamma': v[0]}, '1': {'ard_gamma': v[1]}})
	save("G14_lml_grad", x=x, y=y, s=np.array(s14), ard_gamma=ag, groups_flat=np.array([0, -1, 1, 2]), **out)


	# ---------------------------------------------------------------- G15: KernelizedFeatures surface (SURVEY section 8f rank 3 + the class's
	# inherited / auxiliary methods): seeded feature-space samplers (kernelized_features.py:300-336, :537-551), the dual form
	# (primal=False, n < m: :229-235, :252-254, :285), get_kernel / residuals / logdet_ratio / beta("theory") (:56-106, :553-562).
	# The draws come from torch's global CPU generator (torch.normal), so the fixture stores the seed AND the draw.
	rng15 = np.random.RandomState(151)
	m, d, Ntr, M = 48, 2, 160, 40
	W = rng15.normal(size=(m, d)) / 0.7
	x = rng15.uniform(-1, 1, size=(Ntr, d))
	y = np.cos(2 * x[:, :1]) * x[:, 1:2] + 0.1 * rng15.normal(size=(Ntr, 1))
	xtest = rng15.uniform(-1, 1, size=(M, d))
	s15, lam15, kappa15 = 0.25, 1.7, 1.2

	def make_kf(primal=True, beta_fun=None, n=None):
		emb = RFFEmbedding(gamma=0.7, m=m, d=d, kappa=kappa15)
		emb.W = T(W)
		KF = KernelizedFeatures(embedding=emb, m=m, s=s15, lam=lam15, d=d, primal=primal, beta_fun=beta_fun, bound=1.3)
		nn = Ntr if n is None else n
		KF.fit_gp(T(x[:nn]), T(y[:nn]))
		return KF

	out = {}
	KF = make_kf()
	for size, seed in ((1, 7), (3, 11)):
		torch.manual_seed(seed)
		out["draw_s%d" % size] = N(torch.normal(mean=torch.zeros(size=(m, size), dtype=torch.float64), std=1.))
		torch.manual_seed(seed)
		out["theta_post_s%d" % size] = N(KF.sample_theta(size=size))
		torch.manual_seed(seed)
		out["theta_prior_s%d" % size] = N(KF.sample_theta(size=size, prior=True))
		torch.manual_seed(seed)
		out["f_post_s%d" % size] = N(KF.sample(T(xtest), size=size))
		torch.manual_seed(seed)
		out["f_prior_s%d" % size] = N(KF.sample(T(xtest), size=size, prior=True))
		ko = KernelFunction(kernel_name="squared_exponential", gamma=0.7, kappa=kappa15, d=d)
		torch.manual_seed(seed)
		out["f_matheron_s%d" % size] = N(KF.sample_matheron(T(xtest), ko, size=size))
	torch.manual_seed(23)
	xm, fm = KF.sample_and_max(T(xtest), size=1)
	out["max_x"], out["max_f"] = N(xm), N(fm)
	out["get_kernel_head"] = N(KF.get_kernel()[:6, :6])
	out["get_kernel_trace"] = np.array(float(torch.trace(KF.get_kernel())))
	out["residuals"] = N(KF.residuals())
	out["logdet_ratio_primal"] = N(KF.logdet_ratio())
	out["beta_default"] = np.array(KF.beta())
	out["beta_theory_primal"] = N(make_kf(beta_fun="theory").beta(delta=0.2))
	try:
		out["effective_dim"] = N(KF.effective_dim(T(xtest)))
		out["effective_dim_runs"] = np.array(1)
	except (AttributeError, RuntimeError):    # torch.solve was removed from torch: the reference line raises today
		out["effective_dim_runs"] = np.array(0)
	# dual form: n < m
	nd = 30
	KD = make_kf(primal=False, n=nd)
	assert KD.dual is True
	mu, std = KD.mean_std(T(xtest))
	th, Z = KD.theta_mean(var=True)
	out["dual_n"] = np.array(nd)
	out["dual_mu"], out["dual_std"], out["dual_theta"], out["dual_Z_head"] = N(mu), N(std), N(th), N(Z[:8, :8])
	out["dual_K_head"] = N(KD.K[:6, :6])
	out["dual_logdet_ratio"] = N(KD.logdet_ratio())
	out["dual_invV_head"] = N(KD.get_invV()[:6, :6])
	torch.manual_seed(31)
	out["dual_theta_post_s2"] = N(KD.sample_theta(size=2))
	out["dual_residuals"] = N(KD.residuals())
	out["dual_beta_theory"] = N(make_kf(primal=False, beta_fun="theory", n=nd).beta(delta=0.2))
	# primal=False but n >= m stays primal
	KP = make_kf(primal=False)
	assert KP.dual is False
	mu, std = KP.mean_std(T(xtest))
	out["nondual_mu"], out["nondual_std"] = N(mu), N(std)
	save("G15_kf_surface", W=W, x=x, y=y, xtest=xtest, gamma=np.array(0.7), kappa=np.array(kappa15), s=np.array(s15), lam=np.array(lam15),
		 bound=np.array(1.3), **out)


	# ---------------------------------------------------------------- G16: gradients of log_marginal w.r.t. the map of full-covariance kernels
	# (kernels.py:464-549: z = x[:, group] cov, then SE / Matern on |z_i - z_j| -- torch.mm / exp / cdist, differentiable end to end, so autograd
	# through the reference's log_marginal yields d/dcov)
	rng16 = np.random.RandomState(20241116)
	Ntr, d = 180, 3
	x = rng16.uniform(-1, 1, size=(Ntr, d))
	y = np.sin(2 * x[:, :1]) - 0.4 * x[:, 1:2] * x[:, 2:3] + 0.1 * rng16.normal(size=(Ntr, 1))
	cov16 = np.array([[0.9, 0.2, 0.0], [-0.1, 1.3, 0.3], [0.2, 0.0, 0.7]])
	cov16g = np.array([[1.1, -0.3], [0.25, 0.8]])
	out = {}

	def cov_case(tag, make_kernel, cov0):
		for w in (1.0, 0.5):
			GP = GaussianProcess(kernel=make_kernel(), s=0.2, d=d)
			GP.x, GP.y, GP.n = T(x), T(y), Ntr
			c = torch.tensor(cov0, dtype=torch.float64, requires_grad=True)
			val = GP.log_marginal(GP.kernel_object, {'0': {'cov': c}}, w)
			val.backward()
			sfx = "_w%s" % str(w).replace(".", "")
			out[tag + sfx + "_value"] = N(val)
			out[tag + sfx + "_grad"] = N(c.grad)
		out[tag + "_cov"] = np.asarray(cov0)

	cov_case("se", lambda: KernelFunction(kernel_name="full_covariance_se", cov=T(np.eye(d)), kappa=1.2, d=d), cov16)
	cov_case("se_group", lambda: KernelFunction(kernel_name="full_covariance_se", cov=T(np.eye(2)), kappa=0.9, d=d, group=[0, 2]), cov16g)
	cov_case("matern15", lambda: KernelFunction(kernel_name="full_covariance_matern", cov=T(np.eye(d)), nu=1.5, kappa=0.8, d=d), cov16)
	cov_case("matern25", lambda: KernelFunction(kernel_name="full_covariance_matern", cov=T(np.eye(d)), nu=2.5, kappa=1.1, d=d), cov16)
	save("G16_lml_grad_cov", x=x, y=y, s=np.array(0.2), **out)

	# ---------------------------------------------------------------- B1: beta() and norm() (gauss_procc.py:179-196)
	rng3 = np.random.RandomState(20241103)
	x = rng3.uniform(-1, 1, size=(14, 2)); y = np.sin(x.sum(axis=1, keepdims=True)) + 0.05 * rng3.normal(size=(14, 1))
	GP = GaussianProcess(gamma=0.7, s=0.3, kappa=1.2, kernel_name="squared_exponential", d=2)
	GP.fit_gp(T(x), T(y))
	save("B1_beta_norm", x=x, y=y, gamma=np.array(0.7), s=np.array(0.3), kappa=np.array(1.2),
		 beta_default=N(GP.beta()), beta_d01_n2=N(GP.beta(delta=0.1, norm=2.0)), norm=N(GP.norm()),
		 lcb=N(GP.lcb(T(x[:5]))), ucb=N(GP.ucb(T(x[:5]))))


if __name__ == "__main__":
	main()
