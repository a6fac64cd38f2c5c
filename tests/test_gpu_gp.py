"""
End-to-end GPU parity of the drop-in classes (stpy_amd.KernelFunction / GaussianProcess /
RFFEmbedding) against (a) the golden vectors produced by the real reference and (b) the CPU oracle,
plus size-independent properties at BASELINE config sizes.  Tolerance: 1e-8 relative in fp64
(BASELINE.json north_star), 1e-3 for the fp32 build-only modes.
"""
import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from tests.conftest import golden, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-8


@pytest.fixture(scope="module")
def S(gpu_device):
	import stpy_amd
	return stpy_amd


def se_spec(gamma, kappa=1.0, group=None):
	return [("squared_exponential", {"gamma": float(gamma), "kappa": float(kappa), "group": group}, "-")]


def T(a, cuda=False):
	t = torch.from_numpy(np.ascontiguousarray(a)).double()
	return t.cuda() if cuda else t


def N(t):
	return t.detach().cpu().numpy()


def check_gp(GP, g, prefix="", tol=TOL, cuda=False):
	mu, std = GP.mean_std(T(g["xtest"], cuda))
	assert mu.shape == g[prefix + "mu"].shape and std.shape == g[prefix + "std"].shape
	assert mu.is_cuda == cuda
	assert rel_err(N(mu), g[prefix + "mu"]) < tol
	assert rel_err(N(std), g[prefix + "std"]) < tol
	K = N(GP.get_kernel())
	assert rel_err(K[:8, :8], g[prefix + "K_head"]) < 1e-12
	assert abs(np.trace(K) - g[prefix + "K_trace"]) / g[prefix + "K_trace"] < 1e-12
	assert abs(np.linalg.norm(K) - g[prefix + "K_fro"]) / g[prefix + "K_fro"] < 1e-12
	assert np.abs(K - K.T).max() < 1e-13
	assert rel_err(N(GP.A[:16]), g[prefix + "A_head"]) < tol * 100
	assert abs(float(torch.norm(GP.A)) - g[prefix + "A_norm"]) / g[prefix + "A_norm"] < tol * 100


def lml(GP, X=None, w=1.0):
	v = GP.log_marginal(GP.kernel_object, {} if X is None else X, w)
	assert tuple(v.shape) == (1, 1)
	return float(v.item())


def test_K1_kernels(S):
	g = golden("K1_kernels")
	a, b = T(g["a"]), T(g["b"])
	KF = S.KernelFunction
	ag = torch.tensor([0.5, 1.0, 2.0], dtype=torch.float64)
	out = KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3).kernel(a, b)
	assert tuple(out.shape) == (7, 5) and not out.is_cuda
	assert rel_err(N(out), g["se"]) < 1e-13
	assert rel_err(N(KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3, group=[0, 2]).kernel(a, b)), g["se_group"]) < 1e-13
	assert rel_err(N(KF(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=3).kernel(a, b)), g["ard"]) < 1e-13
	for nu in (0.5, 1.5, 2.5):
		tag = str(nu).replace(".", "")
		assert rel_err(N(KF(kernel_name="matern", gamma=1.7, nu=nu, kappa=1.1, d=3).kernel(a, b)), g["matern_" + tag]) < 1e-13
		assert rel_err(N(KF(kernel_name="ard_matern", ard_gamma=ag, nu=nu, kappa=1.1, d=3).kernel(a, b)), g["ard_matern_" + tag]) < 1e-13
	assert rel_err(N(KF(kernel_name="linear", kappa=2.0, d=3, offset=0.25).kernel(a, b)), g["linear"]) < 1e-14
	k = KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3) + KF(kernel_name="matern", gamma=1.7, nu=2.5, kappa=0.5, d=3)
	assert rel_err(N(k.kernel(a, b)), g["sum"]) < 1e-13
	k = KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3) * KF(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=3)
	assert rel_err(N(k.kernel(a, b)), g["prod"]) < 1e-13
	kse = KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.3, d=3)
	for i, gam in enumerate(g["se_override_gammas"]):
		out = kse.kernel(a, b, **{'0': {'gamma': torch.tensor(gam, dtype=torch.float64)}})
		assert rel_err(N(out), g["se_override_%d" % i]) < 1e-13
	assert rel_err(N(kse.kernel(a, b)), g["se"]) < 1e-13            # stored params untouched by overrides
	ks = N(kse.kernel(T(g["x8"]), T(g["x8"])))
	assert rel_err(ks, g["se_self"]) < 1e-13
	assert np.abs(np.diag(ks) - 1.3).max() < 1e-14
	# GPU-resident inputs give GPU-resident outputs
	assert kse.kernel(a.cuda(), b.cuda()).is_cuda
	with pytest.raises(NotImplementedError):
		KF(kernel_name="laplace", d=3)
	with pytest.raises(NotImplementedError):
		KF(kernel_name="gibbs", d=3)
	with pytest.raises(AssertionError):
		KF(kernel_name="no_such_kernel", d=3)


def k2_objects(S, g):
	"""(fixture key, drop-in KernelFunction) for every kernel of K2_more_kernels."""
	KF = S.KernelFunction
	groups = [[0, 1], [2], [3, 4]]
	ag = T(g["p_ard_gamma"])
	out = [
		("ard_additive", KF(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups)),
		("se_per_group", KF(kernel_name="squared_exponential_per_group", kappa=1.3, d=5, groups=groups, params={'gamma_per_group': list(g["p_gamma_per_group"])})),
		("ard_per_group", KF(kernel_name="ard_per_group", kappa=1.3, d=5, groups=groups, params={'ard_per_group': T(g["p_ard_per_group"])})),
		("fullcov_se", KF(kernel_name="full_covariance_se", cov=T(g["cov"]), kappa=1.2, d=5)),
		("fullcov_se_group", KF(kernel_name="full_covariance_se", cov=T(g["cov3"]), kappa=1.2, d=5, group=[0, 2, 4])),
		("poly_3_group", KF(kernel_name="polynomial", power=3, kappa=1.4, d=5, group=[1, 3])),
		("sum_additive_poly", KF(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups) + KF(kernel_name="polynomial", power=2, kappa=0.3, d=5)),
		("prod_se_additive", KF(kernel_name="squared_exponential", gamma=0.9, kappa=1.1, d=5) * KF(kernel_name="ard", ard_gamma=ag, kappa=0.9, d=5, groups=groups)),
	]
	for nu in (0.5, 1.5, 2.5):
		out.append(("fullcov_matern_" + str(nu).replace(".", ""), KF(kernel_name="full_covariance_matern", cov=T(g["cov"]), nu=nu, kappa=0.7, d=5)))
	for p in (1, 2, 3, 5):
		out.append(("poly_%d" % p, KF(kernel_name="polynomial", power=p, kappa=1.4, d=5)))
	return out


def test_K2_more_kernels(S):
	"""Additive-group SE/ARD, full-covariance SE/Matern and polynomial kernels (kernels.py:464-549, :620-761)
	against the reference's outputs, and at a ragged multi-tile size against the oracle."""
	from tests.test_oracle_golden import k2_specs
	g = golden("K2_more_kernels")
	a, b = T(g["a"]), T(g["b"])
	objs = k2_objects(S, g)
	for key, k in objs:
		out = k.kernel(a, b)
		assert tuple(out.shape) == (9, 6)
		assert rel_err(N(out), g[key]) < 1e-13, key
	KF = S.KernelFunction
	kadd = KF(kernel_name="ard", ard_gamma=T(g["p_ard_gamma"]), kappa=0.9, d=5, groups=[[0, 1], [2], [3, 4]])
	ks = N(kadd.kernel(T(g["x7"]), T(g["x7"])))
	assert rel_err(ks, g["ard_additive_self"]) < 1e-13
	assert rel_err(N(kadd.kernel_self_diag(T(g["x7"]))), np.diag(g["ard_additive_self"])) < 1e-13
	with pytest.raises(NotImplementedError):
		KF(kernel_name="polynomial", power=2, d=5, groups=[[0, 1], [2], [3, 4]])
	with pytest.raises(AssertionError):
		KF(kernel_name="squared_exponential_per_group", d=5, groups=[[0, 1], [2], [3, 4]]).kernel(a, b)
	# multi-tile, ragged sizes on the device against the oracle (same specs the CPU suite pins to the reference)
	rng = np.random.RandomState(77)
	A, B = rng.uniform(-1, 1, size=(300, 5)), rng.uniform(-1, 1, size=(203, 5))
	specs = dict(k2_specs(g))
	for key, k in k2_objects(S, g):
		out = k.kernel(T(A, True), T(B, True))
		assert out.is_cuda
		assert rel_err(N(out), O.kernel(A, B, specs[key])) < 1e-12, key
		# self-kernel diagonal entry point agrees with the full matrix
		full = N(k.kernel(T(A, True), T(A, True)))
		assert rel_err(N(k.kernel_self_diag(T(A, True))), np.diag(full)) < 1e-12, key
	# GP end to end on the additive kernel
	GP = S.GaussianProcess(kernel=KF(kernel_name="ard", ard_gamma=T(g["p_ard_gamma"]), kappa=0.9, d=5, groups=[[0, 1], [2], [3, 4]]), s=0.1, d=5)
	GP.fit_gp(T(g["gp_x"]), T(g["gp_y"]))
	mu, std = GP.mean_std(T(g["gp_xtest"]))
	assert rel_err(N(mu), g["gp_mu"]) < TOL and rel_err(N(std), g["gp_std"]) < TOL
	assert abs(lml(GP) - g["gp_lml"].item()) / abs(g["gp_lml"].item()) < TOL
	# a product with a multi-term item on the training Gram (scratch-buffer path + s^2 on the diagonal)
	kprod = KF(kernel_name="squared_exponential", gamma=0.9, kappa=1.1, d=5) * KF(kernel_name="ard", ard_gamma=T(g["p_ard_gamma"]), kappa=0.9, d=5, groups=[[0, 1], [2], [3, 4]])
	GP2 = S.GaussianProcess(kernel=kprod, s=0.3, d=5)
	GP2.fit_gp(T(g["gp_x"]), T(g["gp_y"]))
	Kref = O.kernel(g["gp_x"], g["gp_x"], specs["prod_se_additive"]) + 0.09 * np.eye(96)
	assert rel_err(N(GP2.get_kernel()), Kref) < 1e-12
	Lo, alo = O.fit(g["gp_x"], g["gp_y"], specs["prod_se_additive"], 0.3)
	muo, stdo = O.mean_std(g["gp_x"], Lo, alo, g["gp_xtest"], specs["prod_se_additive"])
	mu2, std2 = GP2.mean_std(T(g["gp_xtest"]))
	assert rel_err(N(mu2), muo) < TOL and rel_err(N(std2), stdo) < TOL


@pytest.mark.parametrize("name,tol", [("G1_c1_s001", 2e-6), ("G1_c1_s01", TOL)])
@pytest.mark.parametrize("cuda", [False, True])
def test_G1_config1(S, name, tol, cuda):
	"""BASELINE config 1: 1-D SE GP, N=512 / M=256 (tutorial hyper-parameters)."""
	g = golden(name)
	GP = S.GaussianProcess(gamma=float(g["gamma"]), s=float(g["s"]), kappa=float(g["kappa"]), kernel_name="squared_exponential", d=1)
	GP.fit_gp(T(g["x"], cuda), T(g["y"], cuda))
	check_gp(GP, g, tol=tol, cuda=cuda)
	assert abs(lml(GP) - g["lml"][0, 0]) / abs(g["lml"][0, 0]) < tol
	assert abs(lml(GP, w=0.5) - g["lml_w05"][0, 0]) / abs(g["lml_w05"][0, 0]) < tol
	# aliases named in BASELINE.json
	mu2, std2 = GP.mean_var(T(g["xtest"], cuda))
	assert rel_err(N(mu2), g["mu"]) < tol and rel_err(N(std2), g["std"]) < tol


def test_G2_se_d8(S):
	g = golden("G2_se_d8")
	GP = S.GaussianProcess(gamma=1.0, s=0.1, kappa=1.7, kernel_name="squared_exponential", d=8)
	GP.fit_gp(T(g["x"]), T(g["y"]))
	check_gp(GP, g)
	assert abs(lml(GP) - g["lml"][0, 0]) / abs(g["lml"][0, 0]) < TOL
	kg = S.KernelFunction(kernel_name="squared_exponential", gamma=1.0, kappa=1.7, d=8, group=[0, 2, 5])
	GPg = S.GaussianProcess(s=0.1, kernel=kg)
	GPg.fit(T(g["x"]), T(g["y"]))
	check_gp(GPg, g, prefix="group_")
	assert abs(lml(GPg) - g["lml_group"][0, 0]) / abs(g["lml_group"][0, 0]) < TOL


def test_G3_matern(S):
	g = golden("G3_matern_d16")
	for nu in (0.5, 1.5, 2.5):
		tag = "nu%s_" % str(nu).replace(".", "")
		GP = S.GaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="matern", nu=nu, d=16)
		GP.fit_gp(T(g["x"]), T(g["y"]))
		check_gp(GP, g, prefix=tag)
		assert abs(lml(GP) - g[tag + "lml"][0, 0]) / abs(g[tag + "lml"][0, 0]) < TOL


def test_G4_ard(S):
	g = golden("G4_ard_d4")
	ag = T(g["ard_gamma"])
	GP = S.GaussianProcess(s=0.2, kernel=S.KernelFunction(kernel_name="ard", ard_gamma=ag, kappa=1.2, d=4))
	GP.fit_gp(T(g["x"]), T(g["y"]))
	check_gp(GP, g, prefix="ard_")
	assert abs(lml(GP) - g["ard_lml"][0, 0]) / abs(g["ard_lml"][0, 0]) < TOL
	v = lml(GP, {'0': {'ard_gamma': ag * 1.5}})
	assert abs(v - g["ard_lml_override"][0, 0]) / abs(g["ard_lml_override"][0, 0]) < TOL
	for nu in (1.5, 2.5):
		tag = "ardm%s_" % str(nu).replace(".", "")
		GP = S.GaussianProcess(s=0.2, kernel=S.KernelFunction(kernel_name="ard_matern", ard_gamma=ag, nu=nu, kappa=1.2, d=4))
		GP.fit_gp(T(g["x"]), T(g["y"]))
		check_gp(GP, g, prefix=tag)
		assert abs(lml(GP) - g[tag + "lml"][0, 0]) / abs(g[tag + "lml"][0, 0]) < TOL


def test_G5_composite(S):
	g = golden("G5_composite")
	KF = S.KernelFunction
	GP = S.GaussianProcess(s=0.1, kernel=KF(kernel_name="squared_exponential", gamma=0.8, kappa=1.0, d=3) + KF(kernel_name="matern", gamma=1.5, nu=2.5, kappa=0.5, d=3))
	GP.fit_gp(T(g["x"]), T(g["y"]))
	check_gp(GP, g, prefix="sum_")
	assert abs(lml(GP) - g["sum_lml"][0, 0]) / abs(g["sum_lml"][0, 0]) < TOL
	GP = S.GaussianProcess(s=0.1, kernel=KF(kernel_name="squared_exponential", gamma=0.8, kappa=1.0, d=3) * KF(kernel_name="squared_exponential", gamma=2.0, kappa=0.7, d=3, group=[1, 2]))
	GP.fit_gp(T(g["x"]), T(g["y"]))
	check_gp(GP, g, prefix="prod_")
	assert abs(lml(GP) - g["prod_lml"][0, 0]) / abs(g["prod_lml"][0, 0]) < TOL


def test_G6_override_lml(S):
	g = golden("G6_override_lml")
	GP = S.GaussianProcess(gamma=0.5, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=2)
	GP.fit_gp(T(g["x"]), T(g["y"]))
	for i, gam in enumerate(g["gammas"]):
		for j, w in enumerate(g["weights"]):
			v = lml(GP, {'0': {'gamma': torch.tensor(gam, dtype=torch.float64)}}, float(w))
			assert abs(v - g["lmls"][i, j]) / abs(g["lmls"][i, j]) < TOL
	assert abs(lml(GP) - g["lml_default"][0, 0]) / abs(g["lml_default"][0, 0]) < TOL
	# log_marginal without a prior fit (load_data path, estimator.py:28-30)
	GP2 = S.GaussianProcess(gamma=0.5, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=2)
	GP2.load_data((T(g["x"]), T(g["y"])))
	assert abs(lml(GP2) - g["lml_default"][0, 0]) / abs(g["lml_default"][0, 0]) < TOL


def test_G7_full_prior_execute(S):
	g = golden("G7_full_prior")
	xt = T(g["xtest"])
	GP = S.GaussianProcess(gamma=0.6, s=0.1, kappa=1.4, kernel_name="squared_exponential", d=2)
	mu0, cov0 = GP.mean_std(xt, full=True)
	assert np.all(N(mu0) == 0) and rel_err(N(cov0), g["prior_full_cov"]) < 1e-13
	mu0, std0 = GP.mean_std(xt)                      # reference raises here; the intended prior is returned
	assert np.all(N(mu0) == 0) and np.allclose(N(std0), np.sqrt(1.4), rtol=0, atol=1e-15)
	ks0, kss0 = GP.execute(xt)
	assert ks0 is None and rel_err(N(kss0), g["prior_kss"]) < 1e-13
	GP.fit_gp(T(g["x"]), T(g["y"]))
	mu, cov = GP.mean_std(xt, full=True)
	assert rel_err(N(mu), g["full_mu"]) < TOL
	assert rel_err(N(cov), g["full_cov"]) < TOL
	ks, kss = GP.execute(xt)
	assert tuple(ks.shape) == g["exec_ks"].shape
	assert rel_err(N(ks), g["exec_ks"]) < 1e-13 and rel_err(N(kss), g["exec_kss"]) < 1e-13
	assert rel_err(N(GP.mean(xt)), g["mean"]) < TOL
	assert rel_err(N(GP.ucb(xt)), g["ucb"]) < TOL and rel_err(N(GP.lcb(xt)), g["lcb"]) < TOL


def test_G8_chunked(S):
	g = golden("G8_chunked")
	GP = S.GaussianProcess(gamma=0.6, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=2)
	GP.fit_gp(T(g["x"]), T(g["y"]))
	GP.max_size = int(g["max_size"])
	mu, std = GP.mean_std(T(g["xtest"]))
	assert tuple(mu.shape) == g["mu"].shape
	assert rel_err(N(mu), g["mu"]) < TOL and rel_err(N(std), g["std"]) < TOL


def test_G9_add_data_point(S):
	g = golden("G9_add_data_point")
	GP = S.GaussianProcess(gamma=0.6, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=2)
	for i in range(3):
		GP.add_data_point(T(g["x%d" % i]), T(g["y%d" % i]))
	assert GP.n == int(g["n"])
	check_gp(GP, g)
	GP2 = S.GaussianProcess(gamma=0.6, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=2)
	for i in range(3):
		GP2.add_data(T(g["x%d" % i]), T(g["y%d" % i]))
	check_gp(GP2, g)


def test_G10_rff(S):
	g = golden("G10_rff")
	m, d = int(g["m"]), g["W"].shape[1]
	x = T(g["x"])
	emb = S.RFFEmbedding(gamma=0.7, m=m, d=d, kappa=1.0)
	assert tuple(emb.W.shape) == (m, d) and emb.get_m() == m
	emb.W = T(g["W"])
	z = emb.embed(x)
	assert tuple(z.shape) == g["z"].shape and rel_err(N(z), g["z"]) < 1e-14
	emb2 = S.RFFEmbedding(gamma=0.7, m=m, d=d, kappa=2.5)
	emb2.W = T(g["W"])
	assert rel_err(N(emb2.embed(x)), g["z_kappa25"]) < 1e-14
	embb = S.RFFEmbedding(gamma=0.7, m=m, d=d, kappa=2.5, biased=True)
	embb.W, embb.b = T(g["W"]), T(g["b"])
	zb = embb.embed(x)
	assert tuple(zb.shape) == g["z_biased_kappa25"].shape          # (m, n): the reference's double transpose
	assert rel_err(N(zb), g["z_biased_kappa25"]) < 1e-14
	assert rel_err(N(emb.embed(x[:, :3])), g["z_sub3"]) < 1e-14
	np.random.seed(0)
	assert rel_err(N(S.RFFEmbedding(gamma=0.7, m=8, d=3).W), g["W_seed0"]) < 1e-15
	np.random.seed(0)
	eb = S.RFFEmbedding(gamma=0.7, m=8, d=3, biased=True)
	assert rel_err(N(eb.W), g["W_seed0_biased"]) < 1e-15 and rel_err(N(eb.b), g["b_seed0_biased"]) < 1e-15
	with pytest.raises(AssertionError):
		S.RFFEmbedding(gamma=0.7, m=7, d=3)
	# RFF features approximate the SE kernel: Phi Phi^T -> k(x, x) as m grows (sanity of the layout)
	np.random.seed(1)
	big = S.RFFEmbedding(gamma=0.7, m=8192, d=d)
	Phi = N(big.embed(x))
	Kse = O.squared_exponential(g["x"], g["x"], 0.7)
	assert np.abs(Phi @ Phi.T - Kse).max() < 0.06


def test_G11_lu_branch_and_helpers(S):
	g = golden("G11_lu_branch")
	GP = S.GaussianProcess(gamma=1.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=3)
	GP.back_prop = False
	GP.fit_gp(T(g["x"]), T(g["y"]))
	mu, std = GP.mean_std(T(g["xtest"]))
	assert rel_err(N(mu), g["mu_lu"]) < TOL and rel_err(N(std), g["std_lu"]) < TOL
	from stpy_amd.helpers import helper
	h = golden("H1_helpers")
	assert np.array_equal(helper.interval(5, 2), h["interval_5_2"])
	assert np.array_equal(helper.interval(4, 1, L_infinity_ball=0.5), h["interval_4_1_half"])
	assert np.array_equal(helper.cartesian([np.array([1., 2.]), np.array([3., 4., 5.])]), h["cartesian_2x3"])


def test_sample_matches_seeded_reference_formula(S):
	"""gauss_procc.py:461-482 with the same CPU-generator draws (torch.manual_seed): posterior and prior paths."""
	g = golden("G7_full_prior")
	xt = T(g["xtest"])
	nn = xt.shape[0]
	GP = S.GaussianProcess(gamma=0.6, s=0.1, kappa=1.4, kernel_name="squared_exponential", d=2)
	torch.manual_seed(7)
	f0 = GP.sample(xt, size=3)
	torch.manual_seed(7)
	rv = torch.normal(mean=torch.zeros(nn, 3, dtype=torch.float64), std=1.).numpy()
	L0 = np.linalg.cholesky(g["prior_kss"] + 10e-8 * np.eye(nn))
	assert tuple(f0.shape) == (nn, 3) and rel_err(N(f0), L0 @ rv) < 1e-6
	GP.fit_gp(T(g["x"]), T(g["y"]))
	torch.manual_seed(11)
	f = GP.sample(xt, size=2)
	torch.manual_seed(11)
	rv = torch.normal(mean=torch.zeros(nn, 2, dtype=torch.float64), std=1.).numpy()
	Lc = np.linalg.cholesky(g["full_cov"] + 10e-10 * np.eye(nn))
	ref = g["full_mu"] + Lc @ rv
	# the posterior covariance is nearly singular (jitter 1e-9): Cholesky amplifies 1e-12 differences
	assert rel_err(N(f), ref) < 1e-4
	xm, val = GP.sample_and_max(xt, size=2)
	assert tuple(val.shape) == (2,)


def _torch_lml(x, y, s, w, kind, ls, kappa):
	"""the reference formula (gauss_procc.py:631-638) in torch CPU autograd, for gradient parity"""
	xs = x / ls
	if kind == "se":
		sq = (xs ** 2).sum(1, keepdim=True) + (xs ** 2).sum(1, keepdim=True).T - 2 * xs @ xs.T
		K = kappa * torch.exp(-0.5 * sq)
	else:
		diff = xs.unsqueeze(1) - xs.unsqueeze(0)
		r = torch.sqrt((diff ** 2).sum(-1) + 1e-300)
		if kind == "m12":
			K = kappa * torch.exp(-r)
		elif kind == "m32":
			a = r * np.sqrt(3.0)
			K = kappa * (1 + a) * torch.exp(-a)
		else:
			a = r * np.sqrt(5.0)
			K = kappa * (1 + a + a ** 2 / 3.0) * torch.exp(-a)
	K = K + torch.eye(x.shape[0], dtype=torch.float64) * s * s
	return 0.5 * (y.T @ torch.linalg.solve(K, y)) + 0.5 * w * torch.slogdet(K)[1]


@pytest.mark.parametrize("name,kind,nu", [("squared_exponential", "se", None), ("matern", "m32", 1.5), ("matern", "m52", 2.5), ("matern", "m12", 0.5)])
def test_log_marginal_gradient_isotropic(S, name, kind, nu):
	"""d log_marginal / d gamma through autograd, vs torch CPU autograd of the reference formula."""
	rng = np.random.RandomState(5)
	n, d = 300, 3
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.sin(3 * x[:, :1]) + 0.1 * torch.from_numpy(rng.normal(size=(n, 1)))
	kw = dict(nu=nu) if nu else {}
	GP = S.GaussianProcess(gamma=0.9, s=0.2, kappa=1.3, kernel_name=name, d=d, **kw)
	GP.load_data((x, y))
	for w in (1.0, 0.5):
		g = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
		f = GP.log_marginal(GP.kernel_object, {'0': {'gamma': g}}, w)
		assert tuple(f.shape) == (1, 1)
		f.backward()
		gr = torch.tensor([0.7], dtype=torch.float64, requires_grad=True)
		fr = _torch_lml(x, y, 0.2, w, kind, gr, 1.3)
		fr.backward()
		assert abs(float(f.detach()) - float(fr.detach())) / abs(float(fr.detach())) < 1e-9
		assert abs(float(g.grad) - float(gr.grad)) / abs(float(gr.grad)) < 1e-7


def test_log_marginal_gradient_pinned_to_reference(S):
	"""SURVEY section 8f rank 1, pinned: value and gradient of GaussianProcess.log_marginal on the HIP path against golden G14 --
	autograd through the REFERENCE's own log_marginal (gauss_procc.py:631-638, as estimator.py:156-190 drives it) for SE gamma,
	ARD ard_gamma, additive-group ARD, a sum of two items and the noise level, at weights 1 and 0.5.  Value <= 1e-8, gradient
	<= 1e-7.  (The Matern kernels have no gradient in the reference -- kernels.py:840-859 goes through NumPy -- so their device
	gradients stay checked against the torch restatement above, which this fixture in turn validates on the SE / ARD cases.)"""
	g = golden("G14_lml_grad")
	x, y, s0 = T(g["x"]), T(g["y"]), float(g["s"])
	d = x.shape[1]
	ag = torch.from_numpy(g["ard_gamma"])
	KF = S.KernelFunction
	cases = {
		"se": (lambda: KF(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d), lambda v: {'0': {'gamma': v[0]}}, False),
		"se_noise": (lambda: KF(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d), lambda v: {'0': {'gamma': v[0]}}, True),
		"ard": (lambda: KF(kernel_name="ard", ard_gamma=ag.clone(), kappa=0.8, d=d), lambda v: {'0': {'ard_gamma': v[0]}}, False),
		"ard_groups": (lambda: KF(kernel_name="ard", ard_gamma=ag.clone(), kappa=1.1, d=d, groups=[[0], [1, 2]]), lambda v: {'0': {'ard_gamma': v[0]}}, False),
		"sum": (lambda: KF(kernel_name="squared_exponential", gamma=0.9, kappa=1.3, d=d) + KF(kernel_name="ard", ard_gamma=ag.clone(), kappa=0.8, d=d),
				lambda v: {'0': {'gamma': v[0]}, '1': {'ard_gamma': v[1]}}, False),
	}
	for tag, (mk, X_of, s_leaf) in cases.items():
		nleaf = len([k for k in g if k.startswith(tag + "_leaf")])
		for w, sfx in ((1.0, "_w10"), (0.5, "_w05")):
			GP = S.GaussianProcess(kernel=mk(), s=s0, d=d)
			GP.fit_gp(x, y)
			lv = [torch.from_numpy(g["%s_leaf%d" % (tag, i)]).clone().requires_grad_(True) for i in range(nleaf)]
			if s_leaf:
				sv = torch.tensor(s0, dtype=torch.float64, requires_grad=True)
				GP.s = sv
			f = GP.log_marginal(GP.kernel_object, X_of(lv), w)
			assert tuple(f.shape) == (1, 1)
			f.backward()
			ref = g[tag + sfx + "_value"].ravel()[0]
			assert abs(float(f.detach()) - ref) / abs(ref) < 1e-8, (tag, w)
			for i, v in enumerate(lv):
				assert rel_err(N(v.grad), g[tag + sfx + "_grad%d" % i]) < 1e-7, (tag, w, i, N(v.grad), g[tag + sfx + "_grad%d" % i])
			if s_leaf:
				want = g[tag + sfx + "_grad_s"].ravel()[0]
				assert abs(float(sv.grad) - want) / abs(want) < 1e-7, (tag, w)


def test_log_marginal_gradient_full_covariance_pinned_to_reference(S):
	"""The evidence gradient w.r.t. the map of a full-covariance kernel item (kernels.py:464-549; the reference differentiates it by
	autograd): golden G16 = autograd through the REFERENCE's log_marginal for full_covariance_se (all columns and a column group) and
	full_covariance_matern (nu = 1.5, 2.5), weights 1 and 0.5.  Value <= 1e-8, gradient <= 1e-7."""
	g = golden("G16_lml_grad_cov")
	x, y, s0 = T(g["x"]), T(g["y"]), float(g["s"])
	d = x.shape[1]
	KF = S.KernelFunction
	cases = {
		"se": lambda: KF(kernel_name="full_covariance_se", cov=torch.eye(d, dtype=torch.float64), kappa=1.2, d=d),
		"se_group": lambda: KF(kernel_name="full_covariance_se", cov=torch.eye(2, dtype=torch.float64), kappa=0.9, d=d, group=[0, 2]),
		"matern15": lambda: KF(kernel_name="full_covariance_matern", cov=torch.eye(d, dtype=torch.float64), nu=1.5, kappa=0.8, d=d),
		"matern25": lambda: KF(kernel_name="full_covariance_matern", cov=torch.eye(d, dtype=torch.float64), nu=2.5, kappa=1.1, d=d),
	}
	for tag, mk in cases.items():
		for w, sfx in ((1.0, "_w10"), (0.5, "_w05")):
			GP = S.GaussianProcess(kernel=mk(), s=s0, d=d)
			GP.load_data((x, y))
			c = torch.from_numpy(g[tag + "_cov"]).clone().requires_grad_(True)
			f = GP.log_marginal(GP.kernel_object, {'0': {'cov': c}}, w)
			f.backward()
			ref = g[tag + sfx + "_value"].ravel()[0]
			assert abs(float(f.detach()) - ref) / abs(ref) < 1e-8, (tag, w)
			assert tuple(c.grad.shape) == g[tag + "_cov"].shape
			assert rel_err(N(c.grad), g[tag + sfx + "_grad"]) < 1e-7, (tag, w, N(c.grad), g[tag + sfx + "_grad"])


def _tk(x, kind, ls, kappa, cols=None):
	"""one kernel matrix in torch CPU autograd (same forms as _torch_lml)"""
	xs = (x if cols is None else x[:, cols]) / ls
	if kind == "se":
		sq = (xs ** 2).sum(1, keepdim=True) + (xs ** 2).sum(1, keepdim=True).T - 2 * xs @ xs.T
		return kappa * torch.exp(-0.5 * sq)
	diff = xs.unsqueeze(1) - xs.unsqueeze(0)
	a = torch.sqrt((diff ** 2).sum(-1) + 1e-300) * np.sqrt(5.0)
	return kappa * (1 + a + a ** 2 / 3.0) * torch.exp(-a)


def _lml_of(K, y, s, w):
	K = K + torch.eye(K.shape[0], dtype=torch.float64) * s * s
	return 0.5 * (y.T @ torch.linalg.solve(K, y)) + 0.5 * w * torch.slogdet(K)[1]


def test_log_marginal_gradient_composite(S):
	"""Evidence gradients through sums, products and additive-group kernels (the kernels MultipleKernelLearner-style
	searches use, mkl_estimator.py:35-37), against torch CPU autograd of the same formulas."""
	rng = np.random.RandomState(8)
	n, d = 200, 4
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.sin(2 * x[:, :1]) + 0.3 * x[:, 2:3] + 0.1 * torch.from_numpy(rng.normal(size=(n, 1)))
	KF = S.KernelFunction
	ag0 = torch.tensor([0.6, 1.1, 1.7, 0.9], dtype=torch.float64)
	groups = [[0, 1], [2, 3]]

	def check(kernel, X_of, ref_of, leaves, w=1.0):
		GP = S.GaussianProcess(s=0.25, kernel=kernel, d=d)
		GP.load_data((x, y))
		dev_leaves = [v.clone().requires_grad_(True) for v in leaves]
		f = GP.log_marginal(GP.kernel_object, X_of(dev_leaves), w)
		f.backward()
		ref_leaves = [v.clone().requires_grad_(True) for v in leaves]
		fr = _lml_of(ref_of(ref_leaves), y, 0.25, w)
		fr.backward()
		assert abs(float(f.detach()) - float(fr.detach())) / abs(float(fr.detach())) < 1e-9, kernel.description()
		for a, b in zip(dev_leaves, ref_leaves):
			assert rel_err(a.grad.numpy(), b.grad.numpy()) < 1e-7, kernel.description()

	g1, g2 = torch.tensor([0.7], dtype=torch.float64), torch.tensor([1.3], dtype=torch.float64)
	# sum of two items
	check(KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.2, d=d) + KF(kernel_name="matern", gamma=1.3, nu=2.5, kappa=0.6, d=d),
		  lambda v: {'0': {'gamma': v[0]}, '1': {'gamma': v[1], 'nu': 2.5}},
		  lambda v: _tk(x, "se", v[0], 1.2) + _tk(x, "m52", v[1], 0.6), [g1, g2], w=0.5)
	# product of an isotropic and an ARD item
	check(KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.2, d=d) * KF(kernel_name="ard", ard_gamma=ag0.clone(), kappa=0.8, d=d),
		  lambda v: {'0': {'gamma': v[0]}, '1': {'ard_gamma': v[1]}},
		  lambda v: _tk(x, "se", v[0], 1.2) * _tk(x, "se", v[1], 0.8), [g1, ag0])
	# (a + b) * c: the value accumulated before the third item multiplies its derivative
	check((KF(kernel_name="squared_exponential", gamma=0.7, kappa=1.2, d=d, group=[0, 1]) + KF(kernel_name="matern", gamma=1.3, nu=2.5, kappa=0.6, d=d))
		  * KF(kernel_name="squared_exponential", gamma=2.0, kappa=1.0, d=d, group=[2, 3]),
		  lambda v: {'0': {'gamma': v[0]}, '1': {'gamma': v[1], 'nu': 2.5}, '2': {'gamma': v[2]}},
		  lambda v: (_tk(x, "se", v[0], 1.2, [0, 1]) + _tk(x, "m52", v[1], 0.6)) * _tk(x, "se", v[2], 1.0, [2, 3]),
		  [g1, g2, torch.tensor([2.0], dtype=torch.float64)])
	# additive-group ARD: mean over the groups, one lengthscale vector across them
	check(KF(kernel_name="ard", ard_gamma=ag0.clone(), kappa=1.1, d=d, groups=groups),
		  lambda v: {'0': {'ard_gamma': v[0]}},
		  lambda v: 0.5 * (_tk(x, "se", v[0][[0, 1]], 1.1, [0, 1]) + _tk(x, "se", v[0][[2, 3]], 1.1, [2, 3])), [ag0])
	# a parameter that has no device gradient is refused, not silently dropped
	GP = S.GaussianProcess(s=0.25, kernel=KF(kernel_name="full_covariance_se", cov=torch.eye(d, dtype=torch.float64), d=d), d=d)
	GP.load_data((x, y))
	gbad = torch.tensor([1.0], dtype=torch.float64, requires_grad=True)
	with pytest.raises(NotImplementedError):
		GP.log_marginal(GP.kernel_object, {'0': {'gamma': gbad}}, 1.0).backward()


def test_log_marginal_gradient_ard_and_noise(S):
	rng = np.random.RandomState(6)
	n, d = 257, 4
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.cos(x @ torch.tensor([[1.0], [0.5], [2.0], [0.1]], dtype=torch.float64)) + 0.05 * torch.from_numpy(rng.normal(size=(n, 1)))
	ag0 = torch.tensor([0.5, 1.0, 2.0, 4.0], dtype=torch.float64)
	for name, kind, nu in (("ard", "se", None), ("ard_matern", "m52", 2.5)):
		kw = dict(nu=nu) if nu else {}
		GP = S.GaussianProcess(s=0.2, kernel=S.KernelFunction(kernel_name=name, ard_gamma=ag0.clone(), kappa=1.2, d=d, **kw))
		GP.load_data((x, y))
		ag = ag0.clone().requires_grad_(True)
		f = GP.log_marginal(GP.kernel_object, {'0': {'ard_gamma': ag}}, 1.0)
		f.backward()
		agr = ag0.clone().requires_grad_(True)
		fr = _torch_lml(x, y, 0.2, 1.0, kind, agr, 1.2)
		fr.backward()
		assert abs(float(f) - float(fr)) / abs(float(fr)) < 1e-9
		assert rel_err(ag.grad.numpy(), agr.grad.numpy()) < 1e-7
	# noise std through self.s (the "bandwidth+noise" mode, estimator.py:163-166) together with gamma
	GP = S.GaussianProcess(gamma=0.8, s=0.3, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.load_data((x, y))
	sig = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
	g = torch.tensor([0.8], dtype=torch.float64, requires_grad=True)
	GP.s = sig
	f = GP.log_marginal(GP.kernel_object, {'0': {'gamma': g}}, 1.0)
	f.backward()
	sr = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
	gr = torch.tensor([0.8], dtype=torch.float64, requires_grad=True)
	fr = _torch_lml(x, y, sr, 1.0, "se", gr, 1.0)
	fr.backward()
	assert abs(float(g.grad) - float(gr.grad)) / abs(float(gr.grad)) < 1e-7
	assert abs(float(sig.grad) - float(sr.grad)) / abs(float(sr.grad)) < 1e-7


def test_optimize_params_bandwidth(S):
	"""the caller of the hot path (estimator.py:141-256): L-BFGS on the device evidence reaches the optimum that
	the same optimiser reaches on the torch-CPU restatement of the reference objective"""
	import scipy.optimize
	rng = np.random.RandomState(8)
	n, d = 200, 2
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.sin(3 * x[:, :1]) * torch.cos(2 * x[:, 1:2]) + 0.1 * torch.from_numpy(rng.normal(size=(n, 1)))
	GP = S.GaussianProcess(gamma=0.3, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x, y)
	f0 = lml(GP)
	GP.optimize_params(type="bandwidth", restarts=2, optimizer="pytorch-minimize", init_func=lambda dim: np.full(dim, 0.3), maxiter=200)
	g_opt = float(GP.kernel_object.params_dict['0']['gamma'].reshape(-1)[0])
	f1 = lml(GP)
	assert f1 < f0 and GP.fitted

	def fun(v):
		t = torch.tensor(v, dtype=torch.float64, requires_grad=True)
		f = _torch_lml(x, y, 0.1, 1.0, "se", t, 1.0)
		f.backward()
		return float(f.detach()), t.grad.numpy()
	ref = scipy.optimize.minimize(fun, np.array([0.3]), jac=True, method='L-BFGS-B', options={'gtol': 1e-4, 'ftol': 1e-12})
	assert abs(abs(g_opt) - abs(ref.x[0])) / abs(ref.x[0]) < 1e-4
	assert abs(f1 - ref.fun) / abs(ref.fun) < 1e-8
	mu, std = GP.mean_std(x[:10])
	assert mu.shape == (10, 1) and not bool(torch.isnan(std).any())


def test_optimize_params_additive_sum(S):
	"""bandwidth search over a SUM of two kernels on different coordinates (the additive models of mkl_estimator.py)"""
	import scipy.optimize
	rng = np.random.RandomState(9)
	n, d = 220, 2
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.sin(4 * x[:, :1]) + 0.5 * x[:, 1:2] ** 2 + 0.1 * torch.from_numpy(rng.normal(size=(n, 1)))
	KF = S.KernelFunction
	k = KF(kernel_name="squared_exponential", gamma=0.5, kappa=1.0, d=d, group=[0]) + KF(kernel_name="squared_exponential", gamma=0.5, kappa=1.0, d=d, group=[1])
	GP = S.GaussianProcess(s=0.1, kernel=k, d=d)
	GP.fit_gp(x, y)
	f0 = lml(GP)
	GP.optimize_params(type="bandwidth", restarts=1, optimizer="pytorch-minimize", init_func=lambda dim: np.full(dim, 0.5), maxiter=300)
	g = [abs(float(GP.kernel_object.params_dict[key]['gamma'].reshape(-1)[0])) for key in ('0', '1')]
	f1 = lml(GP)
	assert f1 < f0

	def fun(v):
		t = torch.tensor(v, dtype=torch.float64, requires_grad=True)
		f = _lml_of(_tk(x, "se", t[0], 1.0, [0]) + _tk(x, "se", t[1], 1.0, [1]), y, 0.1, 1.0)
		f.backward()
		return float(f.detach()), t.grad.numpy()
	ref = scipy.optimize.minimize(fun, np.array([0.5, 0.5]), jac=True, method='L-BFGS-B', options={'gtol': 1e-4, 'ftol': 1e-12})
	assert abs(f1 - ref.fun) / abs(ref.fun) < 1e-7
	assert abs(g[0] - abs(ref.x[0])) / abs(ref.x[0]) < 1e-3 and abs(g[1] - abs(ref.x[1])) / abs(ref.x[1]) < 1e-3


def test_G12_kernelized_features(S):
	"""primal ridge on RFF features vs the reference golden (pinverse path)"""
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	g = golden("G12_kernelized_features")
	m, d = g["W"].shape
	for cuda in (False, True):
		emb = S.RFFEmbedding(gamma=0.8, m=m, d=d, kappa=1.5)
		emb.W = T(g["W"])
		KF = KernelizedFeatures(embedding=emb, m=m, s=0.2, lam=1.3, d=d)
		KF.fit_gp(T(g["x"], cuda), T(g["y"], cuda))
		mu, std = KF.mean_std(T(g["xtest"], cuda))
		assert mu.is_cuda == cuda and tuple(mu.shape) == g["mu"].shape
		assert rel_err(N(mu), g["mu"]) < TOL and rel_err(N(std), g["std"]) < TOL
		theta, Z = KF.theta_mean(var=True)
		assert rel_err(N(theta), g["theta"]) < TOL
		assert rel_err(N(Z)[:8, :8], g["Z_head"]) < 1e-7
		assert rel_err(N(KF.V)[:8, :8], g["V_head"]) < 1e-12
		kk = KF.kernel(T(g["x"][:5], cuda), T(g["x"][:7], cuda))
		assert tuple(kk.shape) == (7, 5) and rel_err(N(kk), g["kernel_head"]) < 1e-12
	# queued points (kernelized_features.py:108-113): nothing is refactored until the next prediction folds them in
	KF.add_data_point(T(g["x"][:3], True), T(g["y"][:3], True))        # (KF holds cuda data after the loop)
	KF.add_data_point(T(g["x"][3:5], True), T(g["y"][3:5], True))
	assert KF.fitted is False and len(KF.to_add) == 2 and KF.n == g["x"].shape[0]
	mu, std = KF.mean_std(T(g["xtest"], True))
	assert KF.fitted is True and KF.to_add == [] and KF.n == g["x"].shape[0] + 5
	xx, yy = np.concatenate([g["x"], g["x"][:5]]), np.concatenate([g["y"], g["y"][:5]])
	Q = O.rff_embed(xx, g["W"], m, kappa=1.5)
	_, invV, theta = O.kernelized_features_fit(Q, yy, 0.2, 1.3)
	mu_o, std_o = O.kernelized_features_mean_std(O.rff_embed(g["xtest"], g["W"], m, kappa=1.5), invV, theta, 0.2)
	assert rel_err(N(mu), mu_o) < TOL and rel_err(N(std), std_o) < TOL


def _kf15(S, g, cuda, primal=True, beta_fun=None, n=None):
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	m, d = g["W"].shape
	emb = S.RFFEmbedding(gamma=float(g["gamma"]), m=m, d=d, kappa=float(g["kappa"]))
	emb.W = T(g["W"])
	KF = KernelizedFeatures(embedding=emb, m=m, s=float(g["s"]), lam=float(g["lam"]), d=d, primal=primal, beta_fun=beta_fun, bound=float(g["bound"]))
	nn = g["x"].shape[0] if n is None else n
	KF.fit_gp(T(g["x"][:nn], cuda), T(g["y"][:nn], cuda))
	return KF


def test_G15_feature_space_sampling(S):
	"""sample_theta / sample / sample_matheron / sample_and_max (kernelized_features.py:300-336, :537-551) with the
	reference's seeded CPU-generator draws, against the imported reference."""
	g = golden("G15_kf_surface")
	for cuda in (False, True):
		KF = _kf15(S, g, cuda)
		assert isinstance(KF, S.GaussianProcess)
		for size, seed in ((1, 7), (3, 11)):
			torch.manual_seed(seed)
			th = KF.sample_theta(size=size)
			assert tuple(th.shape) == g["theta_post_s%d" % size].shape and th.is_cuda == cuda
			assert rel_err(N(th), g["theta_post_s%d" % size]) < TOL
			torch.manual_seed(seed)
			assert rel_err(N(KF.sample_theta(size=size, prior=True)), g["theta_prior_s%d" % size]) < 1e-14
			torch.manual_seed(seed)
			f = KF.sample(T(g["xtest"], cuda), size=size)
			assert tuple(f.shape) == g["f_post_s%d" % size].shape and rel_err(N(f), g["f_post_s%d" % size]) < TOL
			torch.manual_seed(seed)
			assert rel_err(N(KF.sample(T(g["xtest"], cuda), size=size, prior=True)), g["f_prior_s%d" % size]) < 1e-12
			ko = S.KernelFunction(kernel_name="squared_exponential", gamma=float(g["gamma"]), kappa=float(g["kappa"]), d=g["W"].shape[1])
			torch.manual_seed(seed)
			fm = KF.sample_matheron(T(g["xtest"], cuda), ko, size=size)
			assert tuple(fm.shape) == g["f_matheron_s%d" % size].shape and rel_err(N(fm), g["f_matheron_s%d" % size]) < TOL
		torch.manual_seed(23)
		xm, fmax = KF.sample_and_max(T(g["xtest"], cuda), size=1)
		assert rel_err(N(xm), g["max_x"]) < 1e-15 and rel_err(N(fmax), g["max_f"]) < TOL
	# an unfitted object samples the prior (kernelized_features.py:331-334)
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	m, d = g["W"].shape
	emb = S.RFFEmbedding(gamma=0.7, m=m, d=d)
	emb.W = T(g["W"])
	K0 = KernelizedFeatures(embedding=emb, m=m, s=0.1, lam=float(g["lam"]), d=d)
	torch.manual_seed(7)
	assert rel_err(N(K0.sample_theta(size=1)), g["theta_prior_s1"]) < 1e-14


def test_G15_kernelized_features_surface_and_dual(S):
	"""get_kernel / residuals / logdet_ratio / beta / effective_dim and the dual form (primal=False, n < m) of KernelizedFeatures
	(kernelized_features.py:56-106, :164-174, :229-235, :252-254, :285, :553-562) against the imported reference."""
	g = golden("G15_kf_surface")
	m = g["W"].shape[0]
	s, lam, kappa = float(g["s"]), float(g["lam"]), float(g["kappa"])
	for cuda in (False, True):
		KF = _kf15(S, g, cuda)
		Kk = KF.get_kernel()
		assert rel_err(N(Kk)[:6, :6], g["get_kernel_head"]) < 1e-12 and abs(float(torch.trace(Kk)) - g["get_kernel_trace"]) / g["get_kernel_trace"] < 1e-12
		assert abs(float(KF.residuals()) - g["residuals"]) / g["residuals"] < TOL
		assert abs(float(KF.logdet_ratio()) - g["logdet_ratio_primal"]) < 1e-10
		assert KF.beta() == 2.0 and tuple(KF.K.shape) == (1, 1)
		assert abs(float(_kf15(S, g, cuda, beta_fun="theory").beta(delta=0.2)) - g["beta_theory_primal"]) / abs(g["beta_theory_primal"]) < TOL
		ed = float(KF.effective_dim(T(g["xtest"], cuda)))
		assert abs(ed - O.kernelized_features_effective_dim(O.rff_embed(g["xtest"], g["W"], m, kappa=kappa), lam)) / ed < TOL
		# dual form
		nd = int(g["dual_n"])
		KD = _kf15(S, g, cuda, primal=False, n=nd)
		assert KD.dual is True
		mu, std = KD.mean_std(T(g["xtest"], cuda))
		assert mu.is_cuda == cuda and rel_err(N(mu), g["dual_mu"]) < TOL and rel_err(N(std), g["dual_std"]) < TOL
		th, Z = KD.theta_mean(var=True)
		assert rel_err(N(th), g["dual_theta"]) < TOL and rel_err(N(Z)[:8, :8], g["dual_Z_head"]) < 1e-7
		assert rel_err(N(KD.K)[:6, :6], g["dual_K_head"]) < 1e-12
		assert abs(float(KD.logdet_ratio()) - g["dual_logdet_ratio"]) / abs(g["dual_logdet_ratio"]) < TOL
		assert rel_err(N(KD.get_invV())[:6, :6], g["dual_invV_head"]) < 1e-8
		torch.manual_seed(31)
		assert rel_err(N(KD.sample_theta(size=2)), g["dual_theta_post_s2"]) < TOL
		assert abs(float(KD.residuals()) - g["dual_residuals"]) / g["dual_residuals"] < 1e-6
		assert abs(float(_kf15(S, g, cuda, primal=False, beta_fun="theory", n=nd).beta(delta=0.2)) - g["dual_beta_theory"]) / abs(g["dual_beta_theory"]) < TOL
		# dual + queued points: refit on the concatenated data; reaching n >= m converts to the primal form (check_conversion)
		KD.add_data_point(T(g["x"][nd:nd + 4], cuda), T(g["y"][nd:nd + 4], cuda))
		mu, std = KD.mean_std(T(g["xtest"], cuda))
		Q = O.rff_embed(g["x"][:nd + 4], g["W"], m, kappa=kappa)
		_, invK_V, thd = O.kernelized_features_dual_fit(Q, g["y"][:nd + 4], s, lam)
		mu_o, std_o = O.kernelized_features_dual_mean_std(O.rff_embed(g["xtest"], g["W"], m, kappa=kappa), invK_V, thd)
		assert KD.dual is True and rel_err(N(mu), mu_o) < TOL and rel_err(N(std), std_o) < TOL
		KD.add_data_point(T(g["x"][nd + 4:m + 6], cuda), T(g["y"][nd + 4:m + 6], cuda))
		mu, _ = KD.mean_std(T(g["xtest"], cuda))
		assert KD.dual is False and KD.n == m + 6
		# primal=False with n >= m is the primal path
		KP = _kf15(S, g, cuda, primal=False)
		assert KP.dual is False
		mu, std = KP.mean_std(T(g["xtest"], cuda))
		assert rel_err(N(mu), g["nondual_mu"]) < TOL and rel_err(N(std), g["nondual_std"]) < TOL


def test_kernelized_features_inherits_evidence_and_search(S):
	"""KernelizedFeatures derives from GaussianProcess (kernelized_features.py:12): log_marginal(kernel, X, weight) is the evidence of
	``kernel`` on the stored data (gauss_procc.py:631-638), with its gradient; optimize_params works once a kernel_object is attached."""
	g = golden("G15_kf_surface")
	KF = _kf15(S, g, False)
	d = g["W"].shape[1]
	ko = S.KernelFunction(kernel_name="squared_exponential", gamma=0.9, kappa=float(g["kappa"]), d=d)
	spec = se_spec(0.9, float(g["kappa"]))
	val = KF.log_marginal(ko, {}, 1.0)
	assert tuple(val.shape) == (1, 1)
	ref = O.log_marginal(g["x"], g["y"], spec, float(g["s"]))
	assert abs(float(val) - ref) / abs(ref) < TOL
	gam = torch.tensor([0.6], dtype=torch.float64, requires_grad=True)
	v = KF.log_marginal(ko, {'0': {'gamma': gam}}, 0.5)
	v.backward()
	_, gref, _ = O.log_marginal_grad(g["x"], g["y"], se_spec(0.6, float(g["kappa"])), float(g["s"]), weight=0.5)
	assert abs(float(gam.grad) - gref[0]["gamma"][0]) / abs(gref[0]["gamma"][0]) < 1e-7
	with pytest.raises(AttributeError):
		KF.optimize_params(type="bandwidth", restarts=1)          # no kernel_object: as in the reference
	KF.kernel_object = ko
	assert KF.optimize_params(type="bandwidth", restarts=1, optimizer="pytorch-minimize", maxiter=20, init_func=lambda k: torch.tensor([0.9], dtype=torch.float64)) is True
	g_opt = float(ko.params_dict['0']['gamma'].reshape(-1)[0])
	best = float(KF.log_marginal(ko, {}, 1.0))
	assert best <= float(val) + 1e-9 and g_opt > 0 and KF.fitted is True


def test_B1_beta_norm_bounds(S):
	"""beta(), norm(), lcb(), ucb() (gauss_procc.py:119-134, :179-196) against the reference's outputs."""
	g = golden("B1_beta_norm")
	GP = S.GaussianProcess(gamma=float(g["gamma"]), s=float(g["s"]), kappa=float(g["kappa"]), kernel_name="squared_exponential", d=2)
	assert GP.norm() is None
	GP.fit_gp(T(g["x"]), T(g["y"]))
	assert abs(float(GP.norm()) - g["norm"].item()) / g["norm"].item() < TOL
	assert abs(float(GP.beta()) - g["beta_default"].item()) / g["beta_default"].item() < TOL
	assert abs(float(GP.beta(delta=0.1, norm=2.0)) - g["beta_d01_n2"].item()) / g["beta_d01_n2"].item() < TOL
	assert rel_err(N(GP.lcb(T(g["x"][:5]))), g["lcb"]) < TOL and rel_err(N(GP.ucb(T(g["x"][:5]))), g["ucb"]) < TOL
	# a size where det K itself overflows a double: the factor-based form stays finite
	rng = np.random.RandomState(5)
	x = rng.uniform(-1, 1, size=(3000, 2))
	GP2 = S.GaussianProcess(gamma=0.1, s=30.0, kappa=1.0, kernel_name="squared_exponential", d=2)
	GP2.fit_gp(T(x), T(np.sin(x.sum(axis=1, keepdims=True))))
	assert np.isfinite(float(GP2.beta()))


def test_general_noise_matrix(S):
	"""fit_gp(x, y, Sigma): K = k(x,x) + Sigma^T Sigma (gauss_procc.py:151-163) for a non-diagonal Sigma."""
	import scipy.linalg as sla
	rng = np.random.RandomState(11)
	n, m = 300, 40
	x, xq = rng.uniform(-1, 1, size=(n, 3)), rng.uniform(-1, 1, size=(m, 3))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(n, 1))
	Sigma = np.triu(0.02 * rng.normal(size=(n, n)), 1) + np.diag(rng.uniform(0.2, 0.5, size=n))
	spec = [("squared_exponential", {"gamma": 0.9, "kappa": 1.3}, "-")]
	GP = S.GaussianProcess(gamma=0.9, s=0.3, kappa=1.3, kernel_name="squared_exponential", d=3)
	GP.fit_gp(T(x), T(y), Sigma=T(Sigma))
	K = O.kernel(x, x, spec) + Sigma.T @ Sigma
	assert rel_err(N(GP.get_kernel()), K) < 1e-12
	L = sla.cholesky(K, lower=True)
	alpha = sla.cho_solve((L, True), y)
	Ks = O.kernel(x, xq, spec)
	V = sla.solve_triangular(L, Ks.T, lower=True)
	mu_ref = Ks @ alpha
	std_ref = np.sqrt(1.3 - np.sum(V * V, axis=0)).reshape(-1, 1)
	mu, std = GP.mean_std(T(xq))
	assert rel_err(N(mu), mu_ref) < TOL and rel_err(N(std), std_ref) < TOL
	assert abs(float(GP.norm()) - np.sqrt(alpha.T @ O.kernel(x, x, spec) @ alpha).item()) < 1e-9
	# Sigma = s I reproduces the default path
	GP1 = S.GaussianProcess(gamma=0.9, s=0.3, kappa=1.3, kernel_name="squared_exponential", d=3)
	GP1.fit_gp(T(x), T(y))
	GP2 = S.GaussianProcess(gamma=0.9, s=0.3, kappa=1.3, kernel_name="squared_exponential", d=3)
	GP2.fit_gp(T(x), T(y), Sigma=T(0.3 * np.eye(n)))
	m1, s1 = GP1.mean_std(T(xq)); m2, s2 = GP2.mean_std(T(xq))
	assert rel_err(N(m2), N(m1)) < 1e-10 and rel_err(N(s2), N(s1)) < 1e-10


def test_edge_cases_closed_forms(S):
	"""Anchors the reference's tests lack: N = 1 closed forms, one test point,
	sizes around the 128 tile edge, coincident training points, an empty test set."""
	kappa, s, gamma = 1.7, 0.3, 0.8
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=kappa, kernel_name="squared_exponential", d=2)
	x1 = torch.tensor([[0.2, -0.4]], dtype=torch.float64)
	y1 = torch.tensor([[1.3]], dtype=torch.float64)
	GP.fit_gp(x1, y1)
	xt = torch.tensor([[0.2, -0.4], [0.5, 0.1]], dtype=torch.float64)
	mu, std = GP.mean_std(xt)
	kstar = kappa * np.exp(-0.5 * np.array([0.0, 0.3 ** 2 + 0.5 ** 2]) / gamma ** 2)
	assert np.allclose(N(mu).ravel(), kstar * 1.3 / (kappa + s * s), rtol=1e-13)
	assert np.allclose(N(std).ravel(), np.sqrt(kappa - kstar ** 2 / (kappa + s * s)), rtol=1e-13)
	assert abs(lml(GP) - (0.5 * 1.3 ** 2 / (kappa + s * s) + 0.5 * np.log(kappa + s * s))) < 1e-13
	# one test point; empty test set
	mu1, std1 = GP.mean_std(xt[:1])
	assert tuple(mu1.shape) == (1, 1) and tuple(std1.shape) == (1, 1)
	mu0, std0 = GP.mean_std(xt[:0])
	assert tuple(mu0.shape) == (0, 1) and tuple(std0.shape) == (0, 1)
	# sizes around the tile edge, with two coincident training points (K is singular without the noise term)
	rng = np.random.RandomState(3)
	for n in (2, 127, 128, 129, 257):
		x = rng.uniform(-1, 1, size=(n, 2))
		x[-1] = x[0]
		y = np.sin(x.sum(axis=1, keepdims=True))
		xq = rng.uniform(-1, 1, size=(5, 2))
		spec = [("squared_exponential", {"gamma": gamma, "kappa": kappa}, "-")]
		GP = S.GaussianProcess(gamma=gamma, s=s, kappa=kappa, kernel_name="squared_exponential", d=2)
		GP.fit_gp(T(x), T(y))
		mu, std = GP.mean_std(T(xq))
		L, alpha = O.fit(x, y, spec, s)
		muo, stdo = O.mean_std(x, L, alpha, xq, spec)
		assert rel_err(N(mu), muo) < TOL and rel_err(N(std), stdo) < TOL, n
		assert abs(lml(GP) - O.log_marginal(x, y, spec, s)) / abs(O.log_marginal(x, y, spec, s)) < TOL, n
	# interpolation limit: s -> 0 reproduces the training targets
	x = rng.uniform(-1, 1, size=(40, 2))
	y = np.sin(x.sum(axis=1, keepdims=True))
	GP = S.GaussianProcess(gamma=0.5, s=1e-6, kappa=1.0, kernel_name="squared_exponential", d=2)
	GP.fit_gp(T(x), T(y))
	mu, std = GP.mean_std(T(x))
	assert np.abs(N(mu) - y).max() < 1e-6 and N(std).max() < 1e-4


def test_not_positive_definite_raises(S):
	x = torch.zeros((300, 2), dtype=torch.float64)        # 300 identical points, no noise -> singular
	GP = S.GaussianProcess(gamma=1.0, s=0.0, kappa=1.0, kernel_name="squared_exponential", d=2)
	with pytest.raises(torch.linalg.LinAlgError):
		GP.fit_gp(x, torch.zeros((300, 1), dtype=torch.float64))
	with pytest.raises(NotImplementedError):
		S.GaussianProcess(loss="huber")


# --------------------------------------------------------------------------------------------
# larger sizes: oracle where it finishes in seconds, properties beyond
# --------------------------------------------------------------------------------------------

def synth(n, d, m, seed=1234):
	"""SURVEY.md section 8d generator: x ~ U(-1,1)^{n x d}, y = sin(sum x) + 0.1 N(0,1), xtest ~ U(-1,1)."""
	gx = torch.Generator().manual_seed(seed)
	x = torch.rand(n, d, generator=gx, dtype=torch.float64) * 2 - 1
	gy = torch.Generator().manual_seed(seed + 1)
	y = torch.sin(x.sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=gy, dtype=torch.float64)
	gt = torch.Generator().manual_seed(seed + 2)
	xt = torch.rand(m, d, generator=gt, dtype=torch.float64) * 2 - 1
	return x, y, xt


@pytest.mark.parametrize("kernel_name,nu", [("squared_exponential", None), ("matern", 2.5)])
def test_mid_size_vs_oracle(S, kernel_name, nu):
	n, d, m = 4096, 8, 512
	x, y, xt = synth(n, d, m)
	gamma, s = float(np.sqrt(d)), 0.1
	kw = dict(nu=nu) if nu else {}
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name=kernel_name, d=d, **kw)
	GP.fit_gp(x.cuda(), y.cuda())
	mu, std = GP.mean_std(xt.cuda())
	spec = [(kernel_name, dict(gamma=gamma, kappa=1.0, **kw), "-")]
	L, alpha = O.fit(x.numpy(), y.numpy(), spec, s)
	mu_o, std_o = O.mean_std(x.numpy(), L, alpha, xt.numpy(), spec)
	assert rel_err(N(mu), mu_o) < TOL and rel_err(N(std), std_o) < TOL
	assert rel_err(N(GP.A), alpha) < 1e-6
	lm_o = O.log_marginal(x.numpy(), y.numpy(), spec, s)[0, 0]
	assert abs(lml(GP) - lm_o) / abs(lm_o) < TOL


@pytest.mark.parametrize("n,m", [(3001, 777), (2177, 129), (129, 1)])
def test_ragged_sizes_vs_oracle(S, n, m):
	"""N and M off the 128 tile: the estimator holds the factor (and K*) at the next tile multiple, bordered by
	an identity block / zero rows; mean, std, full covariance, alpha, norm, evidence and its gradient must not see it."""
	d = 5
	x, y, xt = synth(n, d, m, seed=77)
	gamma, s = 1.7, 0.2
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.3, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x.cuda(), y.cuda())
	assert GP._L.shape[0] % 128 == 0 and GP._L.shape[0] >= n and GP.A.shape == (n, 1)
	mu, std = GP.mean_std(xt.cuda())
	assert mu.shape == (m, 1) and std.shape == (m, 1)
	spec = [("squared_exponential", dict(gamma=gamma, kappa=1.3), "-")]
	L, alpha = O.fit(x.numpy(), y.numpy(), spec, s)
	mu_o, std_o = O.mean_std(x.numpy(), L, alpha, xt.numpy(), spec)
	assert rel_err(N(mu), mu_o) < TOL and rel_err(N(std), std_o) < TOL
	assert rel_err(N(GP.A), alpha) < 1e-7
	mu_f, cov = GP.mean_std(xt.cuda(), full=True)
	_, cov_o = O.mean_cov(x.numpy(), L, alpha, xt.numpy(), spec)
	assert cov.shape == (m, m) and rel_err(N(cov), cov_o) < TOL and rel_err(N(mu_f), mu_o) < TOL
	assert rel_err(N(GP.norm()), O.norm(x.numpy(), alpha, spec)) < 1e-7
	lm_o = O.log_marginal(x.numpy(), y.numpy(), spec, s)[0, 0]
	assert abs(lml(GP) - lm_o) / abs(lm_o) < TOL
	# gradient of the evidence w.r.t. the lengthscale against torch autograd of the same formula on the CPU
	g = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
	f = GP.log_marginal(GP.kernel_object, {'0': {'gamma': g}}, 1.0)
	f.backward()
	gr = torch.tensor(gamma, dtype=torch.float64, requires_grad=True)
	fr = _torch_lml(x, y, s, 1.0, "se", gr, 1.3)
	fr.backward()
	assert abs(float(f) - float(fr)) / abs(float(fr)) < 1e-9
	assert abs(float(g.grad) - float(gr.grad)) / abs(float(gr.grad)) < 1e-7


def test_config2_properties(S):
	"""BASELINE config 2: N=16 384, d=8 SE, fp64.  Size-independent checks: (K+s^2 I) alpha = y,
	interpolation-consistency of mean(x) = K alpha, 0 <= sigma <= sqrt(kappa), a 2 048-point oracle
	sub-problem, and linearity of the posterior mean in y."""
	n, d, m = 16384, 8, 4096
	x, y, xt = synth(n, d, m)
	gamma, s = float(np.sqrt(d)), 0.1
	xd, yd, xtd = x.cuda(), y.cuda(), xt.cuda()
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(xd, yd)
	mu, std = GP.mean_std(xtd)
	K = GP.K
	r = K @ GP.A - yd
	assert float(torch.norm(r) / torch.norm(yd)) < 1e-10
	del K
	assert bool(torch.all(std >= 0)) and bool(torch.all(std <= 1.0 + 1e-12)) and not bool(torch.isnan(std).any())
	assert rel_err(N(GP.mean(xtd)), N(mu)) < 1e-9
	# linearity: fit on 2y - 3 -> mean = 2 mu - 3 * K* K^-1 1; check through a second right-hand side
	GP2 = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP2.fit_gp(xd, 2.0 * yd)
	mu2, std2 = GP2.mean_std(xtd)
	assert rel_err(N(mu2), 2 * N(mu)) < 1e-10 and rel_err(N(std2), N(std)) < 1e-12
	# oracle on a sub-problem of the same data
	ns = 2048
	GPs = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GPs.fit_gp(xd[:ns], yd[:ns])
	mus, stds = GPs.mean_std(xtd[:256])
	spec = [("squared_exponential", {"gamma": gamma, "kappa": 1.0}, "-")]
	Lo, ao = O.fit(x[:ns].numpy(), y[:ns].numpy(), spec, s)
	mo, so = O.mean_std(x[:ns].numpy(), Lo, ao, xt[:256].numpy(), spec)
	assert rel_err(N(mus), mo) < TOL and rel_err(N(stds), so) < TOL


def test_headline_size_properties(S):
	"""The benchmarked configuration itself (N = 65 536, d = 16, M = 4096, fp64): size-independent identities.
	mean(x_i) = y_i - s^2 alpha_i on training points (i.e. (K + s^2 I) alpha = y, through the prediction path),
	0 <= sigma <= sqrt(kappa), sigma at training points below the prior, and linearity of the mean in y."""
	n, d, m = 65536, 16, 4096
	x, y, xt = synth(n, d, m)
	gamma, s = float(np.sqrt(d)), 0.1
	xd, yd, xtd = x.cuda(), y.cuda(), xt.cuda()
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(xd, yd)
	mu, std = GP.mean_std(xtd)
	assert not bool(torch.isnan(mu).any()) and not bool(torch.isnan(std).any())
	assert bool(torch.all(std >= 0)) and bool(torch.all(std <= 1.0 + 1e-12))
	idx = torch.arange(0, n, n // 2048, device="cuda")[:2048]
	mu_tr, std_tr = GP.mean_std(xd[idx])
	alpha = GP.A.reshape(-1, 1).cuda()
	expect = yd[idx] - s * s * alpha[idx]
	assert float(torch.norm(mu_tr - expect) / torch.norm(expect)) < 1e-8
	assert float(std_tr.max()) < 1.0 and float(std_tr.mean()) < float(std.mean()) + 1e-12
	lm = lml(GP)
	assert np.isfinite(lm)
	GP2 = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP2.fit_gp(xd, -3.0 * yd)
	mu2, std2 = GP2.mean_std(xtd)
	assert rel_err(N(mu2), -3.0 * N(mu)) < 1e-9 and rel_err(N(std2), N(std)) < 1e-10


def test_fp32_mode(S):
	"""fp32 is a build-only precision mode (the reference is fp64-only); oracle = fp64 on the same
	(up-cast) inputs, tolerance 1e-3 with s >= 0.3 (hard parts)."""
	n, d, m = 2048, 16, 256
	x, y, xt = synth(n, d, m)
	x32, y32, xt32 = x.float(), y.float(), xt.float()
	gamma, s = float(np.sqrt(d)), 0.3
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="matern", nu=2.5, d=d)
	GP.fit_gp(x32.cuda(), y32.cuda())
	mu, std = GP.mean_std(xt32.cuda())
	assert mu.dtype == torch.float32
	spec = [("matern", {"gamma": gamma, "nu": 2.5, "kappa": 1.0}, "-")]
	L, alpha = O.fit(x32.double().numpy(), y32.double().numpy(), spec, s)
	mu_o, std_o = O.mean_std(x32.double().numpy(), L, alpha, xt32.double().numpy(), spec)
	assert rel_err(N(mu).astype(np.float64), mu_o) < 1e-3 and rel_err(N(std).astype(np.float64), std_o) < 1e-3
	lm_o = O.log_marginal(x32.double().numpy(), y32.double().numpy(), spec, s)[0, 0]
	assert abs(lml(GP) - lm_o) / abs(lm_o) < 1e-3


def test_log_marginal_tracks_changed_hyperparameters(S):
	"""The reference rebuilds K from the CURRENT self.s / params_dict on every log_marginal call (gauss_procc.py:631-638).
	The resident factor may only be reused while neither has changed since fit_gp."""
	n, d = 700, 3
	x, y, _ = synth(n, d, 4, seed=5)
	GP = S.GaussianProcess(gamma=1.3, s=0.2, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x.cuda(), y.cuda())
	spec = [("squared_exponential", {"gamma": 1.3, "kappa": 1.0}, "-")]
	v0 = lml(GP)
	assert abs(v0 - O.log_marginal(x.numpy(), y.numpy(), spec, 0.2)[0, 0]) / abs(v0) < TOL
	GP.s = 0.35                                            # noise changed after the fit
	v1 = lml(GP)
	assert abs(v1 - O.log_marginal(x.numpy(), y.numpy(), spec, 0.35)[0, 0]) / abs(v1) < TOL
	GP.kernel_object.params_dict['0']['gamma'] = 0.9       # kernel parameter edited in place after the fit
	spec2 = [("squared_exponential", {"gamma": 0.9, "kappa": 1.0}, "-")]
	v2 = lml(GP)
	assert abs(v2 - O.log_marginal(x.numpy(), y.numpy(), spec2, 0.35)[0, 0]) / abs(v2) < TOL
	GP.fit_gp(x.cuda(), y.cuda())                          # refit: the new factor is current again and is reused
	L_before = GP._L.data_ptr()
	assert abs(lml(GP) - v2) / abs(v2) < 1e-12 and GP._L.data_ptr() == L_before


def test_failed_refit_leaves_the_object_unfitted(S):
	"""fit_gp clears `fitted` before factoring; a refit that is not positive definite must not leave fitted=True behind."""
	n, d = 300, 2
	x, y, xt = synth(n, d, 16, seed=6)
	GP = S.GaussianProcess(gamma=1.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x.cuda(), y.cuda())
	assert GP.fitted
	GP.kernel_object.params_dict['0']['kappa'] = -1.0       # k(x,x) + s^2 I is no longer positive definite
	with pytest.raises(torch.linalg.LinAlgError):
		GP.fit_gp(x.cuda(), y.cuda())
	assert GP.fitted is False and GP._L is None
	mu, sd = GP.mean_std(xt.cuda())                        # prior branch, no crash inside the C ABI
	assert float(mu.abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------- quadrature embeddings
from tests.test_oracle_golden import Q1_CASES, _q1_build, _q1_x          # noqa: E402


@pytest.mark.parametrize("tag", Q1_CASES)
def test_Q1_quadrature_embed(S, tag):
	"""QuadratureEmbedding.embed and its derived classes (embedding.py:450-466) on the device against the reference's outputs:
	fp64 at 1e-13, the transposed form, CPU-in/CPU-out and GPU-in/GPU-out, and fp32 against the fp64 oracle."""
	g = golden("Q1_quadrature")
	emb = _q1_build(tag)
	x = _q1_x(g, tag)
	for cuda in (False, True):
		z = emb.embed(T(x, cuda))
		assert z.is_cuda == cuda and tuple(z.shape) == g[tag + "_z"].shape
		assert rel_err(N(z), g[tag + "_z"]) < 1e-13
	zt = emb.embed_t(T(x, True))
	assert tuple(zt.shape) == g[tag + "_z"].T.shape and rel_err(N(zt), g[tag + "_z"].T) < 1e-13
	z32 = emb.embed(T(x, True).float())
	assert z32.dtype == torch.float32
	ref = O.quadrature_embed(x.astype(np.float32).astype(np.float64), emb.W.float().double().numpy(), emb.weights.numpy(), kappa=emb.kappa, cosine=emb.cosine)
	assert np.abs(N(z32) - ref).max() < 2e-5 * np.abs(ref).max()


def test_Q1_quadrature_embed_larger_shapes(S):
	"""the same entry point at sizes that leave the single-tile regime (n off the tile, m = 2 * 24^2 = 1152 features, d = 2), against the oracle"""
	import stpy_amd.embeddings.embedding as E
	rng = np.random.RandomState(3)
	x = rng.uniform(-1, 1, size=(3001, 2))
	emb = E.HermiteEmbedding(gamma=0.3, m=2 * 24 ** 2, d=2, kappa=1.7)
	z = emb.embed(T(x, True))
	ref = O.quadrature_embed(x, emb.W.numpy(), emb.weights.numpy(), kappa=1.7)
	assert rel_err(N(z), ref) < 1e-13
	embc = E.QuadratureEmbedding(gamma=0.3, m=530, d=2, cosine=True)          # q = 23: 529 features, an odd count
	assert embc.get_m() == 529
	zc = embc.embed(T(x, True))
	assert rel_err(N(zc), O.quadrature_embed(x, embc.W.numpy(), embc.weights.numpy(), cosine=True)) < 1e-13
	zc32 = embc.embed(T(x, True).float())
	refc = O.quadrature_embed(x.astype(np.float32).astype(np.float64), embc.W.float().double().numpy(), embc.weights.numpy(), cosine=True)
	assert np.abs(N(zc32) - refc).max() < 2e-5 * np.abs(refc).max()


def test_G13_kernelized_features_on_hermite(S):
	"""the tutorial's comparison (tutorials/fourier-features.ipynb): primal ridge on Hermite quadrature features against the
	reference's own numbers for it, and the exact GP it approximates (both from the reference)"""
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	import stpy_amd.embeddings.embedding as E
	g = golden("G13_hermite_features")
	d = g["x"].shape[1]
	emb = E.HermiteEmbedding(gamma=float(g["gamma"]), m=int(g["m"]), d=d, kappa=float(g["kappa"]))
	assert emb.get_m() == int(g["m"])
	KF = KernelizedFeatures(embedding=emb, m=emb.get_m(), s=float(g["s"]), lam=float(g["lam"]), d=d)
	KF.fit_gp(T(g["x"], True), T(g["y"], True))
	mu, std = KF.mean_std(T(g["xtest"], True))
	assert rel_err(N(mu), g["mu"]) < TOL and rel_err(N(std), g["std"]) < 1e-7
	GP = S.GaussianProcess(gamma=float(g["gamma"]), s=float(g["s"]), kappa=float(g["kappa"]), kernel_name="squared_exponential", d=d)
	GP.fit_gp(T(g["x"], True), T(g["y"], True))
	mu_gp, std_gp = GP.mean_std(T(g["xtest"], True))
	assert rel_err(N(mu_gp), g["mu_exact_gp"]) < TOL and rel_err(N(std_gp), g["std_exact_gp"]) < TOL


def test_estimator_base_and_general_driver(S):
	"""GaussianProcess derives from Estimator as in the reference (gauss_procc.py:18) and Estimator.optimize_params_general
	(estimator.py:42-257) drives the device evidence directly from a ``params`` dictionary: the steepest-descent branch
	("pymanopt"; written out here when the package is absent), the L-BFGS branch with the noise as a second variable, and the
	reference's bisection on one bounded scalar."""
	from stpy_amd.estimator import Estimator, Euclidean
	assert issubclass(S.GaussianProcess, Estimator)
	rng = np.random.RandomState(18)
	n, d = 180, 2
	x = torch.from_numpy(rng.uniform(-1, 1, size=(n, d)))
	y = torch.sin(3 * x[:, :1]) * torch.cos(2 * x[:, 1:2]) + 0.1 * torch.from_numpy(rng.normal(size=(n, 1)))

	def torch_opt(noise):
		import scipy.optimize

		def fun(v):
			t = torch.tensor(v, dtype=torch.float64, requires_grad=True)
			f = _torch_lml(x, y, t[1] if noise else 0.1, 1.0, "se", t[0], 1.0)
			f.backward()
			return float(f.detach()), t.grad.numpy()
		return scipy.optimize.minimize(fun, np.array([0.3, 0.1] if noise else [0.3]), jac=True, method='L-BFGS-B', options={'gtol': 1e-6, 'ftol': 1e-14})
	ref = torch_opt(False)
	GP = S.GaussianProcess(gamma=0.3, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x, y)
	ok = GP.optimize_params_general(params={'0': {'gamma': (lambda k: np.full(k, 0.3), Euclidean(1), None)}}, restarts=1, optimizer="pymanopt",
									maxiter=300, mingradnorm=1e-5)
	assert ok is True and GP.fitted and GP.back_prop is False
	assert abs(lml(GP) - ref.fun) / abs(ref.fun) < 1e-7
	assert abs(abs(float(GP.kernel_object.params_dict['0']['gamma'].reshape(-1)[0])) - abs(ref.x[0])) / abs(ref.x[0]) < 1e-3
	# two variables: lengthscale + noise (each reads its OWN slice of x)
	ref2 = torch_opt(True)
	GP2 = S.GaussianProcess(gamma=0.3, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP2.fit_gp(x, y)
	GP2.optimize_params(type="bandwidth+noise", restarts=1, optimizer="pytorch-minimize", init_func=lambda k: np.full(k, 0.3), maxiter=300, mingradnorm=1e-6)
	assert abs(lml(GP2) - ref2.fun) / abs(ref2.fun) < 1e-6
	assert abs(abs(float(torch.as_tensor(GP2.s).reshape(-1)[0])) - abs(ref2.x[1])) / abs(ref2.x[1]) < 1e-2
	# bisection (reference semantics: root of the cost on [a, b]); the evidence of this problem is negative at both ends -> 'stop' returns a
	GP3 = S.GaussianProcess(gamma=0.3, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP3.fit_gp(x, y)
	fa = float(GP3.log_marginal(GP3.kernel_object, {'0': {'gamma': torch.tensor([0.2]).double()}}, 1.0))
	GP3.optimize_params_general(params={'0': {'gamma': (None, Euclidean(1), (0.2, 2.0))}}, restarts=1, optimizer="bisection")
	g3 = float(GP3.kernel_object.params_dict['0']['gamma'].reshape(-1)[0])
	assert (fa < 0 and g3 == 0.2) or abs(lml(GP3)) < 1e-6 * max(1.0, abs(fa))
	with pytest.raises(AssertionError):
		GP3.optimize_params_general(params={'0': {'gamma': (None, Euclidean(1), None)}}, optimizer="nope")


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-8), (torch.float32, 2e-3)])
def test_kernelized_features_streaming_vs_oracle(S, dtype, tol):
	"""KernelizedFeatures.fit_gp streams Phi in row slabs (kernelized_features.py:228-240 without the n x m matrix): N = 20 000,
	m = 512 random Fourier features, a slab budget small enough for eight slabs (aligned ones and a ragged last one), against the
	oracle's one-shot normal equations; and the slab size must not change the result beyond rounding."""
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	rng = np.random.RandomState(77)
	n, d, m, M = 20000, 6, 512, 300
	x = rng.uniform(-1, 1, size=(n, d))
	y = np.sin(x[:, :1] * 2) + x[:, 1:2] * x[:, 2:3] + 0.1 * rng.normal(size=(n, 1))
	xt = rng.uniform(-1, 1, size=(M, d))
	W = rng.normal(size=(m, d)) / 1.5
	s, lam, kappa = 0.3, 1.0, 1.2
	emb = S.RFFEmbedding(gamma=1.5, m=m, d=d, kappa=kappa)
	emb.W = torch.from_numpy(W)
	Q = O.rff_embed(x, W, m, kappa=kappa)
	V_o, invV, theta = O.kernelized_features_fit(Q, y, s, lam)
	mu_o, std_o = O.kernelized_features_mean_std(O.rff_embed(xt, W, m, kappa=kappa), invV, theta, s)
	outs = []
	for slab in (m * 8 * 2560 if dtype == torch.float64 else m * 4 * 2560, 2 << 30):
		KF = KernelizedFeatures(embedding=emb, m=m, s=s, lam=lam, d=d)
		KF.slab_bytes = slab
		KF.fit_gp(T(x).to(dtype).cuda(), T(y).to(dtype).cuda())
		mu, std = KF.mean_std(T(xt).to(dtype).cuda())
		assert mu.dtype == dtype and tuple(mu.shape) == (M, 1)
		assert rel_err(N(mu), mu_o) < tol and rel_err(N(std), std_o) < tol
		assert rel_err(N(KF.theta_mean()), theta) < tol * 100
		if dtype == torch.float64:
			assert rel_err(N(KF.V), V_o) < 1e-10
		outs.append((N(mu), N(std)))
	assert rel_err(outs[0][0], outs[1][0]) < (1e-10 if dtype == torch.float64 else 1e-3)


def test_kernelized_features_add_data_point_extends_the_normal_equations(S):
	"""kernelized_features.py:107-112: new rows extend the accumulated Phi^T Phi / Phi^T y (their embedding + one `+=` product + the
	m x m refactorisation) -- the same estimator as a fit on all rows, and as the oracle's one-shot normal equations."""
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	rng = np.random.RandomState(78)
	n, d, m, M = 3000, 4, 256, 100
	x = rng.uniform(-1, 1, size=(n, d))
	y = np.cos(x[:, :1] * 2) + 0.1 * rng.normal(size=(n, 1))
	xt = rng.uniform(-1, 1, size=(M, d))
	W = rng.normal(size=(m, d))
	emb = S.RFFEmbedding(gamma=1.0, m=m, d=d)
	emb.W = torch.from_numpy(W)
	KF = KernelizedFeatures(embedding=emb, m=m, s=0.2, lam=1.0, d=d)
	KF.add_data_point(T(x[:1000], True), T(y[:1000], True))          # first call = fit
	KF.add_data_point(T(x[1000:1003], True), T(y[1000:1003], True))   # a handful of rows
	KF.add_data_point(T(x[1003:], True), T(y[1003:], True))
	assert KF.fitted is False and len(KF.to_add) == 2          # queued as in the reference (:108-113), folded in by the next prediction
	mu, std = KF.mean_std(T(xt, True))
	assert KF.n == n and tuple(KF.x.shape) == (n, d) and KF.fitted is True and KF.to_add == []
	KF2 = KernelizedFeatures(embedding=emb, m=m, s=0.2, lam=1.0, d=d)
	KF2.fit_gp(T(x, True), T(y, True))
	mu2, std2 = KF2.mean_std(T(xt, True))
	assert rel_err(N(mu), N(mu2)) < 1e-10 and rel_err(N(std), N(std2)) < 1e-10
	Q = O.rff_embed(x, W, m)
	_, invV, theta = O.kernelized_features_fit(Q, y, 0.2, 1.0)
	mu_o, std_o = O.kernelized_features_mean_std(O.rff_embed(xt, W, m), invV, theta, 0.2)
	assert rel_err(N(mu), mu_o) < 1e-8 and rel_err(N(std), std_o) < 1e-8
