"""
The BASELINE.json configurations at their FULL sizes on the GPU (run with `-m gpu` on an MI355X), each against the
oracle where the oracle finishes in seconds, against the validated fp64 HIP path where it does not, and through
size-independent properties:

  C2  N = 16 384, d = 8, SE, fp64           : mu, sigma, log-marginal directly against the CPU oracle (1e-8)
  C3  N = 65 536, d = 16, Matern-5/2, fp32  : fit + mean_std + log_marginal against the fp64 HIP path on the same inputs
                                               (1e-3, the tolerance SURVEY.md section 8d states for the build-only fp32
                                               mode; measured 1.7e-4 / 5e-5 / 1.1e-4) and both against the oracle on a
                                               4096-point sub-problem
  C5  RFF N = 262 144, d = 64, m = 32 768, fp32 : sampled rows of the first / middle / last row blocks, every one of the
                                               eight W-eighths (all columns), against the oracle (2e-5 of the amplitude)
  KF  KernelizedFeatures N = 262 144, d = 64, m = 8192, fp32 (the bench's streaming shape): sampled entries of Phi^T Phi and Phi^T y
      against fp64 sums over all rows, the normal-equation residual
  C*  the headline N = 65 536 fp64: test_gpu_gp.py::test_headline_size_properties (identities) and, here, the FACTOR itself against
      the oracle: its leading 16 384 x 16 384 block against the oracle's Cholesky of the leading sub-Gram, and sampled rows of
      L L^T - (K + s^2 I) from the trailing part
  C4  N = 131 072, d = 32, SE, fp64 at full size on ONE GPU (137 GB in place): the single-GPU class and the block-cyclic
      schedule forced onto one rank, against each other and through the training-point identity / linearity / bounds.
      (The 8-GPU run of the same shape needs a node this test box does not have; its schedule is covered by the gloo tests
      and by test_gpu_block_cyclic.py at reduced size.)
"""
import math

import numpy as np
import pytest
import torch

from oracle import gp_oracle as O
from tests.conftest import rel_err
from tests.test_gpu_gp import N, lml, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S(gpu_device):
	import stpy_amd
	return stpy_amd


def _free():
	import gc
	gc.collect()
	torch.cuda.empty_cache()


def test_config2_full_size_vs_oracle(S):
	"""BASELINE config 2 at its own size, directly against the CPU oracle (about 15 s of LAPACK on the box's host cores)."""
	n, d, m = 16384, 8, 4096
	x, y, xt = synth(n, d, m)
	gamma, s = float(np.sqrt(d)), 0.1
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x.cuda(), y.cuda())
	mu, std = GP.mean_std(xt.cuda())
	lm = lml(GP)
	alpha = N(GP.A)
	del GP
	_free()
	spec = [("squared_exponential", {"gamma": gamma, "kappa": 1.0}, "-")]
	xn, yn, xtn = x.numpy(), y.numpy(), xt.numpy()
	L, alpha_o = O.fit(xn, yn, spec, s)
	mu_o, std_o = O.mean_std(xn, L, alpha_o, xtn, spec)
	import scipy.linalg as sla          # log-marginal from the factor already in hand (no second Cholesky): gauss_procc.py:631-638
	zo = sla.solve_triangular(L, yn.reshape(-1, 1), lower=True, check_finite=False)
	lm_o = 0.5 * float((zo.T @ zo)[0, 0]) + 0.5 * 2.0 * float(np.sum(np.log(np.diag(L))))
	e_mu, e_sd, e_lm = rel_err(N(mu), mu_o), rel_err(N(std), std_o), abs(lm - lm_o) / abs(lm_o)
	print("C2 full size vs oracle: mu %.2e  sigma %.2e  lml %.2e  alpha %.2e" % (e_mu, e_sd, e_lm, rel_err(alpha, alpha_o)))
	assert e_mu < 1e-8 and e_sd < 1e-8 and e_lm < 1e-8
	assert rel_err(alpha, alpha_o) < 1e-6


def test_config3_full_size_fp32(S):
	"""BASELINE config 3 (N = 65 536, d = 16, Matern-5/2, fp32, + log_marginal) at full size."""
	n, d, m = 65536, 16, 4096
	x, y, xt = synth(n, d, m)
	x32, y32, xt32 = x.float(), y.float(), xt.float()
	x64, y64, xt64 = x32.double(), y32.double(), xt32.double()          # the same (fp32-representable) inputs in both precisions
	gamma, s = float(np.sqrt(d)), 0.3
	kw = dict(gamma=gamma, s=s, kappa=1.0, kernel_name="matern", nu=2.5, d=d)
	G64 = S.GaussianProcess(**kw)
	G64.fit_gp(x64.cuda(), y64.cuda())
	mu64, sd64 = G64.mean_std(xt64.cuda())
	lm64 = lml(G64)
	a64 = N(G64.A)
	# the fp64 reference itself: training-point identity through the prediction path, (K + s^2 I) alpha = y
	idx = torch.arange(0, n, n // 1024)[:1024]
	mu_tr, _ = G64.mean_std(x64[idx].cuda())
	expect = y64[idx].numpy() - s * s * a64[idx.numpy()]
	assert rel_err(N(mu_tr), expect) < 1e-8
	del G64
	_free()
	G32 = S.GaussianProcess(**kw)
	G32.fit_gp(x32.cuda(), y32.cuda())
	mu32, sd32 = G32.mean_std(xt32.cuda())
	assert mu32.dtype == torch.float32 and sd32.dtype == torch.float32
	lm32 = lml(G32)
	e_mu, e_sd, e_lm = rel_err(N(mu32), N(mu64)), rel_err(N(sd32), N(sd64)), abs(lm32 - lm64) / abs(lm64)
	print("C3 full size fp32 vs fp64: mu %.2e  sigma %.2e  lml %.2e  alpha %.2e" % (e_mu, e_sd, e_lm, rel_err(N(G32.A), a64)))
	assert not bool(torch.isnan(sd32).any())
	assert e_mu < 1e-3 and e_sd < 1e-3 and e_lm < 1e-3
	assert rel_err(N(G32.A), a64) < 5e-3
	del G32
	_free()
	# both precisions against the oracle on a 4096-point sub-problem of the same data
	ns, ms = 4096, 512
	spec = [("matern", {"gamma": gamma, "nu": 2.5, "kappa": 1.0}, "-")]
	xs, ys, xts = x64[:ns].numpy(), y64[:ns].numpy(), xt64[:ms].numpy()
	L, alpha_o = O.fit(xs, ys, spec, s)
	mu_o, sd_o = O.mean_std(xs, L, alpha_o, xts, spec)
	lm_o = O.log_marginal(xs, ys, spec, s)[0, 0]
	for dt, tol in ((torch.float64, 1e-8), (torch.float32, 1e-3)):
		G = S.GaussianProcess(**kw)
		G.fit_gp(x64[:ns].to(dt).cuda(), y64[:ns].to(dt).cuda())
		mu, sd = G.mean_std(xt64[:ms].to(dt).cuda())
		assert rel_err(N(mu), mu_o) < tol and rel_err(N(sd), sd_o) < tol and abs(lml(G) - lm_o) / abs(lm_o) < tol


def test_config5_full_size_rff(S):
	"""BASELINE config 5: RFF embed N = 262 144, d = 64, m = 32 768, fp32 (34 GB of output) through RFFEmbedding.embed.
	Rows from the first, a middle and the last 128-row block are compared over ALL columns (so all eight W-eighths of the
	streaming kernel's blockIdx % 8 split and both the cos and the sin half), plus a strided sample of rows."""
	n, d, m = 262144, 64, 32768
	g = torch.Generator().manual_seed(1237)
	x = torch.rand(n, d, generator=g, dtype=torch.float32)
	np.random.seed(1237)
	emb = S.RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
	emb.W = emb.W.float().cuda()
	W32 = emb.W.double().cpu().numpy()                   # the fp32-rounded frequencies the device actually uses
	z = emb.embed(x.cuda())
	assert tuple(z.shape) == (n, m) and z.dtype == torch.float32
	amp = math.sqrt(2.0 / m)
	rows = np.concatenate([np.arange(0, 128), np.arange(n // 2 - 64, n // 2 + 64), np.arange(n - 128, n), np.arange(777, n, n // 61)])
	zs = z[torch.from_numpy(rows).cuda()].cpu().numpy().astype(np.float64)
	ref = O.rff_embed(x[rows].double().numpy(), W32, m)
	err = np.abs(zs - ref)
	print("C5 full size: max abs err %.2e of amplitude %.2e (%.2e relative); per W-eighth max %s"
		  % (err.max(), amp, err.max() / amp, ["%.1e" % err[:, k * (m // 8):(k + 1) * (m // 8)].max() for k in range(8)]))
	assert err.max() < 2e-5 * amp
	for k in range(8):          # every W-eighth was written with the right frequencies
		assert err[:, k * (m // 8):(k + 1) * (m // 8)].max() < 2e-5 * amp
	# every element written: a full-matrix reduction (sum of squares of a row = kappa up to the cos^2 + sin^2 pairing of different frequencies)
	assert float(z.abs().max()) <= amp * (1 + 1e-5)
	rs = (z[::4096].double() ** 2).sum(dim=1)
	assert float((rs - 1.0).abs().max()) < 0.05
	del z
	_free()


def test_kernelized_features_full_shape_properties(S):
	"""The bench's KernelizedFeatures shape at FULL size (N = 262 144, d = 64, m = 8192 random Fourier features, fp32, four 2 GiB slabs):
	the accumulated normal equations and their solution through size-independent properties, checked with torch in fp64 (checker only):
	  * sampled entries of V_acc = Phi^T Phi (lower tiles; rows from every slab, feature indices from the first / a middle / the last
	    128-tile) against an fp64 sum over ALL rows of the embedding recomputed row block by row block;
	  * Phi^T y likewise;
	  * the solution: || (V_acc_sym + s^2 lam I) theta - Phi^T y || / || Phi^T y ||  (the m x m fp32 factorisation + both vector solves)."""
	from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures
	n, d, m = 262144, 64, 8192
	x = torch.rand(n, d, generator=torch.Generator().manual_seed(1238), dtype=torch.float32).cuda()
	y = (torch.sin(x[:, :4].sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=torch.Generator().manual_seed(1239), dtype=torch.float32).cuda())
	np.random.seed(1238)
	emb = S.RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
	emb.W = emb.W.float()
	kf = KernelizedFeatures(embedding=emb, m=m, s=1.0, lam=1.0, d=d)
	kf.fit_gp(x, y)
	V = kf._Vacc
	assert tuple(V.shape) == (m, m) and V.dtype == torch.float32
	fi = torch.tensor([0, 1, 127, 128, 4095, 4096, 4100, 8063, 8064, 8191], device="cuda:0")
	acc = torch.zeros(len(fi), len(fi), dtype=torch.float64, device="cuda:0")
	rhs = torch.zeros(len(fi), dtype=torch.float64, device="cuda:0")
	for r0 in range(0, n, 32768):
		ph = emb.embed(x[r0:r0 + 32768])[:, fi].double()          # (rows, 10)
		acc += ph.T @ ph
		rhs += (ph * y[r0:r0 + 32768].double()).sum(dim=0)
		del ph
	got = V[fi][:, fi].double()
	low = torch.ones(len(fi), len(fi), device="cuda:0").tril().bool()          # (fi is increasing: i >= j is on or below the diagonal tile)
	scale = float(acc.diagonal().max())
	e_v = float((got - acc)[low].abs().max()) / scale
	e_r = float((kf._rhs.reshape(-1)[fi].double() - rhs).abs().max()) / max(float(rhs.abs().max()), 1.0)
	# the solution against the symmetrised accumulator in fp64
	Vs = torch.tril(V).double()
	Vs = Vs + torch.tril(Vs, -1).T + torch.eye(m, dtype=torch.float64, device="cuda:0") * (kf.s ** 2 * kf.lam)
	th = kf._theta.reshape(-1).double()
	b = kf._rhs.reshape(-1).double()
	e_s = float(torch.linalg.norm(Vs @ th - b) / torch.linalg.norm(b))
	print("KF full shape: V entries %.2e of the diagonal, Phi^T y %.2e, normal-equation residual %.2e" % (e_v, e_r, e_s))
	assert e_v < 2e-6 and e_r < 2e-6 and e_s < 2e-5          # (measured 4.5e-7 / 7e-8 / 3.6e-7)
	del Vs
	_free()


def test_headline_leading_block_vs_oracle(S):
	"""The benchmarked size against the ORACLE, not only through identities.  Right-looking Cholesky never revisits finished
	columns, so L[:m, :m] of the N = 65 536 factor must equal the Cholesky factor of the leading m x m block of K + s^2 I: compared
	with the oracle's factor at m = 16 384 (about 15 s of LAPACK).  The trailing part is checked through the defining identity on
	64 sampled rows of the last 8192: (L L^T)[i, :i+1] against the oracle's kernel row + s^2 at the diagonal (torch on the GPU is
	only the checker's matrix product here)."""
	n, d, m = 65536, 16, 16384
	x, y, _ = synth(n, d, 16)
	gamma, s = float(np.sqrt(d)), 0.1
	GP = S.GaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(x.cuda(), y.cuda())
	L = GP._L
	assert L.shape[0] == n
	lead = torch.tril(L[:m, :m]).cpu().numpy()
	spec = [("squared_exponential", {"gamma": gamma, "kappa": 1.0}, "-")]
	xn = x.numpy()
	Lo = np.linalg.cholesky(O.gram_train(xn[:m], spec, s))
	e_lead = np.linalg.norm(lead - Lo) / np.linalg.norm(Lo)
	del lead, Lo
	rng = np.random.RandomState(11)
	idx = np.sort(rng.choice(np.arange(n - 8192, n), size=64, replace=False))
	idx_d = torch.from_numpy(idx).cuda()
	V = L[idx_d].clone()                                            # 64 x n; entries right of the diagonal are scratch
	cols = torch.arange(n, device="cuda")
	V[cols[None, :] > idx_d[:, None]] = 0.0
	R = torch.empty((64, n), dtype=torch.float64, device="cuda")
	step = 4096
	for j0 in range(0, n, step):
		j1 = j0 + step
		Lc = L[j0:j1, :j1].clone()
		Lc[:, j0:j1].tril_()                                        # (the strict upper triangle of the buffer is scratch)
		R[:, j0:j1] = V[:, :j1] @ Lc.T
		del Lc
	Kr = O.kernel(xn, xn[idx], spec)                                # (64, n): rows idx of k(x, x)
	Kr[np.arange(64), idx] += s * s
	Rn = R.cpu().numpy()
	mask = np.arange(n)[None, :] <= idx[:, None]
	e_rows = np.linalg.norm((Rn - Kr)[mask]) / np.linalg.norm(Kr[mask])
	print("headline factor vs oracle: leading %d block %.2e (Frobenius, relative); L L^T - K on 64 trailing rows %.2e" % (m, e_lead, e_rows))
	assert e_lead < 1e-10
	assert e_rows < 1e-11
	del GP, L, V, R
	_free()


def test_config4_shape_single_gpu(S):
	"""BASELINE config 4's shape -- N = 131 072, d = 32, SE, fp64, M = 4096 -- at full size on one MI355X (137 GB factored in place)
	through BOTH code paths that will carry it: the single-GPU class, and the 2-D block-cyclic schedule forced onto one rank
	(force_path=True: process group, sub-communicators, panel pipeline, distributed solve -- everything but a second GPU).
	The block-cyclic run is fitted on -3 y, so that one comparison checks agreement of the two paths AND linearity of the mean;
	plus the training-point identity mean(x_i) = y_i - s^2 alpha_i and 0 <= sigma <= sqrt(kappa)."""
	import torch.distributed as dist
	n, d, m = 131072, 32, 4096
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
	e_id = float(torch.norm(mu_tr - expect) / torch.norm(expect))
	assert e_id < 1e-8
	assert float(std_tr.max()) < 1.0
	mu_h, std_h = N(mu), N(std)
	del GP, alpha, mu_tr, std_tr
	_free()
	own_group = not dist.is_initialized()
	if own_group:
		import socket
		with socket.socket() as sk:
			sk.bind(("127.0.0.1", 0))
			port = sk.getsockname()[1]
		dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
	try:
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		DG = DistributedGaussianProcess(gamma=gamma, s=s, kappa=1.0, kernel_name="squared_exponential", d=d, force_path=True)
		DG.fit_gp(xd, -3.0 * yd)
		assert DG.NB == 2048                                        # the distribution block this size takes by default
		mu2, std2 = DG.mean_std(xtd)
		e_mu, e_sd = rel_err(N(mu2), -3.0 * mu_h), rel_err(N(std2), std_h)
		print("C4 shape on one GPU: identity %.2e; block-cyclic(one rank, -3y) vs single-GPU class: mu %.2e sigma %.2e" % (e_id, e_mu, e_sd))
		# (two different blockings -- 1024-column panels against 2048-wide distribution blocks -- of a matrix with cond(K) ~ 1e7:
		# measured 1.2e-9 on the mean, 2e-12 on sigma; the bound is the fp64 tolerance of the path, BASELINE.json north_star)
		assert e_mu < 1e-8 and e_sd < 1e-9
		del DG
	finally:
		if own_group:
			dist.destroy_process_group()
		_free()
