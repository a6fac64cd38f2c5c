"""
CPU oracle for the stpy GP dense-linear-algebra hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of what the reference (Mojusko/stpy @ 2024-11-01) computes on
the path named by BASELINE.json:north_star.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing
under ``stpy_amd/`` imports it and the product path never falls back to it.

Parity status: PINNED.  Every function below is checked against golden vectors captured from the
real reference imported in the authoring container (``tests/golden/make_golden.py`` ->
``tests/golden/*.npz``; see ``tests/test_oracle_golden.py``).  The reference's own ``tests/`` hold
no assertions or fixtures, so those captured outputs are the only pin.

Each function cites the reference lines it follows (paths relative to the reference root).

Numerical note: the reference solves with ``torch.linalg.lstsq`` / LU / ``slogdet``
(gauss_procc.py:367-378, :631-638); mathematically that is the SPD solve restated here with a
Cholesky factor (the explicit-Cholesky form the reference itself uses in estimator.py:32-40).
"""
import math

import numpy as np
import scipy.linalg as sla
from scipy.spatial.distance import cdist

SQRT3 = math.sqrt(3.0)
SQRT5 = math.sqrt(5.0)


# --------------------------------------------------------------------------------------------
# kernel functions  (stpy/kernels.py)
# --------------------------------------------------------------------------------------------

def _cols(a, group):
	a = np.asarray(a, dtype=np.float64)
	if group is None:
		return a
	return a[:, list(group)]


def squared_exponential(a, b, gamma=1.0, kappa=1.0, group=None):
	"""kernels.py:368-398 -- norm expansion, no clamp of the squared distance; returns (|b|,|a|)."""
	a = _cols(a, group)
	b = _cols(b, group)
	normx = np.sum(a ** 2, axis=1).reshape(-1, 1)
	normy = np.sum(b ** 2, axis=1).reshape(-1, 1)
	product = b @ a.T
	sqdist = -2 * product + normx.T + normy
	arg = (-0.5 / (gamma * gamma)) * sqdist
	return kappa * np.exp(arg)


def ard(a, b, ard_gamma, kappa=1.0, group=None):
	"""kernels.py:552-583 -- columns scaled by 1/ard_gamma[group], then SE with gamma = 1."""
	ard_gamma = np.asarray(ard_gamma, dtype=np.float64).reshape(-1)
	a = _cols(a, group)
	b = _cols(b, group)
	g = ard_gamma if group is None else ard_gamma[list(group)]
	a = a / g
	b = b / g
	normx = np.sum(a ** 2, axis=1).reshape(-1, 1)
	normy = np.sum(b ** 2, axis=1).reshape(-1, 1)
	sqdist = -2 * (b @ a.T) + normx.T + normy
	return kappa * np.exp(-0.5 * sqdist)


def _matern_of_dist(dists, nu):
	"""kernels.py:844-851 (same expressions at :946-962)."""
	if nu == 0.5:
		return np.exp(-dists)
	if nu == 1.5:
		K = dists * SQRT3
		return (1. + K) * np.exp(-K)
	if nu == 2.5:
		K = dists * SQRT5
		return (1. + K + K ** 2 / 3.0) * np.exp(-K)
	raise NotImplementedError("general-nu Bessel Matern is out of scope (SURVEY.md section 2, row 1)")


def matern(a, b, gamma=1.0, nu=1.5, kappa=1.0, group=None):
	"""kernels.py:811-859 -- scipy cdist of a/gamma, b/gamma (direct differences)."""
	a = _cols(a, group)
	b = _cols(b, group)
	dists = cdist(a / gamma, b / gamma, metric='euclidean').T
	return kappa * _matern_of_dist(dists, nu)


def ard_matern(a, b, ard_gamma, nu=1.5, kappa=1.0, group=None):
	"""kernels.py:917-970 -- scale by 1/ard_gamma[group] first (mm with a diagonal), then the
	column subset, then torch.cdist."""
	ard_gamma = np.asarray(ard_gamma, dtype=np.float64).reshape(-1)
	a = np.asarray(a, dtype=np.float64)
	b = np.asarray(b, dtype=np.float64)
	g = ard_gamma if group is None else ard_gamma[list(group)]
	a = a / g  # reference multiplies by diag(1/g): requires a.shape[1] == len(group)
	b = b / g
	a = _cols(a, group)
	b = _cols(b, group)
	dists = cdist(a, b, metric='euclidean').T
	return kappa * _matern_of_dist(dists, nu)


def linear(a, b, kappa=1.0, offset=0.0, group=None):
	"""kernels.py:300-320."""
	a = _cols(a, group)
	b = _cols(b, group)
	return kappa * (b @ a.T) + offset


def ard_additive(a, b, ard_gamma, groups, kappa=1.0, group=None):
	"""kernels.py:697-725 -- mean over the column groups of ard kernels; the columns are first
	subset by ``group``, each entry of ``groups`` then indexes that subset (and ard_gamma)."""
	a = _cols(a, group)
	b = _cols(b, group)
	r = np.zeros((b.shape[0], a.shape[0]))
	for g in groups:
		r = r + ard(a, b, ard_gamma, kappa, g)
	return r / float(len(groups))


def squared_exponential_per_group(a, b, groups, gamma_per_group, kappa=1.0, inner_kappa=1.0):
	"""kernels.py:669-695 -- kappa * mean_g SE(group g, gamma_g); each SE term applies kappa itself
	as well (the object's, or the overriding one), so the constant enters twice."""
	r = np.zeros((np.asarray(b).shape[0], np.asarray(a).shape[0]))
	for g, gamma in zip(groups, gamma_per_group):
		r = r + squared_exponential(a, b, float(gamma), inner_kappa, g)
	return kappa * r / float(len(groups))


def ard_per_group(a, b, groups, ard_per_group_gamma, kappa=1.0):
	"""kernels.py:620-667 -- consecutive slices of the lengthscale vector belong to consecutive groups."""
	gam = np.asarray(ard_per_group_gamma, dtype=np.float64).reshape(-1)
	a = np.asarray(a, dtype=np.float64)
	b = np.asarray(b, dtype=np.float64)
	r = np.zeros((b.shape[0], a.shape[0]))
	idx = 0
	for g in groups:
		gl = gam[idx:idx + len(g)]
		idx += len(g)
		ax = a[:, list(g)] / gl
		bx = b[:, list(g)] / gl
		normx = np.sum(ax ** 2, axis=1).reshape(-1, 1)
		normy = np.sum(bx ** 2, axis=1).reshape(-1, 1)
		r = r + np.exp(-0.5 * (-2 * (bx @ ax.T) + normx.T + normy))
	return kappa * r / float(len(groups))


def full_covariance_se(a, b, cov, kappa=1.0, group=None):
	"""kernels.py:464-498 -- points mapped by the (square-root covariance) matrix, then SE with gamma = 1."""
	cov = np.asarray(cov, dtype=np.float64)
	a = _cols(a, group) @ cov
	b = _cols(b, group) @ cov
	normx = np.sum(a ** 2, axis=1).reshape(-1, 1)
	normy = np.sum(b ** 2, axis=1).reshape(-1, 1)
	return kappa * np.exp(-0.5 * (-2 * (b @ a.T) + normx.T + normy))


def full_covariance_matern(a, b, cov, nu=1.5, kappa=1.0, group=None):
	"""kernels.py:501-549 -- the same map, then Euclidean distances (torch.cdist) into the Matern forms."""
	cov = np.asarray(cov, dtype=np.float64)
	a = _cols(a, group) @ cov
	b = _cols(b, group) @ cov
	return kappa * _matern_of_dist(cdist(a, b, metric='euclidean').T, nu)


def polynomial(a, b, degree=2, kappa=1.0, group=None):
	"""kernels.py:744-761 -- kappa * (<b, a> + 1)^degree."""
	a = _cols(a, group)
	b = _cols(b, group)
	return kappa * (b @ a.T + 1) ** degree


_KERNELS = {
	"ard_additive": lambda a, b, p: ard_additive(a, b, p["ard_gamma"], p["groups"], p.get("kappa", 1.0), p.get("group")),
	"squared_exponential_per_group": lambda a, b, p: squared_exponential_per_group(a, b, p["groups"], p["gamma_per_group"], p.get("kappa", 1.0),
																					 p.get("inner_kappa", p.get("kappa", 1.0))),
	"ard_per_group": lambda a, b, p: ard_per_group(a, b, p["groups"], p["ard_per_group"], p.get("kappa", 1.0)),
	"full_covariance_se": lambda a, b, p: full_covariance_se(a, b, p["cov"], p.get("kappa", 1.0), p.get("group")),
	"full_covariance_matern": lambda a, b, p: full_covariance_matern(a, b, p["cov"], p.get("nu", 1.5), p.get("kappa", 1.0), p.get("group")),
	"polynomial": lambda a, b, p: polynomial(a, b, p.get("degree", 2), p.get("kappa", 1.0), p.get("group")),
	"squared_exponential": lambda a, b, p: squared_exponential(a, b, p.get("gamma", 1.0), p.get("kappa", 1.0), p.get("group")),
	"ard": lambda a, b, p: ard(a, b, p["ard_gamma"], p.get("kappa", 1.0), p.get("group")),
	"matern": lambda a, b, p: matern(a, b, p.get("gamma", 1.0), p.get("nu", 1.5), p.get("kappa", 1.0), p.get("group")),
	"ard_matern": lambda a, b, p: ard_matern(a, b, p["ard_gamma"], p.get("nu", 1.5), p.get("kappa", 1.0), p.get("group")),
	"linear": lambda a, b, p: linear(a, b, p.get("kappa", 1.0), p.get("offset", 0.0), p.get("group")),
}


def kernel(a, b, spec, overrides=None):
	"""
	Composite kernel evaluation, kernels.py:136-159.

	spec: list of (kernel_name, params_dict, op) with op in {"-", "+", "*"}; the first entry's op
	is "-" (kernels.py:74).  overrides: {'0': {...}, '1': {...}} replaces the stored params of an
	item *wholesale* except that 'group' is re-injected (kernels.py:105-110, :138-144) -- the
	per-kernel functions then fall back to the constructor values for missing keys.
	"""
	out = None
	for i, (name, params, op) in enumerate(spec):
		p = dict(params)
		if overrides:
			p.update(overrides.get(str(i), {}))
		k = _KERNELS[name](a, b, p)
		if op == "+":
			out = out + k
		elif op == "*":
			out = out * k
		else:
			out = k
	return out


# --------------------------------------------------------------------------------------------
# GP fit / predict / log-marginal  (stpy/continuous_processes/gauss_procc.py, stpy/estimator.py)
# --------------------------------------------------------------------------------------------

def gram_train(x, spec, s, overrides=None):
	"""gauss_procc.py:151-163 -- K = k(x,x) + Sigma^T Sigma with Sigma = s*I  ==  k(x,x) + s^2 I."""
	K = kernel(x, x, spec, overrides)
	K[np.diag_indices_from(K)] += s * s
	return K


def fit(x, y, spec, s):
	"""gauss_procc.py:136-177 + :375-376 -- returns (L, alpha) with K = L L^T, alpha = K^-1 y."""
	K = gram_train(x, spec, s)
	L = sla.cholesky(K, lower=True, check_finite=False)
	alpha = sla.cho_solve((L, True), np.asarray(y, dtype=np.float64).reshape(-1, 1), check_finite=False)
	return L, alpha


def mean_std(x, L, alpha, xtest, spec):
	"""
	gauss_procc.py:336-401 (squared loss, full=False): mu = K* alpha; sigma = sqrt(diag k(x*,x*) -
	diag(K* K^-1 K*^T)).  No clamp of the variance (gauss_procc.py:394-395).
	"""
	Ks = kernel(x, xtest, spec)                      # (M, N)   :346
	kdiag = kernel_diag(xtest, spec)                 # :347
	mu = Ks @ alpha                                  # :381
	V = sla.solve_triangular(L, Ks.T, lower=True, check_finite=False)   # (N, M)
	var = kdiag - np.sum(V * V, axis=0)              # :391-394
	with np.errstate(invalid="ignore"):
		std = np.sqrt(var)
	return mu.reshape(-1, 1), std.reshape(-1, 1)


def fit_predict_reference_shaped(x, y, xtest, spec, s):
	"""
	The reference's OWN operation sequence for fit_gp + mean_std (what the CPU baseline of bench.py times), as opposed to
	the Cholesky restatement above which is mathematically equal and far cheaper:
	  gauss_procc.py:151-163  Sigma = s * eye(n) materialised, K = k(x,x) + Sigma^T @ Sigma  (a dense n^3 product of a diagonal);
	  gauss_procc.py:176      fit_gp ends with mean_std(x): K* = k(x, x) (n x n), the per-point Python loop for diag k(x*,x*)
	                          (:347), A = lstsq(K, y) (:376), B = lstsq(K, K*^T)^T with n right-hand sides (:378),
	                          mean = K* A, variance = diag - einsum(B, K*^T)  (:381-395);
	  then the user's mean_std(xtest) repeats the last step with M right-hand sides (and solves for A again, reuse=False).
	torch.linalg.lstsq on CPU uses LAPACK gelsy (complete orthogonal factorisation); scipy's driver of the same name is used here.
	Returns (mu, std) for xtest.
	"""
	x = np.asarray(x, dtype=np.float64)
	y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
	n = x.shape[0]
	Sigma = s * np.eye(n)
	K = kernel(x, x, spec) + Sigma.T @ Sigma

	def mean_std_sub(xt):
		Ks = kernel(x, xt, spec)
		kd = kernel_diag(xt, spec)
		A = sla.lstsq(K, y, lapack_driver="gelsy", check_finite=False)[0]
		B = sla.lstsq(K, Ks.T, lapack_driver="gelsy", check_finite=False)[0].T
		mean = Ks @ A
		var = kd.reshape(-1, 1) - np.einsum('ij,ji->i', B, Ks.T).reshape(-1, 1)
		with np.errstate(invalid="ignore"):
			return mean, np.sqrt(var)
	mean_std_sub(x)                                   # gauss_procc.py:176
	return mean_std_sub(np.asarray(xtest, dtype=np.float64))


def mean_cov(x, L, alpha, xtest, spec):
	"""gauss_procc.py:396-399 (full=True): (mu, K** - K* K^-1 K*^T)."""
	Ks = kernel(x, xtest, spec)
	Kss = kernel(xtest, xtest, spec)
	mu = Ks @ alpha
	V = sla.solve_triangular(L, Ks.T, lower=True, check_finite=False)
	return mu.reshape(-1, 1), Kss - V.T @ V


def kernel_diag(xtest, spec):
	"""gauss_procc.py:347 -- the reference loops kernel(x_i, x_i) over test points."""
	xtest = np.asarray(xtest, dtype=np.float64)
	out = np.empty(xtest.shape[0])
	# row-at-a-time exactly like the reference (cheap for oracle-sized M); vectorised per chunk
	for i0 in range(0, xtest.shape[0], 1024):
		blk = xtest[i0:i0 + 1024]
		out[i0:i0 + 1024] = np.array([kernel(blk[i:i + 1], blk[i:i + 1], spec)[0, 0] for i in range(blk.shape[0])])
	return out


def prior_std(xtest, spec):
	"""gauss_procc.py:349-363 -- unfitted branch: (0, sqrt(diag K**))."""
	return np.zeros((xtest.shape[0], 1)), np.sqrt(kernel_diag(xtest, spec)).reshape(-1, 1)


def log_marginal(x, y, spec, s, overrides=None, weight=1.0):
	"""
	gauss_procc.py:631-638 == estimator.py:32-40:  1/2 y^T K^-1 y + 1/2 * weight * log det K,
	K = k_theta(x,x) + s^2 I.  *Negative* log evidence, no n/2 log(2 pi).  Shape (1,1).
	"""
	K = gram_train(x, spec, s, overrides)
	L = sla.cholesky(K, lower=True, check_finite=False)
	y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
	z = sla.solve_triangular(L, y, lower=True, check_finite=False)
	logdet = 2.0 * np.sum(np.log(np.diag(L)))
	return np.array([[0.5 * float((z.T @ z)[0, 0]) + 0.5 * weight * logdet]])


def _item_dk(x, name, p):
	"""(k_i, {param: [dk_i / dparam_m for every component m]}) of one kernel item on x, x -- the items whose kernel functions
	are torch-differentiable in the reference (kernels.py:368-398 SE, :552-583 ARD, :697-725 additive-group ARD)."""
	x = np.asarray(x, dtype=np.float64)
	if name == "squared_exponential":
		g = float(np.asarray(p.get("gamma", 1.0)).reshape(-1)[0])
		xs = _cols(x, p.get("group"))
		sq = ((xs[:, None, :] - xs[None, :, :]) ** 2).sum(-1)
		k = squared_exponential(x, x, g, p.get("kappa", 1.0), p.get("group"))
		return k, {"gamma": [k * sq / g ** 3]}
	if name in ("ard", "ard_additive"):
		ag = np.asarray(p["ard_gamma"], dtype=np.float64).reshape(-1)
		xs = _cols(x, p.get("group"))
		groups = p["groups"] if name == "ard_additive" else [list(range(xs.shape[1]))]
		k = np.zeros((x.shape[0], x.shape[0]))
		dk = [np.zeros_like(k) for _ in range(ag.shape[0])]
		for grp in groups:
			kg = ard(xs, xs, ag, p.get("kappa", 1.0), grp)
			k += kg / float(len(groups))
			for m in grp:
				dk[m] += kg * (xs[:, None, m] - xs[None, :, m]) ** 2 / ag[m] ** 3 / float(len(groups))
		return k, {"ard_gamma": dk}
	if name in ("full_covariance_se", "full_covariance_matern"):
		# kernels.py:464-549: z = x[:, group] cov, then SE (gamma = 1) or Matern on |z_i - z_j| -- torch ops end to end in the reference (mm,
		# exp / cdist), so autograd yields d/dcov.  k = kappa phi(r):  dk/dcov[a][m] = kappa phi'(r) (z_i - z_j)_m (x_i - x_j)_a / r
		cov = np.asarray(p["cov"], dtype=np.float64)
		xs = _cols(x, p.get("group"))
		z = xs @ cov
		dz = z[:, None, :] - z[None, :, :]
		dx = xs[:, None, :] - xs[None, :, :]
		r = np.sqrt((dz ** 2).sum(-1))
		kappa = p.get("kappa", 1.0)
		if name == "full_covariance_se":
			k = full_covariance_se(x, x, cov, kappa, p.get("group"))
			fac = -k                                        # phi'(r) / r = -exp(-r^2 / 2)
		else:
			nu = p.get("nu", 1.5)
			k = full_covariance_matern(x, x, cov, nu, kappa, p.get("group"))
			with np.errstate(divide="ignore", invalid="ignore"):
				if nu == 0.5:
					fac = np.where(r > 0, -kappa * np.exp(-r) / r, 0.0)
				elif nu == 1.5:
					fac = -3.0 * kappa * np.exp(-SQRT3 * r)
				elif nu == 2.5:
					fac = -(5.0 / 3.0) * kappa * (1.0 + SQRT5 * r) * np.exp(-SQRT5 * r)
				else:
					raise NotImplementedError("full_covariance_matern gradient: nu in {0.5, 1.5, 2.5}")
		return k, {"cov": [fac * dz[:, :, m] * dx[:, :, a] for a in range(cov.shape[0]) for m in range(cov.shape[1])]}
	raise NotImplementedError("no reference gradient for kernel %r (Matern goes through NumPy in the reference, kernels.py:840-859)" % name)


def log_marginal_grad(x, y, spec, s, overrides=None, weight=1.0):
	"""
	Value and gradient of gauss_procc.py:631-638 with respect to the lengthscale parameters of every item and the noise level:
	d/dtheta [1/2 y^T K^-1 y + w/2 log det K] = 1/2 tr((w K^-1 - alpha alpha^T) dK/dtheta)  -- what autograd through the
	reference's log_marginal yields (estimator.py:156-190 hands exactly that to its optimisers); pinned by golden G14.
	Items are combined as kernels.py:146-157: dK/dk_i = (value accumulated before item i, if it is joined by "*", else 1)
	times every later item joined by "*".  Returns (value (1,1), {item index: {param: gradient vector}}, d/ds).
	"""
	x = np.asarray(x, dtype=np.float64)
	y = np.asarray(y, dtype=np.float64).reshape(-1, 1)
	items = []
	for i, (name, params, op) in enumerate(spec):
		p = dict(params)
		if overrides:
			p.update(overrides.get(str(i), {}))
		items.append((op,) + _item_dk(x, name, p))
	# prefix values (before item i) and suffix products (later items joined by "*")
	acc, prefix = None, []
	for op, k, _ in items:
		prefix.append(acc)
		acc = k if acc is None or op == "-" else (acc + k if op == "+" else acc * k)
	K = acc.copy()
	K[np.diag_indices_from(K)] += s * s
	L = sla.cholesky(K, lower=True, check_finite=False)
	alpha = sla.cho_solve((L, True), y, check_finite=False)
	Kinv = sla.cho_solve((L, True), np.eye(K.shape[0]), check_finite=False)
	G = 0.5 * (weight * Kinv - alpha @ alpha.T)
	z = sla.solve_triangular(L, y, lower=True, check_finite=False)
	value = np.array([[0.5 * float((z.T @ z)[0, 0]) + 0.5 * weight * 2.0 * np.sum(np.log(np.diag(L)))]])
	grads = {}
	for i, (op, k, dks) in enumerate(items):
		M = np.ones_like(k) if (op != "*" or prefix[i] is None) else prefix[i]
		for j in range(i + 1, len(items)):
			# a later "+" item leaves d(out)/d(k_i) unchanged; a later "*" item multiplies it
			if items[j][0] == "*":
				M = M * items[j][1]
		grads[i] = {name: np.array([float(np.sum(G * M * d)) for d in dl]) for name, dl in dks.items()}
	return value, grads, float(2.0 * s * np.trace(G))


def norm(x, alpha, spec):
	"""gauss_procc.py:179-184: sqrt(alpha^T k(x,x) alpha)."""
	a = np.asarray(alpha, dtype=np.float64).reshape(-1, 1)
	return np.sqrt(a.T @ kernel(x, x, spec) @ a)


def beta(x, spec, s, delta=1e-3, norm=1):
	"""gauss_procc.py:186-196: s * norm + sqrt(2 log(1/delta + log(det K / s^n)))."""
	K = gram_train(x, spec, s)
	n = K.shape[0]
	return s * norm + np.sqrt(2 * np.log(1. / delta + np.log(np.linalg.det(K) / s ** n)))


# --------------------------------------------------------------------------------------------
# Random Fourier features  (stpy/embeddings/embedding.py)
# --------------------------------------------------------------------------------------------

def rff_embed(x, W, m, kappa=1.0, b=None):
	"""
	embedding.py:225-241.  Unbiased: columns j < m/2 are sqrt(2/m) cos(w_j.x) and columns
	j >= m/2 are sqrt(2/m) sin(w_j.x) -- the sin half uses frequency rows m/2..m-1, *not* the cos
	rows again; result (n, m).  Biased: sqrt(2/m) cos(w_j.x + b_j); the reference transposes that
	branch twice (embedding.py:232 and :241), so the biased result has shape (m, n) -- kept.
	Result times sqrt(kappa).
	"""
	x = np.asarray(x, dtype=np.float64)
	W = np.asarray(W, dtype=np.float64)
	d = x.shape[1]
	q = W[:, 0:d] @ x.T                                  # (m, n)
	c = np.sqrt(2. / float(m))
	if b is not None:
		z = (c * np.cos(q + np.asarray(b, dtype=np.float64).reshape(m, 1))).T
	else:
		h = int(m / 2)
		z = np.concatenate([c * np.cos(q[0:h, :]), c * np.sin(q[h:m, :])])
	return z.T * np.sqrt(kappa)


def rff_sample_W(gamma, m, d, rng_state=None):
	"""embedding.py:159,191,216 -- SE spectral density: W = N(0,1)^{m x d} / gamma (global numpy RNG)."""
	if rng_state is not None:
		np.random.seed(rng_state)
	return np.random.normal(size=(m, d)) * (1. / gamma)


def quadrature_embed(x, W, weights, kappa=1.0, cosine=False):
	"""
	embedding.py:450-466 (QuadratureEmbedding.embed and every derived class): with q = W[:, :d] x^T (m/2, n),
	rows 0..m/2-1 are sqrt(w_j) cos(q_j) and rows m/2..m-1 are sqrt(w_j) sin(q_j) of the SAME nodes; ``cosine``: the
	cosine rows only.  Returns (n, m) times sqrt(kappa).
	"""
	x = np.asarray(x, dtype=np.float64)
	W = np.asarray(W, dtype=np.float64)
	sw = np.sqrt(np.asarray(weights, dtype=np.float64)).reshape(-1, 1)
	q = W[:, 0:x.shape[1]] @ x.T
	z = sw * np.cos(q) if cosine else np.concatenate([sw * np.cos(q), sw * np.sin(q)])
	return z.T * np.sqrt(kappa)


# --------------------------------------------------------------------------------------------
# Primal ridge regression on a finite feature map  (stpy/continuous_processes/kernelized_features.py)
# --------------------------------------------------------------------------------------------

def kernelized_features_fit(Q, y, s, lam):
	"""kernelized_features.py:236-240 + :252-253 (primal): V = Q^T Q + s^2 lam I, invV = pinv(V),
	theta = invV Q^T y, Z = s^2 invV."""
	Q = np.asarray(Q, dtype=np.float64)
	m = Q.shape[1]
	V = Q.T @ Q + s ** 2 * lam * np.eye(m)
	invV = np.linalg.pinv(V)
	theta = invV @ Q.T @ np.asarray(y, dtype=np.float64).reshape(-1, 1)
	return V, invV, theta


def kernelized_features_mean_std(Qtest, invV, theta, s):
	"""kernelized_features.py:269-288: mean = Phi* theta, std = sqrt(s^2 diag(Phi* invV Phi*^T))."""
	Qtest = np.asarray(Qtest, dtype=np.float64)
	mean = Qtest @ theta
	diag = s ** 2 * np.einsum('ij,jk,ik->i', Qtest, invV, Qtest).reshape(-1, 1)
	return mean, np.sqrt(diag)


def kernelized_features_dual_fit(Q, y, s, lam):
	"""kernelized_features.py:229-235 + :252-254 (dual, n < m with primal=False): K = Q Q^T + s^2 lam I, invK = pinv(K),
	invK_V = (I - Q^T invK Q) / lam, theta = Q^T invK y."""
	Q = np.asarray(Q, dtype=np.float64)
	n, m = Q.shape
	K = Q @ Q.T + s ** 2 * lam * np.eye(n)
	invK = np.linalg.pinv(K)
	invK_V = (1. / lam) * (-Q.T @ invK @ Q + np.eye(m))
	theta = Q.T @ invK @ np.asarray(y, dtype=np.float64).reshape(-1, 1)
	return K, invK_V, theta


def kernelized_features_dual_mean_std(Qtest, invK_V, theta):
	"""kernelized_features.py:279, :285-287 (dual): mean = Phi* theta, std = sqrt(diag(Phi* invK_V Phi*^T))."""
	Qtest = np.asarray(Qtest, dtype=np.float64)
	return Qtest @ theta, np.sqrt(np.einsum('ij,jk,ik->i', Qtest, invK_V, Qtest).reshape(-1, 1))


def kernelized_features_dual_get_invV(Q, s, lam):
	"""kernelized_features.py:167-172 (get_invV in the dual form): V = linear_kernel(Q^T, Q^T) + s^2 lam I with the linear kernel
	object of :51 (default d = 1, so group = [0]: only the first COLUMN of Q^T, i.e. the features of the first data point,
	enters) -- V = q0 q0^T + s^2 lam I."""
	Q = np.asarray(Q, dtype=np.float64)
	q0 = Q[0:1, :].T                                      # (m, 1)
	V = q0 @ q0.T + s ** 2 * lam * np.eye(Q.shape[1])
	return np.linalg.solve(V, np.eye(Q.shape[1]))


def kernelized_features_sample_theta(theta_mean, invV, s, random_vector):
	"""kernelized_features.py:327-330: L = chol(invV) s, theta = theta_mean + L r  (r: (basis, size) standard normals)."""
	L = np.linalg.cholesky(np.asarray(invV, dtype=np.float64)) * s
	return np.asarray(theta_mean, dtype=np.float64).reshape(-1, 1) + L @ np.asarray(random_vector, dtype=np.float64)


def kernelized_features_prior_theta(lam, random_vector, prior_mean=0.0):
	"""kernelized_features.py:305-307, :332-334: chol(lam I) r + prior_mean."""
	return math.sqrt(lam) * np.asarray(random_vector, dtype=np.float64) + prior_mean


def kernelized_features_sample_matheron(Q, Qtest, y, K_star, K_train, s, lam, random_vector, prior_mean=0.0):
	"""kernelized_features.py:300-317: theta ~ prior; f = Phi* theta + K* pinv(K + s^2 lam I)(y - Phi theta).
	K_star = kernel(x, xtest) (M, N), K_train = kernel(x, x) of the exact kernel object."""
	theta = kernelized_features_prior_theta(lam, random_vector, prior_mean)
	f_prior_xtest = np.asarray(Qtest, dtype=np.float64) @ theta
	f_prior_x = np.asarray(Q, dtype=np.float64) @ theta
	N = K_train.shape[0]
	K = np.asarray(K_train, dtype=np.float64) + s ** 2 * lam * np.eye(N)
	return f_prior_xtest + np.asarray(K_star, dtype=np.float64) @ np.linalg.pinv(K) @ (np.asarray(y, dtype=np.float64).reshape(-1, 1) - f_prior_x)


def kernelized_features_first_feature_kernel(Qx, Qy, diag_add=0.0):
	"""kernelized_features.py:93-97 / :553-557 with the linear kernel object of :51 (group = [0]): (|y|, |x|) outer product of the
	FIRST feature column, + diag_add I."""
	K = np.asarray(Qy, dtype=np.float64)[:, :1] @ np.asarray(Qx, dtype=np.float64)[:, :1].T
	if diag_add:
		K = K + diag_add * np.eye(K.shape[0])
	return K


def kernelized_features_logdet_ratio(K, s, lam, m):
	"""kernelized_features.py:99-101: logdet(K) - logdet(s^2 lam I_m); K is the ones(1, 1) placeholder in the primal form."""
	return np.linalg.slogdet(np.asarray(K, dtype=np.float64))[1] - m * math.log(s ** 2 * lam)


def kernelized_features_effective_dim(Qtest, lam):
	"""kernelized_features.py:103-106: trace of the solution X of (Phi^T Phi + lam I) X = Phi^T Phi."""
	Q = np.asarray(Qtest, dtype=np.float64)
	A = Q.T @ Q
	return float(np.trace(np.linalg.solve(A + lam * np.eye(A.shape[0]), A)))


def kernelized_features_beta_theory(Q, s, lam, bound, delta):
	"""kernelized_features.py:64-73: bound lam + logdet(Q^T Q / s^2 + lam I) - logdet(lam I) + 2 log(1 / delta)."""
	Q = np.asarray(Q, dtype=np.float64)
	m = Q.shape[1]
	V = Q.T @ Q / s ** 2 + lam * np.eye(m)
	return bound * lam + np.linalg.slogdet(V)[1] - m * math.log(lam) + 2 * math.log(1. / delta)


# --------------------------------------------------------------------------------------------
# synthetic workloads shared by tests and bench  
# --------------------------------------------------------------------------------------------

def simple_1d_function(X):
	"""test_functions/benchmarks.py:478-482."""
	z = (X + 0.5) * 1.2
	return -(1.4 - 3 * z) * np.sin(18 * z)


def interval(n, d, L_infinity_ball=1.0):
	"""helpers/helper.py:27-59,125-136 -- cartesian grid of linspace(-L, L, n) per dimension."""
	arrays = [np.linspace(-L_infinity_ball, L_infinity_ball, n) for _ in range(d)]
	mesh = np.meshgrid(*arrays, indexing="ij")
	return np.stack([m.reshape(-1) for m in mesh], axis=1)
