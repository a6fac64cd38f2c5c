"""
Drop-in for ``stpy.continuous_processes.gauss_procc.GaussianProcess`` on the squared-loss path
(reference: stpy/continuous_processes/gauss_procc.py:18-71 ctor, :100-117 add_data_point / fit,
:136-177 fit_gp, :198-209 execute, :310-418 mean_std / mean, :497-504 + :631-638 log_marginal,
:915 get_kernel; explicit-Cholesky form of the evidence in stpy/estimator.py:32-40).

What runs where
  Gram matrices      stpy_gram              (csrc/gram.hip)
  K = L L^T          stpy_potrf             (csrc/potrf.hip + csrc/gemm.hip, fp64/fp32 MFMA)
  z = L^-1 y, alpha  stpy_trsv              (csrc/solve.hip)
  X = K* L^-T        stpy_trsm_right_lt     (csrc/solve.hip + csrc/gemm.hip)
  mu, sigma          stpy_predict           (csrc/solve.hip)
  log det, y^T K^-1 y  stpy_logdet_quad     (csrc/solve.hip)
Python only sequences those calls and owns the torch tensors they operate on.

The reference solves with lstsq / LU / slogdet; an SPD solve through the Cholesky factor is the
same mathematics (and what Estimator.log_marginal does), agreement is checked to <= 1e-8 relative
against golden vectors captured from the reference (tests/golden).

Differences a caller can observe, all deliberate:
  * ``self.K`` / ``self.Sigma`` are materialised lazily (the factorisation is in place; at
    N = 65 536 a second N x N matrix is 34 GB); ``self.B`` (N x N, a by-product of the reference's
    lstsq, gauss_procc.py:378) is not kept.
  * prediction before ``fit`` with ``full=False`` returns the prior (0, sqrt(diag K**)) that
    gauss_procc.py:349-363 intends; the reference snapshot raises TypeError there (:346).
  * ``add_data`` / ``mean_var`` are aliases of ``add_data_point`` / ``mean_std`` (the names used in
    BASELINE.json); like every ``mean_var`` in stpy they return (mean, *std*).
  * robust losses, sampling helpers, gradients, UCB optimisation are outside the hot path.
"""
import numpy as np
import math

import torch

from .. import _lib
from ..estimator import Estimator, Euclidean
from ..kernels import KernelFunction


class _LogMarginalFn(torch.autograd.Function):
	"""Autograd node of GaussianProcess.log_marginal: forward = the HIP evidence, backward = its analytic gradient."""

	@staticmethod
	def forward(ctx, gp, kernel, X, weight, *tensors):
		val, state = gp._log_marginal_value(kernel, X, weight)
		ctx.gp, ctx.kernel, ctx.X, ctx.weight, ctx.state = gp, kernel, X, weight, state
		ctx.params = gp._grad_params(X)
		return val.clone()

	@staticmethod
	def backward(ctx, gout):
		grads = ctx.gp._log_marginal_grads(ctx.kernel, ctx.X, ctx.weight, ctx.state, ctx.params)
		scale = gout.reshape(-1)[0]
		return (None, None, None, None) + tuple(scale.to(g.device) * g for g in grads)


def _tile_pad(n):
	"""Order at which an n x n SPD matrix is held on the device: the next multiple of the 128 x 128 GEMM tile."""
	return -(-int(n) // 128) * 128


class GaussianProcess(Estimator):

	def __init__(self, gamma=1, s=0.001, kappa=1., kernel_name="squared_exponential", diameter=1.0,
				 groups=None, bounds=None, nu=1.5, kernel=None, d=1, power=2, lam=1., loss='squared', huber_delta=1.35,
				 hyper='classical', B=1., svr_eps=0.1):
		if loss != 'squared':
			raise NotImplementedError("loss='%s': only the squared loss is on the stpy_amd hot path "
									  "(huber/svr/unif need cvxpy/MOSEK, gauss_procc.py:211-308)" % loss)
		self.s = s
		self.d = d
		self.x = None
		self.y = None
		self.n = 0
		self.mu = 0.0
		self.lam = lam
		self.total_bound = B
		self.safe = False
		self.fitted = False
		self.diameter = diameter
		self.bounds = bounds
		self.admits_first_order = False
		self.back_prop = True
		self.loss = loss
		self.hyper = hyper
		self.max_size = 10000               # gauss_procc.py:55: prediction chunk
		self.clamp_variance = False         # the reference takes sqrt of the raw difference (:394-395)
		self.nb = 0                         # outer panel width for potrf/trsm (0 = library default)
		if kernel is not None:
			self.kernel_object = kernel
			self.kernel = kernel.kernel
			self.d = kernel.d
		else:
			self.kernel_object = KernelFunction(kernel_name=kernel_name, gamma=gamma, nu=nu, groups=groups, kappa=kappa,
												power=power, d=d)
			self.kernel = self.kernel_object.kernel
			self.gamma = gamma
			self.v = nu
			self.groups = groups
			self.kappa = kappa
			self.custom = kernel
			self.optkernel = kernel_name
		# device state
		self._xd = None
		self._yd = None
		self._L = None          # N x N, lower triangle = Cholesky factor of k(x,x) + Sigma^T Sigma
		self._winv = None       # inverse 128 x 128 diagonal blocks of L
		self._z = None          # L^-1 y
		self._Sigma = None
		self._alpha_cache = None

	# ------------------------------------------------------------------ small API mirrors
	def description(self):
		return self.kernel_object.description() + "\nlambda=" + str(self.s)

	def embed(self, x):
		return self.kernel_object.embed(x)

	def get_basis_size(self):
		return self.kernel_object.get_basis_size()

	def residuals(self, x, y):
		return self.mean(x) - y

	def add_data_point(self, x, y, Sigma=None):
		"""gauss_procc.py:100-111: concatenate and refit from scratch."""
		if self.x is not None:
			self.x = torch.cat((self.x, x), dim=0)
			self.y = torch.cat((self.y, y), dim=0)
			if Sigma is None and self._Sigma is not None:
				self._Sigma = torch.block_diag(self._Sigma, torch.eye(x.size()[0], dtype=torch.double) * self.s)
		else:
			self.x = x
			self.y = y
			self._Sigma = Sigma
		self.fit_gp(self.x, self.y, Sigma=self._Sigma)

	add_data = add_data_point

	def fit(self, x=None, y=None):
		"""gauss_procc.py:113-117."""
		if x is not None:
			self.fit_gp(x, y)
		else:
			self.fit_gp(self.x, self.y)

	def lcb(self, xtest):
		mu, s = self.mean_std(xtest)
		return mu - 2 * s

	def ucb(self, xtest):
		mu, s = self.mean_std(xtest)
		return mu + 2 * s

	# ------------------------------------------------------------------ factorisation
	@staticmethod
	def _add_noise_gram(K, Sigma):
		"""K += Sigma^T Sigma = K - (-Sigma^T)(Sigma^T)^T: one stpy_gemm_nt in subtract mode."""
		lib = _lib.load()
		St = _lib.to_device(Sigma, K.dtype).t().contiguous()
		nSt = -St
		n = K.shape[0]
		_lib.check(lib.stpy_gemm_nt(_lib.dtype_code(K.dtype), n, n, St.shape[1], _lib.ptr(nSt), _lib.ld(nSt), _lib.ptr(St), _lib.ld(St),
									_lib.ptr(K), _lib.ld(K), 1, 0, _lib.stream_ptr()), "stpy_gemm_nt")

	@staticmethod
	def _check_info(info):
		"""The one synchronisation of a fit: the factorisation's status word (first failing pivot, 1-based; 0 = fine)."""
		bad = int(info.item())
		if bad != 0:
			raise torch.linalg.LinAlgError("stpy_potrf: the leading minor of order %d of K + s^2 I is not positive definite" % bad)

	def _factor(self, xd, kwargs=None, Sigma=None, defer_check=False):
		"""K_theta = k(x,x) + s^2 I (or + Sigma^T Sigma) -> in-place Cholesky.  Returns (L, winv); with ``defer_check`` also the
		device status word, unread -- the caller enqueues what follows the factorisation first and then calls ``_check_info``,
		so the device does not idle through the host round trip."""
		lib = _lib.load()
		n0 = xd.shape[0]
		n = _tile_pad(n0)
		dt = _lib.dtype_code(xd.dtype)
		# The matrix is held at the next multiple of the 128-tile, bordered by an identity block:
		# chol([[K, 0], [0, I]]) = [[L, 0], [0, I]], so every product of the factorisation and of the
		# solves below runs on the tile-aligned kernels whatever N is (a ragged N = 32 700 cost 30 %).
		# Padded entries of y, z, alpha and the padded columns of K* are zero and drop out of every sum.
		Kp = torch.empty((n, n), dtype=xd.dtype, device=xd.device)
		K = Kp[:n0, :n0]
		if n > n0:
			Kp[n0:, :].zero_()
			Kp[:n0, n0:].zero_()              # (upper part of the last diagonal tile: the diagonal-block kernel loads whole tiles)
			Kp[n0:, n0:].diagonal().fill_(1.0)
		if Sigma is None:
			self.kernel_object._kernel_into(xd, xd, K, kwargs, diag_add=float(self.s) ** 2, lower_only=True)
		else:
			# general noise matrix (gauss_procc.py:163): K += Sigma^T Sigma through the NT product
			self.kernel_object._kernel_into(xd, xd, K, kwargs)
			self._add_noise_gram(K, Sigma)
		K = Kp
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(n)),), dtype=xd.dtype, device=xd.device)
		work = torch.empty((int(lib.stpy_potrf_workspace_bytes(dt, n, self.nb)),), dtype=torch.uint8, device=xd.device)
		info = torch.zeros((1,), dtype=torch.int32, device=xd.device)
		rc = lib.stpy_potrf(dt, n, _lib.ptr(K), _lib.ld(K), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, 0, _lib.ptr(info), _lib.stream_ptr())
		_lib.check(rc, "stpy_potrf")
		del work
		if defer_check:
			return K, winv, info
		self._check_info(info)
		return K, winv

	def _forward_y(self, L, winv, yd):
		"""z = L^-1 y (length = the padded order of L; the padding of y is zero)."""
		lib = _lib.load()
		scratch = torch.zeros((L.shape[0],), dtype=L.dtype, device=L.device)
		scratch[:yd.numel()] = yd.reshape(-1)
		z = torch.empty_like(scratch)
		_lib.check(lib.stpy_trsv(_lib.dtype_code(L.dtype), L.shape[0], _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(scratch), _lib.ptr(z), 0,
								 _lib.stream_ptr()), "stpy_trsv")
		return z

	def _backward_z(self, L, winv, z):
		"""alpha = L^-T z."""
		lib = _lib.load()
		scratch = z.clone()
		alpha = torch.empty_like(scratch)
		_lib.check(lib.stpy_trsv(_lib.dtype_code(L.dtype), L.shape[0], _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(scratch), _lib.ptr(alpha), 1,
								 _lib.stream_ptr()), "stpy_trsv")
		return alpha

	@property
	def _alpha(self):
		"""K^-1 y = L^-T z on the device (filled by fit_gp)."""
		if self._alpha_cache is None and self.fitted:
			self._alpha_cache = self._backward_z(self._L, self._winv, self._z)[:self.n]
		return self._alpha_cache

	@property
	def A(self):
		"""K^-1 y, (N, 1)  (gauss_procc.py:376)."""
		a = self._alpha
		return None if a is None else _lib.like_input(a.reshape(-1, 1), self.x)

	def fit_gp(self, x, y, Sigma=None, iterative=False, extrapoint=False):
		"""gauss_procc.py:136-177 (the ``iterative`` branch of the reference is a stub and is ignored)."""
		try:
			self.n, self.d = list(x.size())
		except Exception:
			self.n, self.d = x.shape
		self.x = x
		self.y = y
		self._Sigma = Sigma
		self._xd = _lib.to_device(x)
		self._yd = _lib.to_device(y, self._xd.dtype).reshape(-1, 1)
		# not fitted until the new factor exists: a refit that fails (not positive definite, out of memory) leaves an
		# object that takes the prior branch instead of one that reports fitted=True with no factor behind it
		self.fitted = False
		self._L = self._winv = self._z = self._alpha_cache = None       # release the previous factor before allocating the next
		L, winv, info = self._factor(self._xd, None, Sigma, defer_check=True)
		z = self._forward_y(L, winv, self._yd)
		# A = K^-1 y is part of the fitted state the reference leaves behind (gauss_procc.py:376): computed
		# eagerly even though mean_std itself only needs z
		alpha = self._backward_z(L, winv, z)[:self.n]
		# the two vector solves are already queued behind the factorisation when the host reads its status (on a matrix that is
		# not positive definite they ran on garbage and are dropped with the exception: the object stays unfitted)
		self._check_info(info)
		# ... and the sticky device word of the one-launch vector solves: a hand-off wait that gave up has poisoned z / alpha
		# with NaN (stpy_async_status; the stream is already drained by the read above, so this costs one 4-byte copy)
		_lib.check_async("fit_gp: stpy_trsv")
		self._L, self._winv, self._z, self._alpha_cache = L, winv, z, alpha
		self._factor_key = self._hyper_key(self.kernel_object)
		self.fitted = True
		return None

	def _hyper_key(self, kernel):
		"""What the resident factor was built from: the noise level and every stored kernel parameter, by value.
		log_marginal re-uses the factor only while this is unchanged (the reference rebuilds K from the CURRENT
		self.s / params_dict on every call, gauss_procc.py:631-638)."""
		def freeze(v):
			if torch.is_tensor(v):
				return ("t", tuple(v.detach().reshape(-1).tolist()))
			if isinstance(v, np.ndarray):
				return ("a", tuple(v.reshape(-1).tolist()))
			if isinstance(v, dict):
				return tuple(sorted((str(k), freeze(x)) for k, x in v.items()))
			if isinstance(v, (list, tuple)):
				return tuple(freeze(x) for x in v)
			return v
		return (freeze(self.s), id(kernel), freeze(kernel.params_dict), tuple(kernel.operations))

	# ------------------------------------------------------------------ lazily materialised reference attributes
	@property
	def K(self):
		"""k(x,x) + Sigma^T Sigma (gauss_procc.py:163), recomputed on demand."""
		if not self.fitted:
			return np.array([1.0])              # gauss_procc.py:38
		xd = self._xd
		K = torch.empty((self.n, self.n), dtype=xd.dtype, device=xd.device)
		if self._Sigma is None:
			self.kernel_object._kernel_into(xd, xd, K, None, diag_add=float(self.s) ** 2)
		else:
			self.kernel_object._kernel_into(xd, xd, K, None)
			self._add_noise_gram(K, self._Sigma)
		return _lib.like_input(K, self.x)

	@property
	def Sigma(self):
		if self._Sigma is not None:
			return self._Sigma
		return self.s * torch.eye(self.n, dtype=torch.float64)

	def get_kernel(self):
		return self.K

	def norm(self):
		"""gauss_procc.py:179-184: sqrt(alpha^T k(x,x) alpha).  With (k + s^2 I) alpha = y this is
		sqrt(alpha^T y - s^2 alpha^T alpha): no n x n matrix is formed."""
		if not self.fitted:
			return None
		lib = _lib.load()
		a = self._alpha.reshape(-1).contiguous()

		def dot(u, v):          # <u, v> in the fixed-order reduction kernel, read back as a host scalar
			o = torch.empty((2,), dtype=u.dtype, device=u.device)
			_lib.check(lib.stpy_trace_dot(_lib.dtype_code(u.dtype), u.shape[0], None, 0, _lib.ptr(u), _lib.ptr(v), _lib.ptr(o), _lib.stream_ptr()), "stpy_trace_dot")
			return float(o[1].item())
		if self._Sigma is None:
			noise = float(self.s) ** 2 * dot(a, a)
		else:                                   # general noise matrix: alpha^T Sigma^T Sigma alpha = |Sigma alpha|^2
			Sd = _lib.to_device(self._Sigma, a.dtype).contiguous()
			v = torch.empty((1, Sd.shape[0]), dtype=a.dtype, device=a.device)
			_lib.check(lib.stpy_gemm_nt(_lib.dtype_code(a.dtype), 1, Sd.shape[0], Sd.shape[1], _lib.ptr(a), a.shape[0], _lib.ptr(Sd), _lib.ld(Sd),
										_lib.ptr(v), _lib.ld(v), 0, 0, _lib.stream_ptr()), "stpy_gemm_nt")
			v = v.reshape(-1)
			noise = dot(v, v)
		val = dot(a, self._yd.reshape(-1).contiguous()) - noise
		return _lib.like_input(torch.full((1, 1), math.sqrt(val) if val >= 0 else float("nan"), dtype=a.dtype), self.x)

	def beta(self, delta=1e-3, norm=1):
		"""gauss_procc.py:186-196: s * norm + sqrt(2 log(1/delta + log(det K / s^n))), K = k(x,x) + s^2 I.
		log det K comes from the factor (2 sum log L_ii), so nothing overflows at sizes where det K would."""
		lib = _lib.load()
		L = self._L
		out2 = torch.empty((2,), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_logdet_quad(_lib.dtype_code(L.dtype), L.shape[0], _lib.ptr(L), _lib.ld(L), None, _lib.ptr(out2),
										_lib.stream_ptr()), "stpy_logdet_quad")
		log_ratio = 2.0 * float(out2[0].item()) - self.n * math.log(float(self.s))          # host scalars from here on
		arg = 1.0 / delta + log_ratio
		val = float(self.s) * norm + (math.sqrt(2.0 * math.log(arg)) if arg >= 1.0 else float("nan"))
		return _lib.like_input(torch.full((), val, dtype=L.dtype), self.x)

	# ------------------------------------------------------------------ prediction
	def execute(self, xtest):
		"""gauss_procc.py:198-209: (K* or None, K**)."""
		K_star = self.kernel(self._xd, _lib.to_device(xtest, self._xd.dtype)) if self.fitted else None
		if K_star is not None:
			K_star = _lib.like_input(K_star, xtest)
		K_star_star = self.kernel(xtest, xtest)
		return (K_star, K_star_star)

	def mean_std(self, xtest, full=False, reuse=False):
		"""gauss_procc.py:310-334: chunks of ``max_size`` test points against the resident factor."""
		m = xtest.size()[0]
		if m < self.max_size or full:
			return self.mean_std_sub(xtest, full=full, reuse=reuse)
		dtype = self._xd.dtype if self.fitted else (xtest.dtype if xtest.dtype in (torch.float32, torch.float64) else torch.float64)
		mu = torch.zeros(size=(m, 1), dtype=dtype, device=xtest.device)
		std = torch.zeros(size=(m, 1), dtype=dtype, device=xtest.device)
		for i0 in range(0, m, self.max_size):
			mu[i0:i0 + self.max_size], std[i0:i0 + self.max_size] = self.mean_std_sub(xtest[i0:i0 + self.max_size, :], reuse=True)
		return mu, std

	mean_var = mean_std

	def mean_std_sub(self, xtest, full=False, reuse=False):
		"""gauss_procc.py:336-401 (squared loss)."""
		lib = _lib.load()
		ko = self.kernel_object
		if not self.fitted:
			xt = _lib.to_device(xtest)
			if full:
				cov = torch.empty((xt.shape[0], xt.shape[0]), dtype=xt.dtype, device=xt.device)
				ko._kernel_into(xt, xt, cov)
				yvar = cov
			else:
				kd = torch.empty((xt.shape[0],), dtype=xt.dtype, device=xt.device)
				ko._diag_into(xt, kd)
				sd = torch.empty_like(kd)           # sqrt(diag K** - 0): the prediction epilogue with no data term
				_lib.check(lib.stpy_predict_finish(_lib.dtype_code(xt.dtype), xt.shape[0], None, _lib.ptr(torch.zeros_like(kd)), _lib.ptr(kd), 0.0,
												   _lib.ptr(sd), 0, _lib.stream_ptr()), "stpy_predict_finish")
				yvar = sd.reshape(-1, 1)
			zero = torch.zeros((xt.shape[0], 1), dtype=xt.dtype, device=xt.device)
			return (_lib.like_input(zero, xtest), _lib.like_input(yvar, xtest))

		xd = self._xd
		xt = _lib.to_device(xtest, xd.dtype)
		m, n = xt.shape[0], self.n
		dt = _lib.dtype_code(xd.dtype)
		st = _lib.stream_ptr
		n0, n = n, self._L.shape[0]                                     # n: order of the (tile-padded) factor
		mp = _tile_pad(m)                                               # rows of K* padded to the tile as well (zero rows)
		X = torch.empty((mp, n), dtype=xd.dtype, device=xd.device)
		ko._kernel_into(xd, xt, X[:m, :n0])                             # K* = k(x, xtest): (M, N)   :346
		if n > n0:
			X[:, n0:].zero_()
		if mp > m:
			X[m:, :].zero_()
		tw = torch.empty((int(lib.stpy_trsm_workspace_bytes(dt, mp, n, self.nb)),), dtype=torch.uint8, device=X.device)
		_lib.check(lib.stpy_trsm_right_lt(dt, mp, n, _lib.ptr(self._L), _lib.ld(self._L), _lib.ptr(self._winv), self._winv.numel(),
										  _lib.ptr(X), _lib.ld(X), self.nb, 0, _lib.ptr(tw), tw.numel() * tw.element_size(), st()), "stpy_trsm_right_lt")   # X = K* L^-T
		mu = torch.empty((m,), dtype=xd.dtype, device=xd.device)
		if not full:
			kd = torch.empty((m,), dtype=xd.dtype, device=xd.device)
			ko._diag_into(xt, kd)                                       # diag k(x*, x*)           :347
			sigma = torch.empty((m,), dtype=xd.dtype, device=xd.device)
			_lib.check(lib.stpy_predict(dt, m, n, _lib.ptr(X), _lib.ld(X), _lib.ptr(self._z), _lib.ptr(kd), _lib.ptr(mu),
										_lib.ptr(sigma), 1 if self.clamp_variance else 0, st()), "stpy_predict")
			return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(sigma.reshape(-1, 1), xtest))
		_lib.check(lib.stpy_predict(dt, m, n, _lib.ptr(X), _lib.ld(X), _lib.ptr(self._z), None, _lib.ptr(mu),
									None, 0, st()), "stpy_predict")
		cov = torch.empty((m, m), dtype=xd.dtype, device=xd.device)
		ko._kernel_into(xt, xt, cov)                                    # K**                      :343
		_lib.check(lib.stpy_gemm_nt(dt, m, m, n, _lib.ptr(X), _lib.ld(X), _lib.ptr(X), _lib.ld(X), _lib.ptr(cov),
									_lib.ld(cov), 1, 0, st()), "stpy_gemm_nt")                    # K** - X X^T  :396-399
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(cov, xtest))

	def mean(self, xtest):
		"""gauss_procc.py:403-418: K* alpha."""
		lib = _lib.load()
		xd = self._xd
		xt = _lib.to_device(xtest, xd.dtype)
		m, n = xt.shape[0], self.n
		Ks = torch.empty((m, n), dtype=xd.dtype, device=xd.device)
		self.kernel_object._kernel_into(xd, xt, Ks)
		mu = torch.empty((m,), dtype=xd.dtype, device=xd.device)
		_lib.check(lib.stpy_predict(_lib.dtype_code(xd.dtype), m, n, _lib.ptr(Ks), _lib.ld(Ks), _lib.ptr(self._alpha), None,
									_lib.ptr(mu), None, 0, _lib.stream_ptr()), "stpy_predict")
		return _lib.like_input(mu.reshape(-1, 1), xtest)

	# ------------------------------------------------------------------ sampling
	def sample(self, xtest, size=1, jitter=10e-8):
		"""
		gauss_procc.py:461-482: f = mean + chol(Cov + 1e-9 I) r  (posterior, full covariance) or
		mu + chol(K** + jitter I) r (prior).  The standard-normal draws are taken exactly as in the
		reference -- torch.normal on the CPU generator, shape (nn, size) -- so a seeded reference run
		and a seeded run here see the same random_vector; the M x M Cholesky and the product run in
		stpy_potrf / stpy_gemm_nt.
		"""
		lib = _lib.load()
		nn = list(xtest.size())[0]
		if self.fitted == True:
			(ymean, cov) = self.mean_std(xtest, full=True)
			eps = 10e-10
		else:
			(_, cov) = self.execute(xtest)
			ymean = self.mu
			eps = jitter
		cov = _lib.to_device(cov)
		C = torch.empty_like(cov)
		dt = _lib.dtype_code(C.dtype)
		_lib.check(lib.stpy_combine(dt, nn, nn, _lib.ptr(C), _lib.ld(C), _lib.ptr(cov), _lib.ld(cov), _lib.OUT_SET, eps, _lib.stream_ptr()), "stpy_combine")      # C = cov + eps I
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(nn)),), dtype=C.dtype, device=C.device)
		work = torch.empty((int(lib.stpy_potrf_workspace_bytes(dt, nn, self.nb)),), dtype=torch.uint8, device=C.device)
		info = torch.zeros((1,), dtype=torch.int32, device=C.device)
		_lib.check(lib.stpy_potrf(dt, nn, _lib.ptr(C), _lib.ld(C), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, 0, _lib.ptr(info), _lib.stream_ptr()), "stpy_potrf")
		bad = int(info.item())
		if bad != 0:
			raise torch.linalg.LinAlgError("sample: posterior covariance + jitter is not positive definite (leading minor %d)" % bad)
		_lib.check(lib.stpy_tril(dt, nn, _lib.ptr(C), _lib.ld(C), _lib.stream_ptr()), "stpy_tril")      # the strict upper triangle of an in-place factor is scratch
		random_vector = torch.normal(mean=torch.zeros(nn, size, dtype=torch.float64), std=1.)
		rt = random_vector.T.contiguous().to(device=C.device, dtype=C.dtype)          # (size, nn): the NT operand
		# f = ymean + L r: the accumulating product on a result that starts as the mean in every column
		f = torch.empty((nn, size), dtype=C.dtype, device=C.device)
		if torch.is_tensor(ymean):
			f.copy_(_lib.to_device(ymean, C.dtype).reshape(nn, 1).expand(nn, size))
		else:
			f.fill_(float(ymean))
		_lib.check(lib.stpy_gemm_nt(dt, nn, size, nn, _lib.ptr(C), _lib.ld(C), _lib.ptr(rt), _lib.ld(rt), _lib.ptr(f), _lib.ld(f), 2, 0,
									_lib.stream_ptr()), "stpy_gemm_nt")
		return _lib.like_input(f, xtest)

	def sample_and_max(self, xtest, size=1):
		"""gauss_procc.py:484-494."""
		f = self.sample(xtest, size=size)
		self.temp = f
		val, index = torch.max(f, dim=0)
		return (xtest[index, :], val)

	# ------------------------------------------------------------------ evidence
	def log_marginal(self, kernel, X, weight):
		"""
		gauss_procc.py:497-504 -> :631-638 (== estimator.py:32-40):
		    1/2 y^T (K_theta + s^2 I)^-1 y + 1/2 * weight * log det(K_theta + s^2 I),   shape (1, 1).
		Negative log evidence without the n/2 log(2 pi) constant.  ``X`` holds per-item parameter
		overrides in the kwargs protocol of kernels.py:138-157.  With X empty and ``kernel`` the
		fitted kernel object, the resident factor is reused.

		If a lengthscale tensor in ``X`` ('gamma' / 'ard_gamma'), the map of a full-covariance item ('cov') or the noise ``self.s`` requires grad --
		the way Estimator.optimize_params_general drives this method (estimator.py:156-190) -- the
		result carries an autograd node whose backward is the analytic evidence gradient
		1/2 tr((w K^-1 - alpha alpha^T) dK/dtheta) evaluated on the device (stpy_potri,
		stpy_lml_weight, stpy_gemm_nt).
		"""
		params = self._grad_params(X)
		if params:
			return _LogMarginalFn.apply(self, kernel, X, weight, *[t for (_, _, t) in params])
		return self._log_marginal_value(kernel, X, weight)[0]

	def _grad_params(self, X):
		"""(key, name, tensor) for every hyper-parameter tensor that asks for a gradient."""
		out = []
		for key in sorted(X.keys()) if X else []:
			for name in ("gamma", "ard_gamma", "cov"):
				v = X[key].get(name) if isinstance(X[key], dict) else None
				if torch.is_tensor(v) and v.requires_grad:
					out.append((key, name, v))
		if torch.is_tensor(self.s) and self.s.requires_grad:
			out.append(("likelihood", "sigma", self.s))
		return out

	def _log_marginal_value(self, kernel, X, weight):
		lib = _lib.load()
		if self._xd is None:
			if self.x is None:
				raise AttributeError("log_marginal needs data: call fit_gp or load_data first")
			self._xd = _lib.to_device(self.x)
			self._yd = _lib.to_device(self.y, self._xd.dtype).reshape(-1, 1)
			self.n = self._xd.shape[0]
		reuse = (self.fitted and (not X) and (kernel is self.kernel_object) and self._Sigma is None
				 and getattr(self, "_factor_key", None) == self._hyper_key(kernel))
		if reuse:
			L, winv, z = self._L, self._winv, self._z
		else:
			saved = self.kernel_object
			self.kernel_object = kernel
			try:
				L, winv = self._factor(self._xd, X, None)
			finally:
				self.kernel_object = saved
			z = self._forward_y(L, winv, self._yd)
		out2 = torch.empty((2,), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_logdet_quad(_lib.dtype_code(L.dtype), L.shape[0], _lib.ptr(L), _lib.ld(L), _lib.ptr(z), _lib.ptr(out2),
										_lib.stream_ptr()), "stpy_logdet_quad")
		w = float(weight) if not torch.is_tensor(weight) else float(weight.item())
		logdiag, quad = out2.tolist()                                                   # sum log L_ii, z^T z: host scalars
		val = torch.full((1, 1), 0.5 * quad + 0.5 * w * 2.0 * logdiag, dtype=L.dtype, device=L.device)
		return _lib.like_input(val, self.x), (L, winv, z)

	def _log_marginal_grads(self, kernel, X, weight, state, params):
		"""
		d/dtheta of the value above for every entry of ``params`` (same order), as tensors shaped like the parameters.

		G = w K^-1 - alpha alpha^T (stpy_potri).  The kernel is a chain of items combined by + and *
		(kernels.py:146-157), every item a sum of terms kappa phi(scaled distance).  For a lengthscale l_m of
		a term t of item i:   dK/dl_m = M_i o kappa F_t u_m^2 / l_m,   u_m the scaled coordinate difference, F_t the
		family's derivative factor, and M_i = dK/dK_i the elementwise product of everything item i is multiplied
		with (the value accumulated before it when its own operation is *, and every later item joined by *).
		So H = G o kappa F_t (stpy_lml_weight) o M_i (stpy_gram with the multiply combine), and
		sum_ij H_ij u_m^2 = 2 [ sum_i xs_im^2 h_i - xs_m^T H xs_m ] with h = H 1 -- one stpy_gemm_nt of H
		against [Xs | 1].
		"""
		lib = _lib.load()
		from ..kernels import _dev_const
		L, winv, z = state
		npad = L.shape[0]                               # tile-padded order of the factor (see _factor)
		dt = _lib.dtype_code(L.dtype)
		items = kernel._resolve(dict(X) if X else {})
		w = float(weight) if not torch.is_tensor(weight) else float(weight.item())
		st = _lib.stream_ptr
		xd = self._xd
		n = xd.shape[0]
		alpha = self._backward_z(L, winv, z)[:n]
		Kinv_p = torch.empty((npad, npad), dtype=L.dtype, device=L.device)
		work_p = torch.empty((npad, npad), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_potri(dt, npad, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(Kinv_p), _lib.ld(Kinv_p), _lib.ptr(work_p), work_p.numel() * work_p.element_size(), st()), "stpy_potri")
		# inverse of the bordered matrix = [[K^-1, 0], [0, I]]: everything below works on the leading n x n views
		Kinv, work = Kinv_p[:n, :n], work_p[:n, :n]
		_lib.check(lib.stpy_symmetrize_lower(dt, n, _lib.ptr(Kinv), _lib.ld(Kinv), st()), "stpy_symmetrize_lower")
		td = torch.empty((2,), dtype=L.dtype, device=L.device)                           # tr(K^-1), alpha^T alpha: fixed-order reduction
		_lib.check(lib.stpy_trace_dot(dt, n, _lib.ptr(Kinv), _lib.ld(Kinv), _lib.ptr(alpha), _lib.ptr(alpha), _lib.ptr(td), st()), "stpy_trace_dot")

		wanted = [(key, name, t) for (key, name, t) in params if key != "likelihood"]
		single = len(items) == 1 and len(items[0]['terms']) == 1
		acc = {(key, name): torch.zeros(t.numel(), dtype=L.dtype, device=L.device) for (key, name, t) in wanted}
		tmp = None
		for i, it in enumerate(items):
			mine = [(key, name) for (key, name, _) in wanted if key == str(i)]
			if not mine:
				continue
			for term in it['terms']:
				if term['pname'] is None or (str(i), term['pname']) not in acc:
					continue
				premap = term['premap']
				group = term['group']
				identity = (group == list(range(xd.shape[1])))
				cols = None if identity else _dev_const(group, None, xd.device, int32=True)
				inv_ls = _dev_const(term['inv_ls'], xd.dtype, xd.device)
				if premap is not None:
					# full-covariance item (kernels.py:464-549): the points enter as z = x[:, group] cov with unit lengthscales; the
					# parameter is the map itself.  d/dcov[a][m] = -1/2 sum_ij H_ij (z_i - z_j)_m (x_i - x_j)_a  (stpy_lml_grad_cov_reduce)
					kx = kernel._premap(xd, group, premap)                       # (n, p): what the weight / product kernels see as "the points"
					kcols, kd = None, kx.shape[1]
				else:
					kx, kcols, kd = xd, cols, len(group)
				# H <- (w K^-1 - alpha alpha^T) o kappa F_t: in place over K^-1 when this is the only term, otherwise written
				# to `work` with K^-1 only read (no N x N copy)
				H = Kinv if single else work
				ws = torch.empty((int(lib.stpy_gram_workspace_bytes(dt, n, n, kd)),), dtype=torch.uint8, device=xd.device)
				_lib.check(lib.stpy_lml_weight(term['kind'], dt, _lib.ptr(kx), n, _lib.ld(kx), kd, _lib.ptr(kcols), _lib.ptr(inv_ls),
											   term['kappa'], w, _lib.ptr(alpha), _lib.ptr(Kinv), _lib.ld(Kinv), _lib.ptr(H), _lib.ld(H),
											   _lib.ptr(ws), ws.numel() * ws.element_size(), st()), "stpy_lml_weight")
				# ... o M_i
				factors = []
				if it['op'] == "*" and i > 0:
					factors.append(items[:i])
				for j in range(i + 1, len(items)):
					if items[j]['op'] == "*":
						factors.append([items[j]])
				for fac in factors:
					if len(fac) == 1 and len(fac[0]['terms']) == 1:
						# a single-term factor multiplies straight into H (STPY_OUT_MUL combine of stpy_gram)
						kernel._run_items([dict(fac[0], op="*")], xd, xd, H, first_is_set=False)
						continue
					if tmp is None:
						tmp = torch.empty((n, n), dtype=L.dtype, device=L.device)
					kernel._run_items(fac, xd, xd, tmp)
					_lib.check(lib.stpy_combine(dt, n, n, _lib.ptr(H), _lib.ld(H), _lib.ptr(tmp), _lib.ld(tmp), _lib.OUT_MUL, 0.0, st()), "stpy_combine")
				# [Xs | 1]^T (dg + 1, n): scaled coordinates as the NT operand, then P = H [Xs | 1] and the per-coordinate sums
				dg = kd
				XT = torch.empty((dg + 1, n), dtype=xd.dtype, device=xd.device)
				_lib.check(lib.stpy_scaled_points_t(dt, _lib.ptr(kx), n, _lib.ld(kx), dg, _lib.ptr(kcols), _lib.ptr(inv_ls), _lib.ptr(XT), _lib.ld(XT), 1, st()), "stpy_scaled_points_t")
				P = torch.empty((n, dg + 1), dtype=xd.dtype, device=xd.device)
				_lib.check(lib.stpy_gemm_nt(dt, n, dg + 1, n, _lib.ptr(H), _lib.ld(H), _lib.ptr(XT), _lib.ld(XT), _lib.ptr(P), _lib.ld(P), 0, 0, st()), "stpy_gemm_nt")
				a_ = acc[(str(i), term['pname'])]
				if premap is not None:
					if a_.numel() != len(group) * dg:
						raise ValueError("evidence gradient: 'cov' has %d entries, the item maps %d columns to %d" % (a_.numel(), len(group), dg))
					_lib.check(lib.stpy_lml_grad_cov_reduce(dt, _lib.ptr(xd), n, _lib.ld(xd), len(group), _lib.ptr(cols), _lib.ptr(kx), _lib.ld(kx), dg,
															_lib.ptr(P), _lib.ld(P), _lib.ptr(a_), st()), "stpy_lml_grad_cov_reduce")
					continue
				pidx = _dev_const([int(v) for v in term['pidx']], None, xd.device, int32=True)
				_lib.check(lib.stpy_lml_grad_reduce(dt, _lib.ptr(xd), n, _lib.ld(xd), dg, _lib.ptr(cols), _lib.ptr(inv_ls), _lib.ptr(P), _lib.ld(P),
													_lib.ptr(pidx), _lib.ptr(a_), st()), "stpy_lml_grad_reduce")
		del work, work_p
		for key, name, t in wanted:
			if (key, name) not in acc or int(key) >= len(items) or not any(tm['pname'] == name for tm in items[int(key)]['terms']):
				raise NotImplementedError("evidence gradient: kernel item %s has no '%s' lengthscale on the device path" % (key, name))
		grads = []
		for (key, name, t) in params:
			if key == "likelihood":      # noise std: dK/ds = 2 s I  =>  s tr(w K^-1 - alpha alpha^T), host scalars
				sval = float(t.detach().reshape(-1)[0].item())
				trK, aa = td.tolist()
				g = torch.full(t.shape if t.dim() > 0 else (), sval * (w * trK - aa), dtype=L.dtype)
			else:
				g = acc[(key, name)].reshape(t.shape if t.dim() > 0 else ())
			grads.append(g.to(device=t.device, dtype=t.dtype))
		return grads

	# ------------------------------------------------------------------ hyper-parameter search (caller of the hot path)
	def optimize_params(self, type='bandwidth', restarts=10, regularizer=None,
						maxiter=1000, mingradnorm=1e-4, verbose=False, optimizer="pymanopt", scale=1., weight=1., save=False,
						save_name='model.np', init_func=None, bounds=None, parallel=False, cores=None):
		"""
		gauss_procc.py:640-702 for ``type`` in {"bandwidth", "bandwidth+noise"}: builds the ``params`` dictionary -- every
		kernel item's 'gamma' / 'ard_gamma' on a Euclidean factor, optionally the noise std under the key 'likelihood' -- and
		hands it to ``Estimator.optimize_params_general`` (stpy_amd/estimator.py), which minimises ``log_marginal`` over
		``restarts`` starting points, writes the best one back into ``kernel_object.params_dict`` / ``self.s`` and refits.
		Objective and gradient are the device evidence and its analytic gradient (every evaluation is a full Gram +
		Cholesky [+ inverse]).  The rotation / group / covariance searches of the reference (Stiefel and PSD manifolds,
		discrete group enumeration) are outside the hot path.
		"""
		if regularizer is not None:
			if regularizer[0] == "spectral_norm":          # gauss_procc.py:645-653
				regularizer_func = lambda S: regularizer[1] * torch.norm(1 / S.reshape(1, -1), p='nuc')
			elif regularizer[0] == 'lasso':
				regularizer_func = lambda S: regularizer[1] * torch.norm(1 / S, p=1)
			else:
				regularizer_func = None
		else:
			regularizer_func = None
		if type not in ("bandwidth", "bandwidth+noise"):
			if type in ("rots", "groups", "covariance"):
				raise NotImplementedError("optimize_params(type='%s') is outside the stpy_amd hot path" % type)
			raise AttributeError("This quick-optimization is not implemented.")          # gauss_procc.py:698
		params = {}
		for key, dict2 in self.kernel_object.params_dict.items():
			if 'gamma' in dict2.keys():
				params[key] = {'gamma': (init_func, Euclidean(1), bounds)}
			elif 'ard_gamma' in dict2.keys():
				params[key] = {'ard_gamma': (init_func, Euclidean(len(dict2['group'])), bounds)}
		if type == "bandwidth+noise":
			s0 = self.s
			params['likelihood'] = {'sigma': ((lambda k: s0), Euclidean(1), None)}          # init_func_noise = lambda x: self.s
		return self.optimize_params_general(params=params, restarts=restarts, optimizer=optimizer, regularizer_func=regularizer_func,
											maxiter=maxiter, mingradnorm=mingradnorm, verbose=verbose, scale=scale, weight=weight,
											save=save, save_name=save_name, parallel=parallel, cores=cores)

	def load_data(self, d):
		"""estimator.py:28-30."""
		self.x = d[0]
		self.y = d[1]
		self._xd = self._yd = None
		self.n = self.x.shape[0]
