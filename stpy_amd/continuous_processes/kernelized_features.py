"""
Drop-in for ``stpy.continuous_processes.kernelized_features.KernelizedFeatures``
(reference: kernelized_features.py:12-54 ctor, :56-106 beta / embed / kernel /
logdet_ratio / effective_dim, :108-138 add_data_point / fit_gp, :164-246 get_invV / precompute, :248-298 theta_mean /
mean_std / ucb / lcb, :300-336 sample_matheron / sample_theta, :537-562 sample / sample_and_max / get_kernel / residuals).

As in the reference the class derives from ``GaussianProcess`` and does not run the base constructor: ``log_marginal(kernel, X,
weight)`` (the evidence of ``kernel`` on the stored data, with its analytic gradient), ``optimize_params`` (needs a
``kernel_object`` attribute, which the reference never sets either), ``load_data`` ... are the inherited device paths.

Ridge regression on a finite feature map Phi (n x m), e.g. random Fourier features:

  primal (default, or n >= m):  V = Phi^T Phi + s^2 lam I,  theta = V^-1 Phi^T y,  std = s sqrt(diag(Phi* V^-1 Phi*^T))
  dual (primal=False, n < m):   K = Phi Phi^T + s^2 lam I,  theta = Phi^T K^-1 y,  std^2 = (|phi*|^2 - phi*^T Phi^T K^-1 Phi phi*) / lam
                                -- the GP path on the linear kernel of the features (kernelized_features.py:229-235, :252-254, :285)

Device mapping -- every contraction is the NT MFMA GEMM because the embedding is produced TRANSPOSED (Phi^T, m x n:
``embed_t``), and in the primal form Phi is STREAMED: row slabs of x are embedded one at a time (``slab_bytes`` of features live,
2 GB by default) and accumulated, so the n x m feature matrix is never materialised (at BASELINE config 5's shape it is 34 GB).
That requires an embedding that acts ROW BY ROW (each output row depends on its own input row only); every embedding of
``stpy_amd.embeddings`` does, and a fit of more than one slab checks it on the first row.
    V          += Phi_slab^T Phi_slab   stpy_syrk(mode "+="), lower tiles; + s^2 lam on the diagonal: stpy_combine
    Phi^T y    += Phi_slab^T y_slab     stpy_predict (row sums against y_slab) + stpy_combine(ADD)
    V = L L^T                           stpy_potrf  (the reference takes pinverse(V); V is SPD for s, lam > 0)
    theta                               stpy_trsv forward + backward
    X = Phi* L^-T, mean = X u, std      stpy_trsm_right_lt, stpy_predict, stpy_predict_finish
    samplers                            stpy_potri -> stpy_potrf -> stpy_tril -> stpy_gemm_nt (chol(V^-1) s r, the reference's own factor
                                        of the covariance, so a seeded run draws the same theta); Matheron: stpy_gram + stpy_potrf +
                                        stpy_trsm_right_lt + stpy_gemm_nt
Standard-normal draws are taken exactly as the reference takes them (torch.normal on the CPU generator, shape (basis, size)).

``add_data_point`` queues points as the reference does (:108-113) and the next prediction folds them in: k new rows cost their
embedding, one k-deep ``+=`` product and one m x m refactorisation (the reference's rank-one Woodbury / Schur updates of the
explicit inverse, :181-221, call ``add_points`` with the wrong arity and raise).  The cvxpy / MOSEK constrained fits
(:338-435) and the scipy multistart optimisers (:462-535) are outside the hot path.

Reference quirks kept (pinned by goldens G12 / G15): ``kernel`` and ``get_kernel`` use a linear kernel object built with the
default d = 1, so only the FIRST feature enters (:51, :93-97, :553-557); ``logdet_ratio`` in the primal form reads the
placeholder ``K = ones(1, 1)`` (:25, :99-101); ``get_invV`` in the dual form builds V from that same first-column kernel of
Q^T (:167-172), which is what ``sample_theta`` then draws from.
"""
import math

import numpy as np
import torch

from .. import _lib
from ..kernels import KernelFunction
from .gauss_procc import GaussianProcess


class KernelizedFeatures(GaussianProcess):

	def __init__(self, embedding, m, s=0.001, lam=1., d=1, diameter=1.0, theta_norm=1.0, verbose=True, groups=None,
				 bounds=None, scale=1.0, kappa=1.0, poly=2, primal=True, beta_fun=None, bound=1):
		self.s = s
		self.lam = lam
		self.primal = primal
		self.x = None
		self.y = None
		self.mu = 0.0
		self.m = torch.from_numpy(np.array(m))
		self.fitted = False
		self.data = False
		self.d = d
		self.n = 0
		self.bounds = bounds
		self.groups = groups
		self.diameter = diameter
		self.theta_norm = theta_norm
		self.verbose = verbose
		self.admits_first_order = True
		self.embedding = embedding
		self.embedding_map = embedding
		self.kappa = kappa
		self.scale = scale
		self.poly = poly
		self.to_add = []
		self.prior_mean = 0
		self.dual = False
		self.beta_fun = beta_fun
		self.bound = bound
		# what the inherited GaussianProcess methods read (the reference leaves these unset and its inherited log_marginal
		# stops at ``self.loss``)
		self.loss = 'squared'
		self.back_prop = True
		self.max_size = 10000
		self.clamp_variance = False
		self.kernel_object = None
		self.nb = 0
		self.slab_bytes = 2 << 30          # features held at a time while V and Phi^T y are accumulated (fit_gp)
		self._xd = self._yd = None
		self._L = self._winv = self._z = self._Sigma = self._alpha_cache = None     # (the exact-GP factor of the base class: unused)
		self._Lf = self._winvf = self._u = self._theta = None     # factor of V (primal, m x m) or K (dual, n x n); u = L^-1 rhs
		self._Vacc = self._rhs = self._part = None                # accumulated Phi^T Phi (lower tiles) and Phi^T y
		self._PhiT = None                                         # dual form: Phi^T (m, n), kept (n < m)

	# ------------------------------------------------------------------ small API mirrors
	def description(self):
		return "Custom Features object"

	def embed(self, x):
		return self.embedding.embed(x)

	def set_embedding(self, embed):
		self.embedding_map = embed

	def get_basis_size(self):
		return int(torch.sum(self.m))

	def set_basis_size(self, m):
		self.m = m

	@property
	def K(self):
		"""kernelized_features.py:25 (primal: the placeholder ones(1, 1)) / :232 (dual: Phi Phi^T + s^2 lam I)."""
		if not (self.dual and self.fitted):
			return torch.ones(size=(1, 1)).double()
		return _lib.like_input(self._dual_K(), self.x)

	def _embed_t(self, xd):
		"""Phi^T on the device, (m, n)."""
		if hasattr(self.embedding, "embed_t"):
			return self.embedding.embed_t(xd)
		return _lib.to_device(self.embedding.embed(xd)).T.contiguous()       # generic embeddings: one transpose copy

	def _first_feature_kernel(self, x, y, diag_add=0.0):
		"""(|y|, |x|) linear kernel of the FIRST feature only (see the module header), + diag_add on the diagonal; device tensor."""
		lib = _lib.load()
		ex = _lib.to_device(self.embed(_lib.to_device(x)))[:, :1].contiguous()
		ey = _lib.to_device(self.embed(_lib.to_device(y)), ex.dtype)[:, :1].contiguous()
		out = torch.empty((ey.shape[0], ex.shape[0]), dtype=ex.dtype, device=ex.device)
		dt = _lib.dtype_code(ex.dtype)
		_lib.check(lib.stpy_gemm_nt(dt, ey.shape[0], ex.shape[0], ex.shape[1], _lib.ptr(ey), _lib.ld(ey), _lib.ptr(ex), _lib.ld(ex),
									_lib.ptr(out), _lib.ld(out), 0, 0, _lib.stream_ptr()), "stpy_gemm_nt")
		if diag_add != 0.0:
			_lib.check(lib.stpy_combine(dt, out.shape[0], out.shape[1], _lib.ptr(out), _lib.ld(out), _lib.ptr(out), _lib.ld(out), _lib.OUT_SET, diag_add,
										_lib.stream_ptr()), "stpy_combine")
		return out

	def kernel(self, x, y):
		"""kernelized_features.py:93-97."""
		return _lib.like_input(self._first_feature_kernel(x, y), x)

	def get_kernel(self):
		"""kernelized_features.py:553-557."""
		return _lib.like_input(self._first_feature_kernel(self.x, self.x, float(self.s) ** 2 * float(self.lam)), self.x)

	def _logdet_factor(self):
		"""2 sum log L_ii of the resident factor, host scalar."""
		lib = _lib.load()
		L = self._Lf
		out2 = torch.empty((2,), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_logdet_quad(_lib.dtype_code(L.dtype), L.shape[0], _lib.ptr(L), _lib.ld(L), None, _lib.ptr(out2), _lib.stream_ptr()), "stpy_logdet_quad")
		return 2.0 * float(out2[0].item())

	def logdet_ratio(self):
		"""kernelized_features.py:99-101: logdet(self.K) - logdet(s^2 lam I_m)."""
		self.precompute()
		m = self.get_basis_size()
		ld = self._logdet_factor() if (self.dual and self.fitted) else 0.0        # primal: K is the ones(1, 1) placeholder
		return torch.tensor(ld - m * math.log(float(self.s) ** 2 * float(self.lam)), dtype=torch.float64)

	def effective_dim(self, xtest):
		"""kernelized_features.py:103-106: tr((Phi^T Phi + lam I)^-1 Phi^T Phi) = m - lam tr((Phi^T Phi + lam I)^-1)
		(the reference line calls torch.solve, which current torch no longer has; this is what it computes)."""
		lib = _lib.load()
		xt = _lib.to_device(xtest)
		PhiT = _lib.to_device(self._embed_t(xt), xt.dtype)
		if PhiT.stride(1) != 1:
			PhiT = PhiT.contiguous()
		m, k = PhiT.shape
		dt = _lib.dtype_code(PhiT.dtype)
		st = _lib.stream_ptr
		A = torch.empty((m, m), dtype=PhiT.dtype, device=PhiT.device)
		_lib.check(lib.stpy_gemm_nt(dt, m, m, k, _lib.ptr(PhiT), _lib.ld(PhiT), _lib.ptr(PhiT), _lib.ld(PhiT), _lib.ptr(A), _lib.ld(A), 0, 1, st()), "stpy_gemm_nt")
		_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(A), _lib.ld(A), _lib.ptr(A), _lib.ld(A), _lib.OUT_SET, float(self.lam), st()), "stpy_combine")
		L, winv = self._chol(A, "effective_dim: Phi^T Phi + lam I")
		inv = self._inverse_from_factor(L, winv)
		td = torch.empty((2,), dtype=inv.dtype, device=inv.device)
		_lib.check(lib.stpy_trace_dot(dt, m, _lib.ptr(inv), _lib.ld(inv), None, None, _lib.ptr(td), st()), "stpy_trace_dot")
		return torch.tensor(m - float(self.lam) * float(td[0].item()), dtype=torch.float64)

	def beta(self, delta=0.1, norm=None):
		"""kernelized_features.py:56-76."""
		if norm is None:
			norm = self.theta_norm
		if self.beta_fun is None:
			return 2.0
		if self.beta_fun == "theory":
			# bound lam + logdet(Q^T Q / s^2 + lam I) - logdet(lam I) + 2 log(1/delta), from the resident factor:
			# logdet(Q^T Q + c I_m) = logdet(V) in the primal form, logdet(K) + (m - n) log c in the dual one (c = s^2 lam)
			self.precompute()
			m = self.get_basis_size()
			c = float(self.s) ** 2 * float(self.lam)
			ldV = self._logdet_factor() + ((m - self.n) * math.log(c) if self.dual else 0.0)
			val = float(self.bound) * float(self.lam) + ldV - m * math.log(float(self.s) ** 2) - m * math.log(float(self.lam)) + 2 * np.log(1 / delta)
			return torch.tensor(val, dtype=torch.float64)
		return self.beta_fun(self.K, delta=delta, norm=norm)

	# ------------------------------------------------------------------ fit
	def add_data_point(self, x, y):
		"""kernelized_features.py:108-113: the first point fits, later ones are queued and folded in by the next ``precompute``."""
		if self.n == 0:
			self.fit_gp(x, y)
		else:
			self.to_add.append([x, y])
			self.fitted = False

	add_data = add_data_point

	def add_points(self, d):
		"""kernelized_features.py:140-147."""
		x, y = d
		if self.x is not None:
			self.x = torch.cat((self.x, x), dim=0)
			self.y = torch.cat((self.y, y), dim=0)
		else:
			self.x = x
			self.y = y

	def fit(self, x=None, y=None):
		self.fit_gp(self.x if x is None else x, self.y if y is None else y)

	def fit_gp(self, x, y):
		"""kernelized_features.py:118-138."""
		self.x, self.y = x, y
		self.n = list(x.size())[0]
		self.d = list(x.size())[1]
		self.dual = (self.n < self.get_basis_size()) and not self.primal
		self.data = True
		self.fitted = False
		self.to_add = []
		self._Vacc = self._rhs = self._part = self._PhiT = None
		self._Lf = self._winvf = self._u = self._theta = None
		self.precompute()
		return None

	def precompute(self):
		"""kernelized_features.py:176-246."""
		if self.fitted or not self.data:
			return
		if len(self.to_add) > 0 and self._Vacc is not None and not self.dual:
			# primal: the accumulated normal equations are extended by the queued rows
			newx = torch.cat([p[0] for p in self.to_add], dim=0)
			newy = torch.cat([p[1] for p in self.to_add], dim=0)
			self.to_add = []
			self.add_points((newx, newy))
			self.n = list(self.x.size())[0]
			self._xd = self._yd = None
			dtype = self._Vacc.dtype
			self._accumulate(_lib.to_device(newx, dtype), _lib.to_device(newy, dtype).reshape(-1), first=False)
			self._solve_normal_equations()
			return
		if len(self.to_add) > 0:
			for p in self.to_add:
				self.add_points((p[0], p[1]))
			self.to_add = []
			self.n = list(self.x.size())[0]
			self.dual = (self.n < self.get_basis_size()) and not self.primal          # (check_conversion, :149-162)
		xd = _lib.to_device(self.x)
		yd = _lib.to_device(self.y, xd.dtype).reshape(-1)
		self._xd, self._yd = xd, yd.reshape(-1, 1)
		if self.dual:
			self._fit_dual(xd, yd)
		else:
			self._Vacc = self._rhs = self._part = None
			self._accumulate(xd, yd, first=True)
			self._solve_normal_equations()

	def _accumulate(self, xd, yd, first):
		"""V_acc (+)= Phi^T Phi (lower tiles) and rhs (+)= Phi^T y over row slabs of xd; ``first``: the buffers are (re)created."""
		lib = _lib.load()
		n = xd.shape[0]
		esz = xd.element_size()
		st = _lib.stream_ptr
		dt = _lib.dtype_code(xd.dtype)
		m = None if first else self._Vacc.shape[0]
		rows = n if first else max(128, (int(self.slab_bytes) // (m * esz)) // 128 * 128)
		r0 = 0
		while r0 < n:
			take = min(4096 if m is None else rows, n - r0)          # (first slab of a fit: a probe that tells the feature count)
			PhiT = _lib.to_device(self._embed_t(xd[r0:r0 + take]), xd.dtype)            # (m, take)
			if PhiT.stride(1) != 1:
				PhiT = PhiT.contiguous()
			if m is None:
				m = PhiT.shape[0]
				rows = max(128, (int(self.slab_bytes) // (m * esz)) // 128 * 128)
				self._Vacc = torch.empty((m, m), dtype=xd.dtype, device=xd.device)
				self._rhs = torch.empty((1, m), dtype=xd.dtype, device=xd.device)
				self._part = torch.empty((1, m), dtype=xd.dtype, device=xd.device)
				if take < n:
					# slab-wise embedding is only the embedding of the whole set if the map acts row by row: row 0 alone
					# must reproduce column 0 of the slab
					alone = _lib.to_device(self._embed_t(xd[0:1]), xd.dtype).reshape(-1)
					if not torch.allclose(alone, PhiT[:, 0], rtol=1e-6 if xd.dtype == torch.float32 else 1e-12, atol=1e-6 if xd.dtype == torch.float32 else 1e-12):
						raise ValueError("KernelizedFeatures: the embedding does not act row by row (embed(x[0:1]) differs from the first row of "
										 "embed(x[0:%d])); the streaming fit needs that -- raise slab_bytes so that one slab holds all rows" % take)
			V = self._Vacc
			# V (+)= Phi_slab^T Phi_slab, lower tiles only: mode 0 for the first slab of a fit, 2 (accumulate) afterwards
			# (fp32 slabs of 2048 features and more: the slab is split once into bf16 planes in a workspace and every output tile reads those)
			wb = int(lib.stpy_syrk_workspace_bytes(dt, m, take))
			work = torch.empty((wb,), dtype=torch.uint8, device=PhiT.device) if wb > 0 else None
			_lib.check(lib.stpy_syrk(dt, m, take, _lib.ptr(PhiT), _lib.ld(PhiT), _lib.ptr(V), _lib.ld(V), 0 if first else 2,
									 _lib.ptr(work) if work is not None else None, wb, st()), "stpy_syrk")
			del work
			# Phi_slab^T y_slab: row sums of Phi^T against y
			ys = yd[r0:r0 + take]
			tgt = self._rhs if first else self._part
			_lib.check(lib.stpy_predict(dt, m, take, _lib.ptr(PhiT), _lib.ld(PhiT), _lib.ptr(ys), None, _lib.ptr(tgt), None, 0, st()), "stpy_predict")
			if not first:
				_lib.check(lib.stpy_combine(dt, 1, m, _lib.ptr(self._rhs), m, _lib.ptr(self._part), m, _lib.OUT_ADD, 0.0, st()), "stpy_combine")
			first = False
			r0 += take
			del PhiT

	def _chol(self, A, what, keep=None):
		"""In-place stpy_potrf of A; returns (A, winv).  Raises LinAlgError (and leaves the object unfitted) on a failing pivot."""
		lib = _lib.load()
		n = A.shape[0]
		dt = _lib.dtype_code(A.dtype)
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(n)),), dtype=A.dtype, device=A.device)
		work = torch.empty((int(lib.stpy_potrf_workspace_bytes(dt, n, self.nb)),), dtype=torch.uint8, device=A.device)
		info = torch.zeros((1,), dtype=torch.int32, device=A.device)
		_lib.check(lib.stpy_potrf(dt, n, _lib.ptr(A), _lib.ld(A), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, 0, _lib.ptr(info),
								  _lib.stream_ptr()), "stpy_potrf")
		bad = int(info.item())
		if bad != 0:
			raise torch.linalg.LinAlgError("KernelizedFeatures: %s is not positive definite (leading minor %d)" % (what, bad))
		return A, winv

	def _solve_pair(self, L, winv, rhs):
		"""u = L^-1 rhs, v = L^-T u (rhs is copied: stpy_trsv uses its right-hand side as scratch)."""
		lib = _lib.load()
		n = L.shape[0]
		dt = _lib.dtype_code(L.dtype)
		st = _lib.stream_ptr
		r = rhs.reshape(-1).clone()
		u = torch.empty_like(r)
		_lib.check(lib.stpy_trsv(dt, n, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(r), _lib.ptr(u), 0, st()), "stpy_trsv")
		scratch = u.clone()
		v = torch.empty_like(u)
		_lib.check(lib.stpy_trsv(dt, n, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(scratch), _lib.ptr(v), 1, st()), "stpy_trsv")
		_lib.check_async("KernelizedFeatures: stpy_trsv")           # a hand-off wait that gave up has poisoned u / v with NaN
		return u, v

	def _solve_normal_equations(self):
		"""V = V_acc + s^2 lam I -> Cholesky, u = L^-1 (Phi^T y), theta = L^-T u."""
		lib = _lib.load()
		Vacc = self._Vacc
		m = Vacc.shape[0]
		dt = _lib.dtype_code(Vacc.dtype)
		V = torch.empty_like(Vacc)
		# V = V_acc, then + s^2 lam on the diagonal (one pass of the elementwise kernel)
		_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(V), _lib.ld(V), _lib.ptr(Vacc), _lib.ld(Vacc), _lib.OUT_SET, float(self.s) ** 2 * float(self.lam), _lib.stream_ptr()), "stpy_combine")
		self.fitted = False
		L, winv = self._chol(V, "Phi^T Phi + s^2 lam I")
		u, theta = self._solve_pair(L, winv, self._rhs)
		self._Lf, self._winvf, self._u, self._theta = L, winv, u, theta
		self.fitted = True

	def _dual_K(self):
		"""Phi Phi^T + s^2 lam I (n x n, full) from the kept Phi^T."""
		lib = _lib.load()
		Phi = self._PhiT.t().contiguous()                                 # (n, m); n < m
		n, m = Phi.shape
		dt = _lib.dtype_code(Phi.dtype)
		K = torch.empty((n, n), dtype=Phi.dtype, device=Phi.device)
		_lib.check(lib.stpy_gemm_nt(dt, n, n, m, _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(K), _lib.ld(K), 0, 0, _lib.stream_ptr()), "stpy_gemm_nt")
		_lib.check(lib.stpy_combine(dt, n, n, _lib.ptr(K), _lib.ld(K), _lib.ptr(K), _lib.ld(K), _lib.OUT_SET, float(self.s) ** 2 * float(self.lam), _lib.stream_ptr()), "stpy_combine")
		return K

	def _fit_dual(self, xd, yd):
		"""kernelized_features.py:229-235, :252-254: K = Q Q^T + s^2 lam I -> Cholesky, z = L^-1 y, theta = Q^T K^-1 y."""
		lib = _lib.load()
		PhiT = _lib.to_device(self._embed_t(xd), xd.dtype)                # (m, n)
		if PhiT.stride(1) != 1:
			PhiT = PhiT.contiguous()
		self._PhiT = PhiT
		self.fitted = False
		L, winv = self._chol(self._dual_K(), "Phi Phi^T + s^2 lam I")
		z, alpha = self._solve_pair(L, winv, yd)
		m, n = PhiT.shape
		theta = torch.empty((m,), dtype=PhiT.dtype, device=PhiT.device)
		_lib.check(lib.stpy_predict(_lib.dtype_code(PhiT.dtype), m, n, _lib.ptr(PhiT), _lib.ld(PhiT), _lib.ptr(alpha), None, _lib.ptr(theta), None, 0,
									_lib.stream_ptr()), "stpy_predict")                             # theta = Phi^T alpha
		self._Lf, self._winvf, self._u, self._theta = L, winv, z, theta
		self.fitted = True

	# ------------------------------------------------------------------ V, V^-1
	def _inverse_from_factor(self, L, winv):
		"""(L L^T)^-1, full symmetric, device."""
		lib = _lib.load()
		m = L.shape[0]
		dt = _lib.dtype_code(L.dtype)
		out = torch.empty((m, m), dtype=L.dtype, device=L.device)
		work = torch.empty((m, m), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_potri(dt, m, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(out), _lib.ld(out), _lib.ptr(work), work.numel() * work.element_size(),
								  _lib.stream_ptr()), "stpy_potri")
		_lib.check(lib.stpy_symmetrize_lower(dt, m, _lib.ptr(out), _lib.ld(out), _lib.stream_ptr()), "stpy_symmetrize_lower")
		return out

	def _V_device(self):
		"""The matrix ``self.V`` of the reference: primal Phi^T Phi + s^2 lam I (:239); dual the first-column form of get_invV (:167-170)."""
		lib = _lib.load()
		c = float(self.s) ** 2 * float(self.lam)
		if self.dual:
			q0 = self._PhiT[:, :1].contiguous()                          # Q^T[:, group = [0]]: the features of the first data point
			m = q0.shape[0]
			dt = _lib.dtype_code(q0.dtype)
			V = torch.empty((m, m), dtype=q0.dtype, device=q0.device)
			_lib.check(lib.stpy_gemm_nt(dt, m, m, 1, _lib.ptr(q0), 1, _lib.ptr(q0), 1, _lib.ptr(V), _lib.ld(V), 0, 0, _lib.stream_ptr()), "stpy_gemm_nt")
			_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(V), _lib.ld(V), _lib.ptr(V), _lib.ld(V), _lib.OUT_SET, c, _lib.stream_ptr()), "stpy_combine")
			return V
		Vacc = self._Vacc
		m = Vacc.shape[0]
		dt = _lib.dtype_code(Vacc.dtype)
		V = torch.empty_like(Vacc)
		_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(V), _lib.ld(V), _lib.ptr(Vacc), _lib.ld(Vacc), _lib.OUT_SET, c, _lib.stream_ptr()), "stpy_combine")
		_lib.check(lib.stpy_symmetrize_lower(dt, m, _lib.ptr(V), _lib.ld(V), _lib.stream_ptr()), "stpy_symmetrize_lower")
		return V

	@property
	def V(self):
		"""Phi^T Phi + s^2 lam I (kernelized_features.py:239), full symmetric; built on demand from the accumulated lower tiles."""
		self.precompute()
		return _lib.like_input(self._V_device(), self.x)

	def _invV_device(self):
		if self.dual:
			V = self._V_device()
			L, winv = self._chol(V, "V (dual get_invV)")
			return self._inverse_from_factor(L, winv)
		return self._inverse_from_factor(self._Lf, self._winvf)

	@property
	def invV(self):
		"""V^-1 (the reference keeps pinverse(V), kernelized_features.py:240); from the factor on demand."""
		return self.get_invV()

	def get_invV(self):
		"""kernelized_features.py:164-174."""
		self.precompute()
		return _lib.like_input(self._invV_device(), self.x)

	def theta_mean(self, var=False, prior=False):
		"""kernelized_features.py:248-264."""
		lib = _lib.load()
		self.precompute()
		if self.fitted and not prior:
			theta = _lib.like_input(self._theta.reshape(-1, 1), self.x)
		else:
			theta = 0 * torch.ones(size=(self.get_basis_size(), 1)).double()
		if var is False:
			return theta
		if not (self.fitted and not prior):
			raise UnboundLocalError("theta_mean(var=True) needs a fitted model (the reference leaves Z undefined here, :258-264)")
		if not self.dual:
			Z = float(self.s) ** 2 * self.invV                       # (host-side scale of a returned matrix, as :257)
			return (theta, Z)
		# dual: Z = invK_V = (I - Q^T K^-1 Q) / lam = (I - W W^T) / lam with W = Q^T L^-T  (m x n)
		L = self._Lf
		m, n = self._PhiT.shape
		dt = _lib.dtype_code(L.dtype)
		st = _lib.stream_ptr
		W = self._PhiT.clone()
		_lib.check(lib.stpy_trsm_right_lt(dt, m, n, _lib.ptr(L), _lib.ld(L), _lib.ptr(self._winvf), self._winvf.numel(), _lib.ptr(W), _lib.ld(W), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
		Z = torch.eye(m, dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_gemm_nt(dt, m, m, n, _lib.ptr(W), _lib.ld(W), _lib.ptr(W), _lib.ld(W), _lib.ptr(Z), _lib.ld(Z), 1, 0, st()), "stpy_gemm_nt")
		if float(self.lam) != 1.0:
			inv_lam = torch.full((m, m), 1.0 / float(self.lam), dtype=L.dtype, device=L.device)
			_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(Z), _lib.ld(Z), _lib.ptr(inv_lam), _lib.ld(inv_lam), _lib.OUT_MUL, 0.0, st()), "stpy_combine")
		return (theta, _lib.like_input(Z, self.x))

	# ------------------------------------------------------------------ predict
	def mean(self, xtest):
		return self.mean_std(xtest)[0]

	def mean_std(self, xtest):
		"""kernelized_features.py:269-288."""
		lib = _lib.load()
		self.precompute()
		L = self._Lf
		xt = _lib.to_device(xtest, L.dtype)
		Phi = _lib.to_device(self.embedding.embed(xt), L.dtype)          # (M, m): rows = right-hand sides
		if Phi.stride(1) != 1:
			Phi = Phi.contiguous()
		M, m = Phi.shape
		dt = _lib.dtype_code(L.dtype)
		st = _lib.stream_ptr
		mu, ss = torch.empty((M,), dtype=L.dtype, device=L.device), torch.empty((M,), dtype=L.dtype, device=L.device)
		std = torch.empty_like(ss)
		if not self.dual:
			X = Phi.clone()
			_lib.check(lib.stpy_trsm_right_lt(dt, M, m, _lib.ptr(L), _lib.ld(L), _lib.ptr(self._winvf), self._winvf.numel(), _lib.ptr(X), _lib.ld(X), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
			_lib.check(lib.stpy_predict(dt, M, m, _lib.ptr(X), _lib.ld(X), _lib.ptr(self._u), None, _lib.ptr(mu), _lib.ptr(ss), 2, st()), "stpy_predict")
			# std = s sqrt(ss) = sqrt(0 - (-s^2) ss): the prediction epilogue with a zero prior term (no torch arithmetic on the vectors)
			_lib.check(lib.stpy_predict_finish(dt, M, None, _lib.ptr(ss), _lib.ptr(torch.zeros_like(ss)), -float(self.s) ** 2, _lib.ptr(std), 0, st()), "stpy_predict_finish")
			return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(std.reshape(-1, 1), xtest))
		# dual: K* = Phi* Phi^T (M x n), X = K* L^-T, mean = X z, var = (|phi*|^2 - rowsum(X o X)) / lam
		PhiTr = self._PhiT.t().contiguous()                               # (n, m)
		n = PhiTr.shape[0]
		X = torch.empty((M, n), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_gemm_nt(dt, M, n, m, _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(PhiTr), _lib.ld(PhiTr), _lib.ptr(X), _lib.ld(X), 0, 0, st()), "stpy_gemm_nt")
		_lib.check(lib.stpy_trsm_right_lt(dt, M, n, _lib.ptr(L), _lib.ld(L), _lib.ptr(self._winvf), self._winvf.numel(), _lib.ptr(X), _lib.ld(X), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
		_lib.check(lib.stpy_predict(dt, M, n, _lib.ptr(X), _lib.ld(X), _lib.ptr(self._u), None, _lib.ptr(mu), _lib.ptr(ss), 2, st()), "stpy_predict")
		kd = torch.empty_like(ss)                                         # |phi*|^2: the same row-sum kernel on Phi*
		_lib.check(lib.stpy_predict(dt, M, m, _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(self._theta), None, None, _lib.ptr(kd), 2, st()), "stpy_predict")
		_lib.check(lib.stpy_predict_finish(dt, M, None, _lib.ptr(ss), _lib.ptr(kd), 1.0, _lib.ptr(std), 1 if self.clamp_variance else 0, st()), "stpy_predict_finish")
		if float(self.lam) != 1.0:                                        # ... / sqrt(lam): the epilogue's scale on a vector
			_lib.check(lib.stpy_predict_finish(dt, M, _lib.ptr(std), None, None, 1.0 / math.sqrt(float(self.lam)), None, 0, st()), "stpy_predict_finish")
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(std.reshape(-1, 1), xtest))

	mean_var = mean_std

	def ucb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu + np.sqrt(self.beta(delta=delta)) * std

	def lcb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu - np.sqrt(self.beta(delta=delta)) * std

	def residuals(self):
		"""kernelized_features.py:559-562: sum (mean(x) - y)^2."""
		lib = _lib.load()
		mu, _ = self.mean_std(self.x)
		r = _lib.to_device(mu).reshape(-1, 1).clone()
		y = _lib.to_device(self.y, r.dtype).reshape(-1, 1).contiguous()
		dt = _lib.dtype_code(r.dtype)
		one = torch.ones((1, 1), dtype=r.dtype, device=r.device)
		n = r.shape[0]
		# r -= y 1^T (a K = 1 product in the subtracting mode), then <r, r>
		_lib.check(lib.stpy_gemm_nt(dt, n, 1, 1, _lib.ptr(y), 1, _lib.ptr(one), 1, _lib.ptr(r), 1, 1, 0, _lib.stream_ptr()), "stpy_gemm_nt")
		td = torch.empty((2,), dtype=r.dtype, device=r.device)
		_lib.check(lib.stpy_trace_dot(dt, n, None, 0, _lib.ptr(r), _lib.ptr(r), _lib.ptr(td), _lib.stream_ptr()), "stpy_trace_dot")
		return _lib.like_input(td[1].reshape(()).clone(), self.x)

	# ------------------------------------------------------------------ sampling
	def _draw(self, basis, size):
		"""The reference's draw (kernelized_features.py:302-303, :323-324): CPU generator, (basis, size) standard normals."""
		zeros = torch.zeros(size=(basis, size), dtype=torch.float64)
		return torch.normal(mean=zeros, std=1.)

	def _prior_theta_t(self, random_vector, dtype, device):
		"""theta^T (size, basis) of the prior branch: chol(lam I) r + prior_mean = sqrt(lam) r + prior_mean (:305-307, :332-334).
		The scaling is applied to the host-side draw before it is uploaded."""
		th = math.sqrt(float(self.lam)) * random_vector
		if torch.is_tensor(self.prior_mean) or self.prior_mean != 0:
			th = th + self.prior_mean
		return th.T.contiguous().to(device=device, dtype=dtype)

	def _sample_theta_t(self, size=1, prior=False):
		"""theta^T on the device, (size, basis)."""
		lib = _lib.load()
		basis = self.get_basis_size()
		random_vector = self._draw(basis, size)
		self.precompute()
		if not (self.fitted == True and prior == False):
			dev = _lib.device()
			dtype = self._Lf.dtype if self._Lf is not None else torch.float64
			return self._prior_theta_t(random_vector, dtype, dev)
		# L = chol(get_invV()) * s, theta = theta_mean + L r  (:328-330).  theta^T = 1 theta_mean^T + (s r)^T L^T: the accumulating
		# NT product of the scaled draw (size x basis) with the lower-triangular factor
		invV = self._invV_device()
		dt = _lib.dtype_code(invV.dtype)
		st = _lib.stream_ptr
		Lc, _ = self._chol(invV, "V^-1 (sample_theta)")
		_lib.check(lib.stpy_tril(dt, basis, _lib.ptr(Lc), _lib.ld(Lc), st()), "stpy_tril")
		rt = (float(self.s) * random_vector).T.contiguous().to(device=Lc.device, dtype=Lc.dtype)          # (size, basis)
		thT = torch.empty((size, basis), dtype=Lc.dtype, device=Lc.device)
		thT.copy_(self._theta.reshape(1, basis).expand(size, basis))
		_lib.check(lib.stpy_gemm_nt(dt, size, basis, basis, _lib.ptr(rt), _lib.ld(rt), _lib.ptr(Lc), _lib.ld(Lc), _lib.ptr(thT), _lib.ld(thT), 2, 0, st()), "stpy_gemm_nt")
		return thT

	def sample_theta(self, size=1, prior=False):
		"""kernelized_features.py:319-336: (basis, size)."""
		thT = self._sample_theta_t(size=size, prior=prior)
		return _lib.like_input(thT.t(), self.x if self.x is not None else torch.zeros(1))

	def _features_times_theta(self, xtest, thT, out=None, mode=0):
		"""Phi(xtest) theta, (M, size), device (``out`` given: accumulated per ``mode``)."""
		lib = _lib.load()
		xt = _lib.to_device(xtest, thT.dtype)
		Phi = _lib.to_device(self.embedding.embed(xt), thT.dtype)
		if Phi.stride(1) != 1:
			Phi = Phi.contiguous()
		M, m = Phi.shape
		size = thT.shape[0]
		if out is None:
			out = torch.empty((M, size), dtype=thT.dtype, device=thT.device)
		_lib.check(lib.stpy_gemm_nt(_lib.dtype_code(thT.dtype), M, size, m, _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(thT), _lib.ld(thT), _lib.ptr(out), _lib.ld(out), mode, 0,
									_lib.stream_ptr()), "stpy_gemm_nt")
		return out

	def sample(self, xtest, size=1, prior=False):
		"""kernelized_features.py:537-543: Phi(xtest) theta for a sampled theta, (M, size)."""
		thT = self._sample_theta_t(size=size, prior=prior)
		return _lib.like_input(self._features_times_theta(xtest, thT), xtest)

	def sample_and_max(self, xtest, size=1):
		"""kernelized_features.py:545-551."""
		f = self.sample(xtest, size=size)
		index = torch.argmax(f, dim=0)
		return (xtest[index, :], f[index, :])

	def sample_matheron(self, xtest, kernel_object, size=1):
		"""
		kernelized_features.py:300-317 (pathwise / Matheron update): a prior draw in feature space, corrected by the exact GP of
		``kernel_object`` on the data:  f = Phi* theta + K* (K + s^2 lam I)^-1 (y - Phi theta).
		"""
		lib = _lib.load()
		basis = self.get_basis_size()
		random_vector = self._draw(basis, size)
		xd = _lib.to_device(self.x)
		dtype, dev = xd.dtype, xd.device
		dt = _lib.dtype_code(dtype)
		st = _lib.stream_ptr
		thT = self._prior_theta_t(random_vector, dtype, dev)                                    # (size, basis)
		xt = _lib.to_device(xtest, dtype)
		N, M = xd.shape[0], xt.shape[0]
		f = self._features_times_theta(xt, thT)                                                 # f_prior_xtest (M, size)
		# R^T = 1 y^T - theta^T Phi^T  (size, N): the residual of the prior draw on the data, rows = right-hand sides
		Rt = torch.empty((size, N), dtype=dtype, device=dev)
		Rt.copy_(_lib.to_device(self.y, dtype).reshape(1, N).expand(size, N))
		Phi = _lib.to_device(self.embedding.embed(xd), dtype)
		if Phi.stride(1) != 1:
			Phi = Phi.contiguous()
		_lib.check(lib.stpy_gemm_nt(dt, size, N, basis, _lib.ptr(thT), _lib.ld(thT), _lib.ptr(Phi), _lib.ld(Phi), _lib.ptr(Rt), _lib.ld(Rt), 1, 0, st()), "stpy_gemm_nt")
		del Phi
		# K = k(x, x) + s^2 lam I -> Cholesky;  f += (K* L^-T) (R^T L^-T)^T
		K = torch.empty((N, N), dtype=dtype, device=dev)
		kernel_object._kernel_into(xd, xd, K, None, diag_add=float(self.s) ** 2 * float(self.lam), lower_only=True)
		L, winv = self._chol(K, "k(x, x) + s^2 lam I (sample_matheron)")
		_lib.check(lib.stpy_trsm_right_lt(dt, size, N, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(Rt), _lib.ld(Rt), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
		X = torch.empty((M, N), dtype=dtype, device=dev)
		kernel_object._kernel_into(xd, xt, X)                                                   # K* = k(x, xtest): (M, N)
		_lib.check(lib.stpy_trsm_right_lt(dt, M, N, _lib.ptr(L), _lib.ld(L), _lib.ptr(winv), winv.numel(), _lib.ptr(X), _lib.ld(X), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
		_lib.check(lib.stpy_gemm_nt(dt, M, size, N, _lib.ptr(X), _lib.ld(X), _lib.ptr(Rt), _lib.ld(Rt), _lib.ptr(f), _lib.ld(f), 2, 0, st()), "stpy_gemm_nt")
		return _lib.like_input(f, xtest)
