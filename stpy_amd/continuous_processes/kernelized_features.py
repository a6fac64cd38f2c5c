"""
Drop-in for ``stpy.continuous_processes.kernelized_features.KernelizedFeatures`` in its primal form
(reference: kernelized_features.py:12-52 ctor, :81-100 embed / kernel,
:118-138 fit_gp, :176-246 precompute, :248-267 theta_mean, :269-288 mean_std).

Ridge regression on a finite feature map Phi (n x m), e.g. random Fourier features:
    V = Phi^T Phi + s^2 lam I,   theta = V^-1 Phi^T y,   mean = Phi* theta,
    std = s sqrt(diag(Phi* V^-1 Phi*^T)).
Device mapping -- every contraction is the NT MFMA GEMM because the embedding is produced
TRANSPOSED (Phi^T, m x n: ``embed_t``), and Phi is STREAMED: row slabs of x are embedded one at a time (``slab_bytes`` of
features live, 2 GB by default) and accumulated, so the n x m feature matrix is never materialised (SURVEY.md section 8f
rank 2: at BASELINE config 5's shape it would be 34 GB):
    V          += Phi_slab^T Phi_slab   stpy_gemm_nt(Phi_slab^T, Phi_slab^T, mode "+=")  lower triangle, then + s^2 lam on the diagonal
    Phi^T y    += Phi_slab^T y_slab     stpy_predict (row sums against y_slab) + stpy_combine(ADD)
    V = L L^T   stpy_potrf          (the reference takes pinverse(V); V is SPD for s, lam > 0)
    Phi^T y     stpy_predict (row sums against y),   theta: stpy_trsv forward + backward
    X = Phi* L^-T      stpy_trsm_right_lt,   mean = X (L^-1 Phi^T y),  std = s sqrt(rowsum(X o X))
The dual form (n < m with primal=False), the Woodbury/Schur rank-one updates of ``add_data_point``
(here: the accumulated normal equations are extended by the new rows and refactored), Matheron sampling, and the cvxpy-based
constrained fits are outside the hot path.
"""
import numpy as np
import torch

from .. import _lib
from ..kernels import KernelFunction


class KernelizedFeatures:

	def __init__(self, embedding, m, s=0.001, lam=1., d=1, diameter=1.0, theta_norm=1.0, verbose=True, groups=None,
				 bounds=None, scale=1.0, kappa=1.0, poly=2, primal=True, beta_fun=None, bound=1):
		if not primal:
			raise NotImplementedError("only the primal form of KernelizedFeatures is on the stpy_amd path")
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
		self.prior_mean = 0
		self.dual = False
		self.beta_fun = beta_fun
		self.bound = bound
		self.nb = 0
		self.slab_bytes = 2 << 30          # features held at a time while V and Phi^T y are accumulated (fit_gp)
		self._L = self._winv = self._u = self._theta = None
		self._Vacc = self._rhs = self._part = None          # accumulated Phi^T Phi (lower tiles) and Phi^T y: what add_data_point extends

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

	def beta(self, delta=0.1, norm=None):
		if self.beta_fun is None:
			return 2.0
		raise NotImplementedError("beta_fun variants are outside the stpy_amd path")

	def _embed_t(self, xd):
		"""Phi^T on the device, (m, n)."""
		if hasattr(self.embedding, "embed_t"):
			return self.embedding.embed_t(xd)
		return _lib.to_device(self.embedding.embed(xd)).T.contiguous()       # generic embeddings: one transpose copy

	def kernel(self, x, y):
		"""kernelized_features.py:96-100: linear kernel of the embeddings, (|y|, |x|).  Reference quirk kept: its
		linear kernel object is built with the default d=1, so group=[0] and only the FIRST feature enters (:49)."""
		lib = _lib.load()
		ex = _lib.to_device(self.embed(_lib.to_device(x)))[:, :1].contiguous()
		ey = _lib.to_device(self.embed(_lib.to_device(y)), ex.dtype)[:, :1].contiguous()
		out = torch.empty((ey.shape[0], ex.shape[0]), dtype=ex.dtype, device=ex.device)
		_lib.check(lib.stpy_gemm_nt(_lib.dtype_code(ex.dtype), ey.shape[0], ex.shape[0], ex.shape[1], _lib.ptr(ey), ey.stride(0), _lib.ptr(ex), ex.stride(0),
									_lib.ptr(out), out.stride(0), 0, 0, _lib.stream_ptr()), "stpy_gemm_nt")
		return _lib.like_input(out, x)

	# ------------------------------------------------------------------ fit
	def add_data_point(self, x, y):
		"""kernelized_features.py:107-112 (the reference queues rank-one updates of V^-1).  Here the accumulated normal equations are
		kept (Phi^T Phi and Phi^T y, see fit_gp), so k new rows cost their embedding, one k-deep `+=` product and the m x m
		refactorisation -- not a pass over all n rows."""
		if self.n == 0 or self._Vacc is None:
			self.fit_gp(x, y)
			return
		self.x = torch.cat((self.x, x), dim=0)
		self.y = torch.cat((self.y, y), dim=0)
		self.n = list(self.x.size())[0]
		xd = _lib.to_device(x, self._Vacc.dtype)
		yd = _lib.to_device(y, self._Vacc.dtype).reshape(-1)
		self._accumulate(xd, yd, first=False)
		self._solve_normal_equations()

	def fit(self, x=None, y=None):
		self.fit_gp(self.x if x is None else x, self.y if y is None else y)

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
			V = self._Vacc
			# V (+)= Phi_slab^T Phi_slab, lower tiles only: mode 0 for the first slab of a fit, 2 (accumulate) afterwards
			_lib.check(lib.stpy_gemm_nt(dt, m, m, take, _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(V), V.stride(0),
										0 if first else 2, 1, st()), "stpy_gemm_nt")
			# Phi_slab^T y_slab: row sums of Phi^T against y
			ys = yd[r0:r0 + take]
			tgt = self._rhs if first else self._part
			_lib.check(lib.stpy_predict(dt, m, take, _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(ys), None, _lib.ptr(tgt), None, 0, st()), "stpy_predict")
			if not first:
				_lib.check(lib.stpy_combine(dt, 1, m, _lib.ptr(self._rhs), m, _lib.ptr(self._part), m, _lib.OUT_ADD, 0.0, st()), "stpy_combine")
			first = False
			r0 += take
			del PhiT

	def _solve_normal_equations(self):
		"""V = V_acc + s^2 lam I -> Cholesky, u = L^-1 (Phi^T y), theta = L^-T u."""
		lib = _lib.load()
		st = _lib.stream_ptr
		Vacc = self._Vacc
		m = Vacc.shape[0]
		dt = _lib.dtype_code(Vacc.dtype)
		V = torch.empty_like(Vacc)
		# V = V_acc, then + s^2 lam on the diagonal (one pass of the elementwise kernel)
		_lib.check(lib.stpy_combine(dt, m, m, _lib.ptr(V), V.stride(0), _lib.ptr(Vacc), Vacc.stride(0), _lib.OUT_SET, float(self.s) ** 2 * float(self.lam), st()), "stpy_combine")
		self._Vlow = V.clone()                                         # lower triangle of V, for the ``V`` property
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(m)),), dtype=V.dtype, device=V.device)
		work = torch.empty((int(lib.stpy_potrf_workspace_bytes(dt, m, self.nb)),), dtype=torch.uint8, device=V.device)
		info = torch.zeros((1,), dtype=torch.int32, device=V.device)
		_lib.check(lib.stpy_potrf(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, 0, _lib.ptr(info), st()), "stpy_potrf")
		bad = int(info.item())
		if bad != 0:
			self.fitted = False
			raise torch.linalg.LinAlgError("KernelizedFeatures: Phi^T Phi + s^2 lam I is not positive definite (leading minor %d)" % bad)
		rhs = self._rhs.reshape(-1).clone()          # (stpy_trsv uses its right-hand side as scratch; the accumulated one is kept)
		u = torch.empty_like(rhs)
		_lib.check(lib.stpy_trsv(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(rhs), _lib.ptr(u), 0, st()), "stpy_trsv")
		scratch = u.clone()
		theta = torch.empty_like(u)
		_lib.check(lib.stpy_trsv(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(scratch), _lib.ptr(theta), 1, st()), "stpy_trsv")
		self._L, self._winv, self._u, self._theta = V, winv, u, theta
		self.fitted = True

	def fit_gp(self, x, y):
		"""kernelized_features.py:118-138 + :236-240, streaming over row slabs of x (see the module header)."""
		self.x, self.y = x, y
		self.n = list(x.size())[0]
		self.d = list(x.size())[1]
		self.data = True
		xd = _lib.to_device(x)
		yd = _lib.to_device(y, xd.dtype).reshape(-1)
		self.fitted = False
		self._Vacc = self._rhs = self._part = None
		self._accumulate(xd, yd, first=True)
		self._solve_normal_equations()
		return None

	def precompute(self):
		if not self.fitted and self.data:
			self.fit_gp(self.x, self.y)

	@property
	def V(self):
		"""Phi^T Phi + s^2 lam I (kernelized_features.py:239), full symmetric."""
		lib = _lib.load()
		V = self._Vlow.clone()
		_lib.check(lib.stpy_symmetrize_lower(_lib.dtype_code(V.dtype), V.shape[0], _lib.ptr(V), V.stride(0), _lib.stream_ptr()), "stpy_symmetrize_lower")
		return _lib.like_input(V, self.x)

	@property
	def invV(self):
		"""V^-1 (the reference keeps pinverse(V), kernelized_features.py:240); from the factor on demand."""
		lib = _lib.load()
		L = self._L
		m = L.shape[0]
		dt = _lib.dtype_code(L.dtype)
		out = torch.empty((m, m), dtype=L.dtype, device=L.device)
		work = torch.empty((m, m), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_potri(dt, m, _lib.ptr(L), L.stride(0), _lib.ptr(self._winv), self._winv.numel(), _lib.ptr(out), out.stride(0), _lib.ptr(work), work.numel() * work.element_size(), _lib.stream_ptr()), "stpy_potri")
		_lib.check(lib.stpy_symmetrize_lower(dt, m, _lib.ptr(out), out.stride(0), _lib.stream_ptr()), "stpy_symmetrize_lower")
		return _lib.like_input(out, self.x)

	def theta_mean(self, var=False, prior=False):
		"""kernelized_features.py:248-267."""
		self.precompute()
		if self.fitted and not prior:
			theta = _lib.like_input(self._theta.reshape(-1, 1), self.x)
		else:
			theta = 0 * torch.ones(size=(self.get_basis_size(), 1)).double()
		if var is False:
			return theta
		return (theta, float(self.s) ** 2 * self.invV)

	# ------------------------------------------------------------------ predict
	def mean(self, xtest):
		return self.mean_std(xtest)[0]

	def mean_std(self, xtest):
		"""kernelized_features.py:269-288."""
		lib = _lib.load()
		self.precompute()
		L = self._L
		xt = _lib.to_device(xtest, L.dtype)
		Phi = _lib.to_device(self.embedding.embed(xt), L.dtype)          # (M, m): rows = right-hand sides
		if Phi.stride(1) != 1:
			Phi = Phi.contiguous()
		X = Phi.clone()
		M, m = X.shape
		dt = _lib.dtype_code(L.dtype)
		st = _lib.stream_ptr
		_lib.check(lib.stpy_trsm_right_lt(dt, M, m, _lib.ptr(L), L.stride(0), _lib.ptr(self._winv), self._winv.numel(), _lib.ptr(X), X.stride(0), self.nb, 0, None, 0, st()), "stpy_trsm_right_lt")
		mu, ss = torch.empty((M,), dtype=L.dtype, device=L.device), torch.empty((M,), dtype=L.dtype, device=L.device)
		_lib.check(lib.stpy_predict(dt, M, m, _lib.ptr(X), X.stride(0), _lib.ptr(self._u), None, _lib.ptr(mu), _lib.ptr(ss), 2, st()), "stpy_predict")
		# std = s sqrt(ss) = sqrt(0 - (-s^2) ss): the prediction epilogue with a zero prior term (no torch arithmetic on the vectors)
		std = torch.empty_like(ss)
		_lib.check(lib.stpy_predict_finish(dt, M, None, _lib.ptr(ss), _lib.ptr(torch.zeros_like(ss)), -float(self.s) ** 2, _lib.ptr(std), 0, st()), "stpy_predict_finish")
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(std.reshape(-1, 1), xtest))

	mean_var = mean_std

	def ucb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu + np.sqrt(self.beta(delta=delta)) * std

	def lcb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu - np.sqrt(self.beta(delta=delta)) * std
