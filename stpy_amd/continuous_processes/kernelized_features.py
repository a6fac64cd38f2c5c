"""
Drop-in for ``stpy.continuous_processes.kernelized_features.KernelizedFeatures`` in its primal form
(reference: kernelized_features.py:12-52 ctor, :81-100 embed / kernel,
:118-138 fit_gp, :176-246 precompute, :248-267 theta_mean, :269-288 mean_std).

Ridge regression on a finite feature map Phi (n x m), e.g. random Fourier features:
    V = Phi^T Phi + s^2 lam I,   theta = V^-1 Phi^T y,   mean = Phi* theta,
    std = s sqrt(diag(Phi* V^-1 Phi*^T)).
Device mapping -- every contraction is the NT MFMA GEMM because the embedding is produced
TRANSPOSED (Phi^T, m x n: ``embed_t``):
    V           stpy_gemm_nt(Phi^T, Phi^T)  lower triangle  (+ s^2 lam on the diagonal)
    V = L L^T   stpy_potrf          (the reference takes pinverse(V); V is SPD for s, lam > 0)
    Phi^T y     stpy_predict (row sums against y),   theta: stpy_trsv forward + backward
    X = Phi* L^-T      stpy_trsm_right_lt,   mean = X (L^-1 Phi^T y),  std = s sqrt(rowsum(X o X))
The dual form (n < m with primal=False), the Woodbury/Schur rank-one updates of ``add_data_point``
(refit here), Matheron sampling, and the cvxpy-based constrained fits are outside the hot path.
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
		self._L = self._winv = self._u = self._theta = None

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
		"""kernelized_features.py:107-112 (the reference queues a rank-one update; here: concatenate and refit)."""
		if self.n == 0:
			self.fit_gp(x, y)
		else:
			self.fit_gp(torch.cat((self.x, x), dim=0), torch.cat((self.y, y), dim=0))

	def fit(self, x=None, y=None):
		self.fit_gp(self.x if x is None else x, self.y if y is None else y)

	def fit_gp(self, x, y):
		"""kernelized_features.py:118-138 + :236-240."""
		lib = _lib.load()
		self.x, self.y = x, y
		self.n = list(x.size())[0]
		self.d = list(x.size())[1]
		self.data = True
		xd = _lib.to_device(x)
		yd = _lib.to_device(y, xd.dtype).reshape(-1)
		PhiT = _lib.to_device(self._embed_t(xd), xd.dtype)            # (m, n)
		m, n = PhiT.shape
		dt = _lib.dtype_code(PhiT.dtype)
		st = _lib.stream_ptr
		V = torch.empty((m, m), dtype=PhiT.dtype, device=PhiT.device)
		_lib.check(lib.stpy_gemm_nt(dt, m, m, n, _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(V), V.stride(0), 0, 1, st()), "stpy_gemm_nt")
		V.diagonal().add_(float(self.s) ** 2 * float(self.lam))
		self._Vlow = V.clone()                                         # lower triangle of V, for the ``V`` property
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(m)),), dtype=V.dtype, device=V.device)
		work = torch.empty((int(lib.stpy_potrf_workspace_bytes(dt, m, self.nb)),), dtype=torch.uint8, device=V.device)
		info = torch.zeros((1,), dtype=torch.int32, device=V.device)
		_lib.check(lib.stpy_potrf(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, 0, _lib.ptr(info), st()), "stpy_potrf")
		bad = int(info.item())
		if bad != 0:
			raise torch.linalg.LinAlgError("KernelizedFeatures: Phi^T Phi + s^2 lam I is not positive definite (leading minor %d)" % bad)
		# rhs = Phi^T y (row sums of Phi^T against y), u = L^-1 rhs, theta = L^-T u
		rhs = torch.empty((m,), dtype=V.dtype, device=V.device)
		_lib.check(lib.stpy_predict(dt, m, n, _lib.ptr(PhiT), PhiT.stride(0), _lib.ptr(yd), None, _lib.ptr(rhs), None, 0, st()), "stpy_predict")
		u = torch.empty_like(rhs)
		_lib.check(lib.stpy_trsv(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(rhs), _lib.ptr(u), 0, st()), "stpy_trsv")
		scratch = u.clone()
		theta = torch.empty_like(u)
		_lib.check(lib.stpy_trsv(dt, m, _lib.ptr(V), V.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(scratch), _lib.ptr(theta), 1, st()), "stpy_trsv")
		self._L, self._winv, self._u, self._theta = V, winv, u, theta
		self.fitted = True
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
		std = float(self.s) * torch.sqrt(ss)
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(std.reshape(-1, 1), xtest))

	mean_var = mean_std

	def ucb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu + np.sqrt(self.beta(delta=delta)) * std

	def lcb(self, xtest, delta=0.1):
		mu, std = self.mean_std(xtest)
		return mu - np.sqrt(self.beta(delta=delta)) * std
