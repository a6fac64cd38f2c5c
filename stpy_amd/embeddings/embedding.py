"""
Drop-in for ``stpy.embeddings.embedding.{Embedding, RFFEmbedding}`` (reference:
stpy/embeddings/embedding.py:53-129 base class, :139-241 RFF).  ``embed`` runs in
``stpy_rff_embed`` (csrc/rff.hip); weight sampling stays on the host exactly as in the
reference (global numpy RNG), and ``W`` / ``b`` may be injected.
"""
import numpy as np
import torch

from .. import _lib


class Embedding():
	"""embedding.py:53-129."""

	def __init__(self, gamma=0.1, nu=0.5, m=100, d=1, diameter=1.0, groups=None, kappa=1.0,
				 kernel="squared_exponential", cosine=False, approx="rff", **kwargs):
		self.gamma = float(gamma)
		self.n = nu
		self.m = int(m)
		self.d = int(d)
		self.nu = nu
		self.kappa = kappa
		self.cosine = cosine
		self.diameter = diameter
		self.groups = groups
		self.kernel = kernel
		self.approx = approx
		self.gradient_avail = 0
		if self.m % 2 == 1:
			raise AssertionError("Number of random features has to be even.")

	def sample(self):
		raise AttributeError("Only derived classes can call this method.")

	def embed(self, x):
		raise AttributeError("Only derived classes can call this method.")

	def get_m(self):
		return self.m


class RFFEmbedding(Embedding):
	"""Random Fourier features, embedding.py:139-241."""

	def __init__(self, biased=False, **kwargs):
		super().__init__(**kwargs)
		self.biased = biased
		self.sample()

	def sampler(self, size):
		"""embedding.py:149-210.  Only the samplers that work in the reference snapshot are kept:
		SE + "rff" (N(0,1)/gamma) and "orf"; the others call helpers that do not exist there."""
		if self.kernel != "squared_exponential":
			raise NotImplementedError("RFF sampling for kernel '%s' is outside the stpy_amd hot path" % self.kernel)
		if self.approx == "rff":
			self.W = np.random.normal(size=size) * (1. / self.gamma)                  # embedding.py:159,191
		elif self.approx == "orf":
			from scipy.stats import chi
			self.W = np.random.normal(size=size) * (1.)                               # embedding.py:200-208
			self.Q, _ = np.linalg.qr(self.W)
			self.S = np.diag(chi.rvs(size[1], size=size[0]))
			self.W = np.dot(self.S, self.Q) / self.gamma ** 2
		else:
			raise NotImplementedError("approx='%s' is outside the stpy_amd hot path" % self.approx)
		return self.W

	def sample(self):
		"""embedding.py:212-223."""
		self.W = self.sampler(size=(self.m, self.d))
		self.W = torch.from_numpy(self.W)
		if self.biased == True:
			self.b = 2. * np.pi * np.random.uniform(size=(self.m))
			self.bs = self.b.reshape(self.m, 1)
			self.b = torch.from_numpy(self.b)
			self.bs = torch.from_numpy(self.bs)

	def _embed_device(self, xd, transposed):
		lib = _lib.load()
		(times, d) = xd.shape
		Wd = _lib.to_device(self.W, xd.dtype)
		bd = _lib.to_device(self.b, xd.dtype) if self.biased == True else None
		shape = (self.m, times) if transposed else (times, self.m)
		out = torch.empty(shape, dtype=xd.dtype, device=xd.device)
		scale = float(np.sqrt(2. / float(self.m)) * np.sqrt(self.kappa))
		rc = lib.stpy_rff_embed(_lib.dtype_code(xd.dtype), _lib.ptr(xd), times, xd.stride(0), d, _lib.ptr(Wd), Wd.stride(0),
								self.m, _lib.ptr(bd), scale, _lib.ptr(out), out.stride(0), 1 if transposed else 0, _lib.stream_ptr())
		_lib.check(rc, "stpy_rff_embed")
		return out

	def embed(self, x):
		"""
		embedding.py:225-241.  x: (n, d_x) -> (n, m); uses W[:, 0:d_x].  With ``biased=True`` the
		reference transposes twice (:232 and :241) and returns (m, n); that orientation is kept.
		"""
		out = self._embed_device(_lib.to_device(x), False)
		if self.biased == True:
			out = torch.t(out)
		return _lib.like_input(out, x)

	def embed_t(self, x):
		"""Phi^T, (m, n): the operand layout of the feature-space solves (no reference counterpart; used by
		KernelizedFeatures so that Phi^T Phi is an NT contraction)."""
		return _lib.like_input(self._embed_device(_lib.to_device(x), True), x)
