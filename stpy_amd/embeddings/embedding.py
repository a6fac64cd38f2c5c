"""
Drop-in for ``stpy.embeddings.embedding.{Embedding, RFFEmbedding, QuadratureEmbedding, HermiteEmbedding, ...}``
(reference: stpy/embeddings/embedding.py:53-129 base class, :139-241 RFF, :250-466 quadrature base, :501-670 the
node / weight rules).  ``embed`` runs in ``stpy_rff_embed`` (csrc/rff.hip); weight sampling (RFF: global numpy RNG)
and the quadrature node / weight tables (a few hundred numbers) stay on the host exactly as in the reference, and
``W`` / ``b`` / ``weights`` may be injected.
"""
import numpy as np
import torch

from .. import _lib
from ..helpers.helper import cartesian


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
		# (large fp32 d = 64 shapes: a workspace for the split W lets the contraction run on the bf16 matrix cores; 0 bytes otherwise)
		wb = 0 if transposed else int(lib.stpy_rff_workspace_bytes(_lib.dtype_code(xd.dtype), times, d, self.m))
		work = torch.empty((wb,), dtype=torch.uint8, device=xd.device) if wb > 0 else None
		rc = lib.stpy_rff_embed(_lib.dtype_code(xd.dtype), _lib.ptr(xd), times, xd.stride(0), d, _lib.ptr(Wd), Wd.stride(0),
								self.m, _lib.ptr(bd), None, scale, _lib.ptr(out), out.stride(0), 1 if transposed else 0,
								_lib.ptr(work), wb, _lib.stream_ptr())
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


class QuadratureEmbedding(Embedding):
	"""
	Quadrature Fourier features on a tensor grid (embedding.py:250-466): q one-dimensional nodes omega_k with weights
	w_k, all q^d combinations as frequencies W (q^d, d) with product weights, and

	    Phi(x) = sqrt(kappa) [ sqrt(w_j) cos(<W_j, x>)  ;  sqrt(w_j) sin(<W_j, x>) ]        (n, 2 q^d)

	(cos and sin of the SAME node, unlike the RFF layout; ``cosine=True`` keeps the cosine half only).  The base class
	integrates the spectral density by Gauss-Legendre nodes mapped to the half line through omega = scale / tan(t)
	(:425-448); derived classes replace ``nodesAndWeights``.  On the device this is ``stpy_rff_embed`` with the
	frequency table stacked twice and the amplitudes sqrt(w_j) as per-feature scale.
	"""

	def __init__(self, scale=1.0, **kwargs):
		Embedding.__init__(self, **kwargs)
		self.scale = scale
		self.compute()

	def reorder_complexity(self, omegas, weights):
		"""embedding.py:260-265: nodes sorted by magnitude."""
		order = np.argsort(np.abs(omegas))
		return omegas[order], weights[order]

	def transform(self):
		"""embedding.py:393-423: the spectral density handed to the node rules (squared exponential; the laplace /
		modified-Matern densities are kept for MaternEmbedding)."""
		g = self.gamma
		if self.kernel == "squared_exponential":
			return lambda om: np.exp(-np.sum(om ** 2, axis=1).reshape(-1, 1) / 2 * (g ** 2)) * (g / np.sqrt(2 * np.pi)) * (np.pi / 2)
		if self.kernel == "laplace":
			return lambda om: np.prod(1. / ((g ** 2) * (om ** 2) + 1.), axis=1).reshape(-1, 1) * (g / 2.)
		if self.kernel == "modified_matern" and self.nu in (2, 3, 4):
			c = {2: 1.0, 3: 4.0 / 3.0, 4: 8.0 / 5.0}[self.nu]
			return lambda om: np.prod(1. / ((g ** 2) * (om ** 2) + 1.) ** self.nu, axis=1).reshape(-1, 1) * (g * c)
		raise NotImplementedError("no spectral density for kernel '%s' (nu=%s)" % (self.kernel, self.nu))

	def nodesAndWeights(self, q):
		"""embedding.py:425-448: 2q-point Gauss-Legendre, upper half, mapped by omega = scale * cot(pi (t + 1) / 2)."""
		t, w = np.polynomial.legendre.leggauss(2 * q)
		t, w = t[q:], 2 * w[q:]
		ang = ((t + 1.) / 2.) * np.pi
		omegas = self.scale / np.tan(ang)
		dens = self.transform()
		weights = self.scale * (1. / np.sin(ang) ** 2) * w * dens(omegas.reshape(-1, 1)).flatten()
		return omegas, weights

	def compute(self, complexity_reorder=True):
		"""embedding.py:366-391: tensor grid of the 1-D rule; m becomes 2 q^d (q^d with ``cosine``)."""
		base = self.m if self.cosine else self.m // 2
		self.q = int(np.power(base, 1. / self.d))
		self.m = self.q ** self.d
		omegas, weights = self.nodesAndWeights(self.q)
		if complexity_reorder:
			omegas, weights = self.reorder_complexity(omegas, weights)
		self.weights = torch.from_numpy(np.prod(cartesian([weights] * self.d), axis=1))
		self.W = torch.from_numpy(cartesian([omegas] * self.d))
		if not self.cosine:
			self.m = self.m * 2

	def _operands(self, dtype, d):
		"""(frequency rows, amplitudes, bias) of the device call for this embedding."""
		Wd = _lib.to_device(self.W, dtype)[:, :d]
		amp = torch.sqrt(_lib.to_device(self.weights, dtype).reshape(-1))
		if self.cosine:          # all-cosine: the biased form of the kernel with a zero phase
			return Wd.contiguous(), amp, torch.zeros_like(amp)
		return torch.cat([Wd, Wd]).contiguous(), torch.cat([amp, amp]), None

	def _embed_device(self, xd, transposed):
		lib = _lib.load()
		times, d = xd.shape
		Wd, amp, bias = self._operands(xd.dtype, d)
		m = Wd.shape[0]
		out = torch.empty((m, times) if transposed else (times, m), dtype=xd.dtype, device=xd.device)
		rc = lib.stpy_rff_embed(_lib.dtype_code(xd.dtype), _lib.ptr(xd), times, xd.stride(0), d, _lib.ptr(Wd), Wd.stride(0), m,
								_lib.ptr(bias), _lib.ptr(amp), float(np.sqrt(self.kappa)), _lib.ptr(out), out.stride(0), 1 if transposed else 0, None, 0, _lib.stream_ptr())
		_lib.check(rc, "stpy_rff_embed")
		return out

	def embed(self, x):
		"""embedding.py:450-466: (n, d) -> (n, m)."""
		return _lib.like_input(self._embed_device(_lib.to_device(x), False), x)

	def embed_t(self, x):
		"""Phi^T (m, n), the operand layout of the feature-space normal equations (see KernelizedFeatures)."""
		return _lib.like_input(self._embed_device(_lib.to_device(x), True), x)

	def get_sub_indices(self, group):
		"""embedding.py:468-486."""
		m2 = self.m
		mhalf = int(np.power(self.m // 2, 1. / self.d))
		mquater = mhalf // 2
		if group == 0:
			return (np.arange(mquater * mhalf, (mquater + 1) * mhalf, 1).tolist()
					+ np.arange(m2 // 2 + (mquater * mhalf), m2 // 2 + (mquater + 1) * mhalf, 1).tolist())
		return np.arange(mquater, m2 // 2, mhalf).tolist() + np.arange(m2 // 2 + mquater, m2, mhalf).tolist()


class TrapezoidalEmbedding(QuadratureEmbedding):
	"""embedding.py:508-529: equispaced nodes with step sqrt(pi / q) / gamma^2."""

	def __init__(self, **kwargs):
		QuadratureEmbedding.__init__(self, **kwargs)
		if self.kernel != "squared_exponential":
			raise AssertionError("This embeding is allowed only with Squared Exponential Kernel")

	def nodesAndWeights(self, q):
		dens = self.transform()
		h = np.sqrt(np.pi / q) / self.gamma ** 2
		nodes = np.linspace(-q // 2, q // 2, q) * h
		return nodes, h * dens(nodes.reshape(-1, 1)).flatten() * (2 / np.pi)


class ClenshawCurtisEmbedding(QuadratureEmbedding):
	"""embedding.py:532-553: nodes cot(pi k / (q + 2)) / gamma."""

	def __init__(self, **kwargs):
		QuadratureEmbedding.__init__(self, **kwargs)
		if self.kernel != "squared_exponential":
			raise AssertionError("This embeding is allowed only with Squared Exponential Kernel")

	def nodesAndWeights(self, q):
		Lg = 1. / self.gamma
		dens = self.transform()
		ang = np.pi * np.linspace(0, q + 1, q + 2)[1:-1] / (q + 2)
		nodes = Lg / np.tan(ang)
		weights = Lg * (np.pi / (q + 2)) * (1. / np.sin(ang) ** 2)
		return nodes, weights * dens(nodes.reshape(-1, 1)).flatten() * (2. / np.pi)


class HermiteEmbedding(QuadratureEmbedding):
	"""Gauss-Hermite quadrature Fourier features for the squared exponential kernel (embedding.py:578-607)."""

	def __init__(self, ones=False, cosine=False, **kwargs):
		self.ones = ones
		# reference quirk kept (embedding.py:583-586): the keyword is consumed here and the base constructor then resets
		# self.cosine to its own default, so HermiteEmbedding(cosine=True) still yields the cos | sin layout
		self.cosine = cosine
		QuadratureEmbedding.__init__(self, **kwargs)
		if self.kernel != "squared_exponential":
			raise AssertionError("Hermite Embedding is allowed only with Squared Exponential Kernel")

	def nodesAndWeights(self, q):
		nodes, weights = np.polynomial.hermite.hermgauss(2 * q)
		nodes, weights = nodes[q:], 2 * weights[q:]          # positive half; the negative nodes are the sine partners
		if self.ones == True:
			weights = np.ones(q)
		return np.sqrt(2) * nodes / self.gamma, weights / np.sqrt(np.pi)


class OverCompleteHermiteEmbedding(HermiteEmbedding):
	"""embedding.py:610-626: all q Hermite nodes (both signs)."""

	def nodesAndWeights(self, q):
		nodes, weights = np.polynomial.hermite.hermgauss(q)
		return np.sqrt(2) * nodes / self.gamma, weights / np.sqrt(np.pi)


class MaternEmbedding(QuadratureEmbedding):
	"""embedding.py:629-651 (the base-class constructor already evaluates its own node rule, i.e. needs the density of
	'laplace' / 'modified_matern' above)."""

	def __init__(self, **kwargs):
		super().__init__(**kwargs)
		if self.kernel != "modified_matern" and self.kernel != "laplace":
			raise AssertionError("Matern Embedding is allowed only with Matern Kernel")

	def nodesAndWeights(self, q):
		nodes, weights = np.polynomial.hermite.hermgauss(q)
		return np.sqrt(2) * nodes / self.gamma, weights / np.sqrt(np.pi)


class LatticeEmbedding(QuadratureEmbedding):
	"""embedding.py:686-707: integer lattice sqrt(2) k / gamma with uniform weights."""

	def nodesAndWeights(self, q):
		return np.sqrt(2) * np.arange(1, q + 1, 1) / self.gamma, np.ones(q) / (2 * q)


class ConcatEmbedding(Embedding):
	"""embedding.py:710-717."""

	def __init__(self, embeddings):
		self.embeddings = embeddings
		self.m = sum([emb.get_m() for emb in embeddings])

	def embed(self, xtest):
		return torch.hstack([emb.embed(xtest) for emb in self.embeddings])
