"""
CPU stand-in for ``stpy_amd.parallel.block_cyclic.HipLocalOps`` -- TEST INFRASTRUCTURE ONLY.
Lets the multi-process tests drive the block-cyclic schedule (index arithmetic + collectives) on
gloo without a GPU.  It re-states, at the same granularity, what the HIP entry points do:
128-tile staircase predicate of ``stpy_gemm_nt_bc``, in-place lower Cholesky + inverse 128-blocks
of ``stpy_potrf``, ``B L^-T`` of ``stpy_trsm_right_lt``.  Kernel values come from the oracle.
"""
import numpy as np
import scipy.linalg as sla
import torch

from oracle import gp_oracle as O

IB = 128


class CpuLocalOps:
	def __init__(self):
		self.device = torch.device("cpu")
		self.dtype = torch.float64

	def empty(self, *shape):
		return torch.full(shape, float("nan"), dtype=self.dtype)      # poison: unwritten reads show up

	def zeros(self, *shape):
		return torch.zeros(shape, dtype=self.dtype)

	def to_device(self, t):
		return torch.as_tensor(t, dtype=self.dtype).contiguous()

	def _eval(self, kernel_object, a, b, kwargs=None):
		items = kernel_object._resolve(dict(kwargs) if kwargs else {})
		out = None
		for it in items:
			g = it['group']
			inv = np.asarray(it['inv_ls'])
			aa, bb = a[:, g] * inv, b[:, g] * inv
			kind = it['kind']
			if kind == 0:
				k = O.squared_exponential(aa, bb, 1.0, it['kappa'])
			elif kind in (1, 2, 3):
				k = O.matern(aa, bb, 1.0, {1: 0.5, 2: 1.5, 3: 2.5}[kind], it['kappa'])
			else:
				k = O.linear(aa, bb, it['kappa'], it['offset'])
			out = k if it['op'] == "-" else (out + k if it['op'] == "+" else out * k)
		return out

	def gram(self, kernel_object, xa, xb, out, kwargs=None, add=False):
		k = torch.from_numpy(self._eval(kernel_object, xa.numpy(), xb.numpy(), kwargs))
		out.copy_(out + k if add else k)

	def add_into(self, out, src):
		out.add_(src)

	def beside_update(self, on):
		pass

	def kdiag(self, kernel_object, xt):
		x = xt.numpy()
		return torch.tensor([self._eval(kernel_object, x[i:i + 1], x[i:i + 1])[0, 0] for i in range(x.shape[0])], dtype=self.dtype)

	def potrf(self, A):
		n = A.shape[0]
		a = A.numpy()
		low = np.tril(a) + np.tril(a, -1).T
		info = torch.zeros((1,), dtype=torch.int32)
		try:
			L = np.linalg.cholesky(low)
		except np.linalg.LinAlgError:
			info[0] = 1
			L = np.eye(n)
		A.copy_(torch.from_numpy(np.tril(L) + np.triu(a, 1)))       # strict upper triangle untouched (scratch)
		nb = (n + IB - 1) // IB
		winv = np.zeros((nb, IB, IB))
		for bi in range(nb):
			c, cb = bi * IB, min(IB, n - bi * IB)
			winv[bi] = np.eye(IB)
			winv[bi][:cb, :cb] = np.linalg.inv(L[c:c + cb, c:c + cb])
		return torch.from_numpy(winv.reshape(-1)), info

	def trsm_right_lt(self, B, L, winv):
		if B.shape[0] == 0:
			return
		Lm = np.tril(L.numpy())
		B.copy_(torch.from_numpy(sla.solve_triangular(Lm, B.numpy().T, lower=True).T))

	def gemm_nt(self, A, B, C, mode, bc=None):
		m, k = A.shape
		n = B.shape[0]
		if m == 0 or n == 0 or k == 0:
			return
		prod = A @ B.T
		if bc is None:
			C.copy_(prod if mode == 0 else C - prod)
			return
		nbd, pr, pc, myr, myc, i0, j0 = bc
		nbt = nbd // IB
		for ti in range((m + IB - 1) // IB):
			for tj in range((n + IB - 1) // IB):
				I = (ti // nbt + i0) * pr + myr
				J = (tj // nbt + j0) * pc + myc
				if I < J or (I == J and tj % nbt > ti % nbt):
					continue
				rs, cs = slice(ti * IB, min(m, ti * IB + IB)), slice(tj * IB, min(n, tj * IB + IB))
				C[rs, cs] = prod[rs, cs] if mode == 0 else C[rs, cs] - prod[rs, cs]

	def row_sums(self, X, z, out=None):
		if out is None:
			out = self.empty(2, X.shape[0])
		out[0] = X @ z
		out[1] = (X * X).sum(dim=1)
		return out[0], out[1]

	def predict_finish(self, mu, sumsq, kdiag, scale, clamp, want_sigma=True):
		mu *= scale
		if not want_sigma:
			return mu, None
		var = kdiag - scale * sumsq
		if clamp:
			var = var.clamp(min=0)
		return mu, torch.sqrt(var)

	def logdet(self, L):
		return torch.log(torch.diagonal(L)).sum()
