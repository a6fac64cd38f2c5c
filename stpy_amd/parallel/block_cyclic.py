"""
2-D block-cyclic GP fit / predict across the GPUs of one node (BASELINE config 4; north_star:
"N x N Gram matrix sharded 2D block-cyclic ... panel broadcast ... on RCCL over xGMI").

One process per GPU (torch.distributed, backend "nccl" = RCCL), process grid P = P_r x P_c,
distribution block NB (a multiple of 128).  Global block (I, J) of K = k(x,x) + s^2 I lives on
rank (I mod P_r, J mod P_c) at local block (I div P_r, J div P_c) of that rank's local matrix
``Aloc`` (row-major, contiguous).  x, y and xtest are replicated (N*d*8 B = 33 MB at C4).

fit (right-looking Cholesky, one block column K per step)
  1. rank (K%P_r, K%P_c) factors the NB x NB diagonal block with the single-GPU ``stpy_potrf``
     and broadcasts L_KK + its inverse 128-blocks down its process COLUMN;
  2. the ranks of that process column solve their part of the panel, L_IK = A_IK L_KK^-T
     (``stpy_trsm_right_lt``, in place in Aloc);
  3. panel broadcast along process ROWS: every rank obtains L_IK for its own local block rows I;
  4. "transposed" exchange inside each process column: rank (r', c) owns, after step 3, the blocks
     L_JK with J%P_r == r'; the ones with J%P_c == c are broadcast down the column so every rank
     has L_JK for its own local block columns J;
  5. local trailing update  Aloc[I>K, J>K] -= Prow Pcol^T  by ONE launch of the MFMA GEMM with the
     block-cyclic staircase predicate (``stpy_gemm_nt_bc``) -- no communication.
  Right-looking Cholesky needs no reduction of the trailing update; the only collectives are the
  three broadcasts per step.  Volume per rank and step: (rows/P_r + cols/P_c) * NB * 8 B.

predict  (X = K* L^-T, rows = test points, column blocks distributed like L's and replicated down
each process column)  left-looking, so only M x NB blocks move, never L:
  X_K = (K*_K - sum_{J<K} X_J L_KJ^T) L_KK^-T :  every rank of process row K%P_r forms the partial
  sum over ITS local block columns J < K with one GEMM (the operands are contiguous in local
  storage), the partials are summed onto the diagonal owner (reduce along the process row), the
  owner finishes the block and broadcasts it down its process column.  mu = X z and
  sigma^2 = kdiag - rowsum(X o X) are all-reduced partial sums; z = L^-1 y is the same solve with
  a single right-hand side.

All arithmetic goes through a ``LocalOps`` object.  The product backend is ``HipLocalOps`` (the C
ABI of libstpy_hip; raises without a GPU).  Tests inject a CPU backend from ``tests/`` to exercise
the index arithmetic and the collectives on gloo; nothing here imports the oracle.

Look-ahead: one block column.  After panel K is in place the local update of block column K+1 runs
first; panel K+1 (diagonal factor, panel solve and all three broadcasts) is then issued on a
high-priority side stream while the main stream applies panel K to the remaining columns.

Round-1 status: exercised on gloo with 2 and 4 CPU ranks and with 1/2/4 ranks sharing one MI355X
(host-staged collectives); the RCCL path itself has only been read, not run -- the authoring loop
has a single GPU.  At world size 1 this code path costs 1.91 s for the N=65536 bench step against
1.81 s for the single-GPU class (tools/dist_bench.py).
"""
import ctypes
import math

import numpy as np
import torch
import torch.distributed as dist

from .. import _lib
from ..kernels import KernelFunction

IB = 128


def default_grid(world):
	"""P_r x P_c with P_r the largest divisor of P not above sqrt(P): 1x1, 1x2, 2x2, 2x4."""
	pr = 1
	for c in range(1, int(math.isqrt(world)) + 1):
		if world % c == 0:
			pr = c
	return pr, world // pr


class HipLocalOps:
	"""Local tile arithmetic on this process's GPU through libstpy_hip.  No CPU path."""

	def __init__(self, dtype=torch.float64, nb=0):
		self.lib = _lib.load()
		self.device = _lib.device()
		self.dtype = dtype
		self.code = _lib.dtype_code(dtype)
		self.nb = nb
		self.flags = 0

	def empty(self, *shape):
		return torch.empty(shape, dtype=self.dtype, device=self.device)

	def beside_update(self, on):
		"""Brackets the launches of a panel step that is enqueued while the trailing update occupies the chip:
		potrf / trsm_right_lt below then pass STPY_FLAG_BESIDE_UPDATE (a per-call flag, no process-wide switch).
		The one-volley K = 128 kernel holds 128 KiB of LDS, i.e. it needs a CU with no update workgroup on
		it, and without preemption it then waits for the update's grid to drain (13.8 ms for a 14-workgroup
		product in the trace of N = 65 536); the 32 KiB kernels fit beside one update workgroup."""
		self.flags = _lib.FLAG_BESIDE_UPDATE if on else 0

	def zeros(self, *shape):
		return torch.zeros(shape, dtype=self.dtype, device=self.device)

	def to_device(self, t):
		return _lib.to_device(t, self.dtype)

	def gram(self, kernel_object, xa, xb, out, kwargs=None):
		"""out[j, i] = k(xb_j, xa_i); out may be a strided 2-D view."""
		kernel_object._kernel_into(xa, xb, out, kwargs)

	def kdiag(self, kernel_object, xt):
		out = self.empty(xt.shape[0])
		kernel_object._diag_into(xt, out)
		return out

	def potrf(self, A):
		"""In-place Cholesky of the (strided) square view A; returns (winv, info_tensor)."""
		n = A.shape[0]
		winv = self.empty(int(self.lib.stpy_potrf_winv_elems(n)))
		work = torch.empty((int(self.lib.stpy_potrf_workspace_bytes(self.code, n, self.nb)),), dtype=torch.uint8, device=self.device)
		info = torch.zeros((1,), dtype=torch.int32, device=self.device)
		_lib.check(self.lib.stpy_potrf(self.code, n, _lib.ptr(A), A.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(work), work.numel() * work.element_size(), self.nb, self.flags, _lib.ptr(info), _lib.stream_ptr()), "stpy_potrf")
		return winv, info

	def trsm_right_lt(self, B, L, winv):
		"""B <- B L^-T in place; B: (m, n) strided view, L: (n, n) view, winv from potrf(L)."""
		m, n = B.shape
		if m == 0:
			return
		_lib.check(self.lib.stpy_trsm_right_lt(self.code, m, n, _lib.ptr(L), L.stride(0), _lib.ptr(winv), winv.numel(), _lib.ptr(B), B.stride(0),
											   self.nb, self.flags, None, 0, _lib.stream_ptr()), "stpy_trsm_right_lt")

	def gemm_nt(self, A, B, C, mode, bc=None):
		"""C (mode 0: =, 1: -=) A B^T.  bc = (nb_dist, pr, pc, myr, myc, i0, j0) enables the staircase."""
		m, k = A.shape
		n = B.shape[0]
		if m == 0 or n == 0 or k == 0:
			return
		if bc is None:
			passes = int(self.lib.stpy_gemm_nt_splitk_passes(m, n, k))
			if passes > 1:        # few output tiles, long K (partial sums of the distributed solve)
				work = self.empty(passes * m * n)
				_lib.check(self.lib.stpy_gemm_nt_splitk(self.code, m, n, k, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(C), C.stride(0),
														mode, passes, _lib.ptr(work), work.numel() * work.element_size(), _lib.stream_ptr()), "stpy_gemm_nt_splitk")
				return
			rc = self.lib.stpy_gemm_nt(self.code, m, n, k, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(C), C.stride(0),
									   mode, 0, _lib.stream_ptr())
		else:
			rc = self.lib.stpy_gemm_nt_bc(self.code, m, n, k, _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(C), C.stride(0),
										  mode, *[int(v) for v in bc], _lib.stream_ptr())
		_lib.check(rc, "stpy_gemm_nt")

	def row_sums(self, X, z, out=None):
		"""(sum_k X[i,k] z[k], sum_k X[i,k]^2) for every row of the strided view X; ``out``: a contiguous (2, m) buffer."""
		m, n = X.shape
		if out is None:
			out = self.empty(2, m)
		s1, s2 = out[0], out[1]
		if n == 0:
			out.zero_()
			return s1, s2
		_lib.check(self.lib.stpy_predict(self.code, m, n, _lib.ptr(X), X.stride(0), _lib.ptr(z), None, _lib.ptr(s1), _lib.ptr(s2), 2,
										 _lib.stream_ptr()), "stpy_predict")
		return s1, s2

	def predict_finish(self, mu, sumsq, kdiag, scale, clamp):
		"""mu *= scale (in place); sigma = sqrt(kdiag - scale * sumsq): the epilogue after the all-reduce of the partial sums."""
		sigma = self.empty(mu.shape[0])
		_lib.check(self.lib.stpy_predict_finish(self.code, mu.shape[0], _lib.ptr(mu), _lib.ptr(sumsq), _lib.ptr(kdiag), float(scale), _lib.ptr(sigma),
												1 if clamp else 0, _lib.stream_ptr()), "stpy_predict_finish")
		return mu, sigma

	def logdet(self, L):
		"""sum_i log L_ii of the (strided) factor block L."""
		out2 = self.empty(2)
		_lib.check(self.lib.stpy_logdet_quad(self.code, L.shape[0], _lib.ptr(L), L.stride(0), None, _lib.ptr(out2), _lib.stream_ptr()), "stpy_logdet_quad")
		return out2[0]


class DistributedGaussianProcess:
	"""``GaussianProcess`` on a P_r x P_c process grid: same constructor, ``fit_gp`` / ``mean_std`` /
	``log_marginal`` (default hyper-parameters only), results replicated on every rank."""

	def __init__(self, gamma=1, s=0.001, kappa=1., kernel_name="squared_exponential", nu=1.5, kernel=None, d=1,
				 grid=None, nb_dist=1024, ops=None, group=None):
		self.s = s
		self.d = d
		self.kernel_object = kernel if kernel is not None else KernelFunction(kernel_name=kernel_name, gamma=gamma, nu=nu, kappa=kappa, d=d)
		self.ops = ops if ops is not None else HipLocalOps()
		if not dist.is_initialized():
			raise RuntimeError("DistributedGaussianProcess needs torch.distributed to be initialised (one process per GPU)")
		self.world = dist.get_world_size()
		self.rank = dist.get_rank()
		self.Pr, self.Pc = grid if grid is not None else default_grid(self.world)
		if self.Pr * self.Pc != self.world:
			raise ValueError("process grid %dx%d does not match world size %d" % (self.Pr, self.Pc, self.world))
		if nb_dist % IB != 0:
			raise ValueError("nb_dist must be a multiple of %d" % IB)
		self.NB = nb_dist
		self.nb = 0
		self.myr, self.myc = self.rank // self.Pc, self.rank % self.Pc      # row-major rank -> (row, col)
		# sub-communicators: every rank creates every group, in the same order
		# (RCCL: communication kernels go to high-priority streams, i.e. hardware queues of their own, so a
		# panel broadcast is never queued behind the trailing update it is meant to overlap)
		def make_groups(kw):
			rows = [dist.new_group([r * self.Pc + c for c in range(self.Pc)], **kw) for r in range(self.Pr)]
			cols = [dist.new_group([r * self.Pc + c for r in range(self.Pr)], **kw) for c in range(self.Pc)]
			return rows, cols
		kw = {}
		if dist.get_backend() == "nccl":
			try:
				kw["pg_options"] = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
			except Exception:        # option class not exposed by this build: plain groups
				kw = {}
		try:
			self.row_groups, self.col_groups = make_groups(kw)
		except (TypeError, ValueError, RuntimeError):
			if not kw:
				raise
			# (raised identically on every rank before any communicator exists, so the retry stays collective)
			self.row_groups, self.col_groups = make_groups({})
		self.fitted = False
		self.clamp_variance = False
		self.max_size = 10000

	# ------------------------------------------------------------------ index arithmetic
	def _rank_of(self, r, c):
		return r * self.Pc + c

	def _first_local_above(self, K, my, P):
		"""index of the first local block (along one axis) whose global index exceeds K"""
		return 0 if K < my else (K - my) // P + 1

	def _count_local_below(self, K, my, P):
		"""number of local blocks (along one axis) whose global index is < K"""
		return 0 if K <= my else (K - my + P - 1) // P

	# collectives: RCCL on device tensors; under gloo (debug / single-GPU rehearsal with several
	# ranks sharing one card) device tensors are staged through the host
	def _staged(self, t):
		return t.is_cuda and dist.get_backend() == "gloo"

	def _bcast(self, t, src, group, size):
		if size <= 1:
			return
		if self._staged(t):
			h = t.cpu()
			dist.broadcast(h, src=src, group=group)
			t.copy_(h)
		else:
			dist.broadcast(t, src=src, group=group)

	def _reduce_sum(self, t, dst, group):
		if self._staged(t):
			h = t.cpu()
			dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
			t.copy_(h)
		else:
			dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)

	def _allreduce(self, t, op):
		if self._staged(t):
			h = t.cpu()
			dist.all_reduce(h, op=op)
			t.copy_(h)
		else:
			dist.all_reduce(t, op=op)

	# ------------------------------------------------------------------ fit
	def fit_gp(self, x, y):
		ops, NB, Pr, Pc, myr, myc = self.ops, self.NB, self.Pr, self.Pc, self.myr, self.myc
		xd = ops.to_device(x)
		yd = ops.to_device(y).reshape(-1)
		n = xd.shape[0]
		self.n = n
		self.x, self.y = x, y
		self._xd = xd
		nblk = (n + NB - 1) // NB
		self.nblk = nblk
		nr = (nblk - myr + Pr - 1) // Pr if nblk > myr else 0         # local block rows / cols
		nc = (nblk - myc + Pc - 1) // Pc if nblk > myc else 0
		self.nr, self.nc = nr, nc

		# ---- local Gram fill: one launch over (local row points) x (local col points)
		def global_index(nloc, my, P):
			idx = (torch.arange(nloc * NB, device=xd.device) // NB * P + my) * NB + torch.arange(nloc * NB, device=xd.device) % NB
			return idx
		gr = global_index(nr, myr, Pr)
		gc = global_index(nc, myc, Pc)
		self._gc = gc
		self._Aloc = self._zloc = None          # release the previous factor before allocating the next one
		self._winv = {}
		self.fitted = False
		Aloc = ops.empty(max(nr * NB, 1), max(nc * NB, 1))
		if nr > 0 and nc > 0:
			xr = xd[gr.clamp(max=n - 1)].contiguous()
			xc = xd[gc.clamp(max=n - 1)].contiguous()
			ops.gram(self.kernel_object, xc, xr, Aloc)
			# padding (global index >= n): identity block; noise s^2 on the global diagonal
			# (the last local index is host arithmetic: ((nloc-1)*P + my)*NB + NB-1 -- no device read-back)
			if ((nr - 1) * Pr + myr + 1) * NB > n:
				Aloc[gr >= n, :] = 0
			if ((nc - 1) * Pc + myc + 1) * NB > n:
				Aloc[:, gc >= n] = 0
			s2 = float(self.s) ** 2
			for i in range(nr):
				I = i * Pr + myr
				if I % Pc == myc:
					j = I // Pc
					dblk = Aloc[i * NB:(i + 1) * NB, j * NB:(j + 1) * NB].diagonal()
					gdiag = I * NB + torch.arange(NB, device=xd.device)
					dblk.add_(torch.where(gdiag < n, torch.full_like(dblk, s2), torch.ones_like(dblk)))
		self._Aloc = Aloc
		self._winv = {}
		bad = [torch.zeros((1,), dtype=torch.int32, device=xd.device)]

		# persistent double buffers for the panel operands (allocated once on the caller's stream, so the
		# caching allocator never recycles them while the other stream still reads them)
		prow_buf = [ops.empty(max(nr * NB, 1), NB) for _ in range(2)]
		pcol_buf = [ops.empty(max(nc * NB, 1), NB) for _ in range(2)]

		lcm = Pr * Pc // math.gcd(Pr, Pc)

		def panel_step(K, slot):
			"""Steps 1-4 of the header for block column K on the CURRENT stream: diagonal factor + its
			broadcast, local panel solve, row broadcast, column exchange.  Fills the slot's prow / pcol."""
			kr, kc, lkr, lkc = K % Pr, K % Pc, K // Pr, K // Pc
			i0 = self._first_local_above(K, myr, Pr)
			j0 = self._first_local_above(K, myc, Pc)
			rows_below, cols_right = (nr - i0) * NB, (nc - j0) * NB
			prow = prow_buf[slot][:max(rows_below, 0)]
			pcol = pcol_buf[slot][:max(cols_right, 0)]
			if myc == kc:
				winv_elems = (NB // IB) * IB * IB
				dpack = ops.empty(NB * NB + winv_elems)          # [L_KK | inverse 128-blocks of L_KK]
				if myr == kr:
					D = Aloc[lkr * NB:(lkr + 1) * NB, lkc * NB:(lkc + 1) * NB]
					winv, info = ops.potrf(D)
					bad[0] = torch.maximum(bad[0], torch.where(info > 0, info + K * NB, info))
					dpack[:NB * NB].copy_(D.reshape(-1))
					dpack[NB * NB:].copy_(winv)
				self._bcast(dpack, self._rank_of(kr, kc), self.col_groups[kc], Pr)
				Lkk = dpack[:NB * NB].reshape(NB, NB)
				wkk = dpack[NB * NB:]
				self._winv[K] = (Lkk, wkk)
				if rows_below > 0:
					panel = Aloc[i0 * NB:, lkc * NB:(lkc + 1) * NB]
					ops.trsm_right_lt(panel, Lkk, wkk)
					prow.copy_(panel)
			if rows_below > 0:
				self._bcast(prow, self._rank_of(myr, kc), self.row_groups[myr], Pc)
			# column operand: L_JK for this rank's local block columns J > K
			# The wanted J (J % Pc == myc, J % Pr == rp, J > K) form an arithmetic progression with stride
			# lcm(Pr, Pc), so source and destination are strided slices: no index tensors, no host sync.
			for rp in range(Pr):
				J0 = next((J for J in range(K + 1, min(nblk, K + 1 + lcm)) if J % Pc == myc and J % Pr == rp), None)
				if J0 is None:
					continue
				cnt = (nblk - 1 - J0) // lcm + 1
				i0p = self._first_local_above(K, rp, Pr)
				a, sa = J0 // Pr - i0p, lcm // Pr
				b, sb = J0 // Pc - j0, lcm // Pc
				dst = pcol.reshape(-1, NB, NB)[b:b + (cnt - 1) * sb + 1:sb]
				if myr == rp:
					src = prow.reshape(-1, NB, NB)[a:a + (cnt - 1) * sa + 1:sa]
					buf = src if sa == 1 else src.contiguous()
				else:
					buf = dst if sb == 1 else ops.empty(cnt * NB, NB).reshape(cnt, NB, NB)
				self._bcast(buf, self._rank_of(rp, myc), self.col_groups[myc], Pr)
				if buf is not dst:
					dst.copy_(buf)
			return i0, j0, prow, pcol

		# One block column of look-ahead: after panel K is in place, the local update of block column
		# K+1 goes first; then panel K+1 (diagonal factor, solves and ALL its broadcasts) runs on a side
		# stream while this stream applies panel K to the remaining columns.  Collectives are issued
		# in the same program order on every rank.  On CPU tensors (tests) there are no streams and the
		# same statements simply run in order.
		on_gpu = xd.is_cuda
		main = torch.cuda.current_stream() if on_gpu else None
		# (ONE side stream per object: the caching allocator keeps a pool per stream, so a fresh stream per
		# fit would turn every panel buffer of a re-fit into a hipMalloc / hipFree with its device sync)
		if on_gpu and getattr(self, "_side", None) is None:
			self._side = torch.cuda.Stream(priority=-1)
			self._ev_col, self._ev_panel = torch.cuda.Event(), torch.cuda.Event()
		side = self._side if on_gpu else None
		ev_col = self._ev_col if on_gpu else None
		ev_panel = self._ev_panel if on_gpu else None

		cur = panel_step(0, 0)
		for K in range(nblk):
			i0, j0, prow, pcol = cur
			if K + 1 >= nblk:
				break
			rows_below = prow.shape[0]
			nxt_c = (K + 1) % Pc
			j1 = self._first_local_above(K + 1, myc, Pc)             # first local block column beyond K+1
			if myc == nxt_c and rows_below > 0:                      # this rank owns (part of) block column K+1
				jc = (K + 1) // Pc
				ops.gemm_nt(prow, pcol[(jc - j0) * NB:(jc - j0 + 1) * NB], Aloc[i0 * NB:, jc * NB:(jc + 1) * NB], 1,
							bc=(NB, Pr, Pc, myr, myc, i0, jc))
			if on_gpu:
				ev_col.record(main)
			# the remaining columns: compute only, enqueued before the (possibly host-blocking) collectives below
			if rows_below > 0 and nc - j1 > 0:
				ops.gemm_nt(prow, pcol[(j1 - j0) * NB:], Aloc[i0 * NB:, j1 * NB:], 1, bc=(NB, Pr, Pc, myr, myc, i0, j1))
			if on_gpu:
				with torch.cuda.stream(side):
					side.wait_event(ev_col)
					ops.beside_update(True)
					try:
						cur = panel_step(K + 1, (K + 1) % 2)
					finally:
						ops.beside_update(False)
					ev_panel.record(side)
				main.wait_event(ev_panel)
			else:
				cur = panel_step(K + 1, (K + 1) % 2)
		bad = bad[0]

		self._allreduce(bad, dist.ReduceOp.MAX)
		if int(bad.item()) != 0:
			raise torch.linalg.LinAlgError("distributed potrf: the leading minor of order %d is not positive definite" % int(bad.item()))
		# z = L^-1 y through the same left-looking solve with one right-hand side
		ypad = ops.zeros(1, nblk * NB)
		ypad[0, :n] = yd
		self._zloc = self._solve_rows(ypad, None)
		self.fitted = True
		return None

	fit = fit_gp

	# ------------------------------------------------------------------ X = B L^-T, left-looking, column blocks distributed
	def _solve_rows(self, rhs_full, xtest):
		"""
		rhs rows against L.  Either ``rhs_full`` (m x Npad, replicated; used for y) or ``xtest``
		(m x d): then block K of the right-hand side, k(x_K, xtest), is formed by the diagonal owner.
		Returns this rank's column blocks of X: (m, nc*NB).
		"""
		ops, NB, Pr, Pc, myr, myc = self.ops, self.NB, self.Pr, self.Pc, self.myr, self.myc
		Aloc, n = self._Aloc, self.n
		m = rhs_full.shape[0] if rhs_full is not None else xtest.shape[0]
		Xloc = ops.zeros(m, max(self.nc * NB, 1))
		for K in range(self.nblk):
			kr, kc, lkr, lkc = K % Pr, K % Pc, K // Pr, K // Pc
			if myr == kr:
				jc = self._count_local_below(K, myc, Pc)
				S = ops.zeros(m, NB)
				if jc > 0:
					ops.gemm_nt(Xloc[:, :jc * NB], Aloc[lkr * NB:(lkr + 1) * NB, :jc * NB], S, 0)
				if Pc > 1:
					self._reduce_sum(S, self._rank_of(kr, kc), self.row_groups[kr])
			if myc == kc:
				XK = ops.empty(m, NB)
				if myr == kr:
					if rhs_full is not None:
						XK.copy_(rhs_full[:, K * NB:(K + 1) * NB])
					else:
						gk = (K * NB + torch.arange(NB, device=Xloc.device))
						xk = self._xd[gk.clamp(max=n - 1)].contiguous()
						ops.gram(self.kernel_object, xk, xtest, XK)          # XK[t, i] = k(xtest_t, x_{K,i})
						if (K + 1) * NB > n:
							XK[:, gk >= n] = 0
					XK.sub_(S)
					Lkk, wkk = self._winv[K]
					ops.trsm_right_lt(XK, Lkk, wkk)
				self._bcast(XK, self._rank_of(kr, kc), self.col_groups[kc], Pr)
				Xloc[:, lkc * NB:(lkc + 1) * NB] = XK
		return Xloc

	# ------------------------------------------------------------------ predict
	def mean_std(self, xtest, full=False, reuse=False):
		if full:
			raise NotImplementedError("full covariance is not provided on the distributed path")
		if not self.fitted:
			raise RuntimeError("fit_gp first")
		ops = self.ops
		xt = ops.to_device(xtest)
		Xloc = self._solve_rows(None, xt)
		red = ops.empty(2, xt.shape[0])
		ops.row_sums(Xloc[:, :self.nc * self.NB], self._zloc.reshape(-1), out=red)
		self._allreduce(red, dist.ReduceOp.SUM)
		kd = ops.kdiag(self.kernel_object, xt)
		# every process row holds a replica of its columns: scale = 1 / P_r; mu in place, sigma = sqrt(kd - scale * sumsq)
		mu, sigma = ops.predict_finish(red[0], red[1], kd, 1.0 / self.Pr, self.clamp_variance)
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(sigma.reshape(-1, 1), xtest))

	mean_var = mean_std

	def log_marginal(self, kernel=None, X=None, weight=1.0):
		"""1/2 z^T z + 1/2 * weight * 2 sum log L_ii for the fitted hyper-parameters (estimator.py:32-40)."""
		if X:
			raise NotImplementedError("hyper-parameter overrides are not provided on the distributed path")
		ops = self.ops
		acc = ops.zeros(2)
		for K in range(self.nblk):
			if K % self.Pr == self.myr and K % self.Pc == self.myc:
				Lkk, _ = self._winv[K]
				acc[0] += ops.logdet(Lkk)
		if self.myr == 0 and self.nc > 0:
			z = self._zloc[:, :self.nc * self.NB]
			acc[1] = ops.row_sums(z, z.reshape(-1))[1][0]
		self._allreduce(acc, dist.ReduceOp.SUM)
		val = 0.5 * acc[1] + 0.5 * float(weight) * 2.0 * acc[0]
		return _lib.like_input(val.reshape(1, 1), self.x)
