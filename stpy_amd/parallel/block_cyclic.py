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
  X_K = (K*_K - sum_{J<K} X_J L_KJ^T) L_KK^-T :  every rank of process row K%P_r forms the (negative)
  partial sum over ITS local block columns J < K (the operands are contiguous in local storage), the
  partials are summed onto the diagonal owner (reduce along the process row), the owner adds K*_K,
  finishes the block and broadcasts it down its process column.  One block of look-ahead: the part of
  block K+1's partial sum that does not involve X_K (all J <= K-1) is formed on the side stream while
  block K's reduce / solve / broadcast chain runs; only the single J = K term (one NB-deep product)
  sits between two chains.  mu = X z and sigma^2 = kdiag - rowsum(X o X) are all-reduced partial sums
  finished by ``stpy_predict_finish``; z = L^-1 y is the same solve with a single right-hand side.

All arithmetic goes through a ``LocalOps`` object.  The product backend is ``HipLocalOps`` (the C
ABI of libstpy_hip; raises without a GPU).  Tests inject a CPU backend from ``tests/`` to exercise
the index arithmetic and the collectives on gloo; nothing here imports the oracle.

Look-ahead: one block column.  After panel K is in place the local update of block column K+1 runs
first; panel K+1 (diagonal factor, panel solve and all three broadcasts) is then issued on a
high-priority side stream while the main stream applies panel K to the remaining columns.

Status: exercised on gloo with 2 ... 8 CPU ranks (CPU stand-in for the tile arithmetic, against the oracle) and
with 1 / 2 / 4 ranks sharing one MI355X (real HIP kernels, collectives staged through pinned host memory,
against the oracle and the single-GPU class).  The RCCL transport between GPUs has only run at world size 1
(process group, high-priority sub-communicators, both streams): the authoring loop has a single GPU and RCCL
refuses two ranks on one device -- ``DistributedGaussianProcess`` therefore warns once when it is constructed
on the nccl backend with more than one rank.  DESIGN.md section (f) holds the per-step cost model a measured
1 / 2 / 4 / 8-GPU curve is to be judged against.
"""
import contextlib
import warnings
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
		self._skwork = {}

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

	def gram(self, kernel_object, xa, xb, out, kwargs=None, add=False):
		"""out[j, i] (+)= k(xb_j, xa_i); out may be a strided 2-D view.  ``add``: the value is added to what ``out`` holds."""
		if add:
			items = kernel_object._resolve(dict(kwargs) if kwargs else {})
			if len(items) == 1:
				kernel_object._run_items([dict(items[0], op="+")], xa, xb, out, first_is_set=False)
				return
			tmp = torch.empty_like(out)
			kernel_object._run_items(items, xa, xb, tmp)
			self.add_into(out, tmp)
			return
		kernel_object._kernel_into(xa, xb, out, kwargs)

	def add_into(self, out, src):
		"""out += src on (strided) 2-D views."""
		_lib.check(self.lib.stpy_combine(self.code, out.shape[0], out.shape[1], _lib.ptr(out), out.stride(0), _lib.ptr(src), src.stride(0),
										 _lib.OUT_ADD, 0.0, _lib.stream_ptr()), "stpy_combine")

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
				# one workspace per stream, grown on demand and kept: no allocation inside the per-block loop
				key = torch.cuda.current_stream().cuda_stream
				work = self._skwork.get(key)
				if work is None or work.numel() < passes * m * n:
					work = self._skwork[key] = self.empty(passes * m * n)
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

	def predict_finish(self, mu, sumsq, kdiag, scale, clamp, want_sigma=True):
		"""mu *= scale (in place); sigma = sqrt(kdiag - scale * sumsq): the epilogue after the all-reduce of the partial sums."""
		sigma = self.empty(mu.shape[0]) if want_sigma else None
		_lib.check(self.lib.stpy_predict_finish(self.code, mu.shape[0], _lib.ptr(mu), _lib.ptr(sumsq), _lib.ptr(kdiag), float(scale), _lib.ptr(sigma),
												1 if clamp else 0, _lib.stream_ptr()), "stpy_predict_finish")
		return mu, sigma

	def logdet(self, L):
		"""sum_i log L_ii of the (strided) factor block L."""
		out2 = self.empty(2)
		_lib.check(self.lib.stpy_logdet_quad(self.code, L.shape[0], _lib.ptr(L), L.stride(0), None, _lib.ptr(out2), _lib.stream_ptr()), "stpy_logdet_quad")
		return out2[0]


class _Factor:
	"""The distributed factor of one (x, hyper-parameter) pair: this rank's local blocks of L, the inverse diagonal
	blocks the solves reuse, and z = L^-1 y in the column distribution."""
	__slots__ = ("n", "nblk", "nr", "nc", "Aloc", "winv", "zloc", "xd", "kwargs", "NB")


class DistributedGaussianProcess:
	"""``GaussianProcess`` on a P_r x P_c process grid: same constructor arguments for the path, ``add_data_point`` /
	``fit`` / ``fit_gp`` / ``mean_std`` (``full=True``, ``max_size`` chunking) / ``mean_var`` / ``log_marginal(kernel, X,
	weight)`` with the kwargs-override protocol; results replicated on every rank.  With one rank nothing is distributed:
	the object then simply holds a single-GPU ``GaussianProcess`` (``force_path=True`` keeps the block-cyclic code path,
	which is how its own overhead is measured)."""

	def __init__(self, gamma=1, s=0.001, kappa=1., kernel_name="squared_exponential", nu=1.5, kernel=None, d=1,
				 grid=None, nb_dist=None, ops=None, group=None, force_path=False, transport="collective", col_exchange="allgather", audit=False):
		self.s = s
		self.d = d
		self.kernel_object = kernel if kernel is not None else KernelFunction(kernel_name=kernel_name, gamma=gamma, nu=nu, kappa=kappa, d=d)
		self.kernel = self.kernel_object.kernel
		if not dist.is_initialized():
			raise RuntimeError("DistributedGaussianProcess needs torch.distributed to be initialised (one process per GPU)")
		self.world = dist.get_world_size()
		self.rank = dist.get_rank()
		self.Pr, self.Pc = grid if grid is not None else default_grid(self.world)
		if self.Pr * self.Pc != self.world:
			raise ValueError("process grid %dx%d does not match world size %d" % (self.Pr, self.Pc, self.world))
		# distribution block: None = by the problem size at fit time (auto_nb_dist: 2048 from N = 131 072 on -- half the block steps,
		# i.e. half the latency-bound panel chains and collectives, and K = 2048 trailing updates -- 1024 below)
		if nb_dist is not None and (nb_dist <= 0 or nb_dist % IB != 0):
			raise ValueError("nb_dist must be a positive multiple of %d" % IB)
		self._nb_dist_arg = nb_dist
		self.NB = nb_dist if nb_dist is not None else 1024
		if transport not in ("auto", "collective", "fanout"):
			raise ValueError("transport: 'auto', 'collective' (RCCL broadcast) or 'fanout' (root -> every peer point to point)")
		if col_exchange not in ("allgather", "bcast"):
			raise ValueError("col_exchange: 'allgather' (one collective per step) or 'bcast' (one broadcast per process row)")
		self._transport_arg, self.col_exchange = transport, col_exchange
		self.transport = "collective"
		self.audit = [] if audit else None          # per collective: (communicator, op, root, bytes, logical stream) -- tests/test_block_cyclic_cpu.py
		self._lstream = "main"
		self.nb = 0
		self.fitted = False
		self.clamp_variance = False
		self.max_size = 10000               # gauss_procc.py:55: prediction chunk
		self.x = self.y = None
		self.n = 0
		self._f = None
		self.stats = {"bcast_bytes": 0, "reduce_bytes": 0, "collectives": 0}
		# one rank and nothing injected: there is nothing to distribute -- the single-GPU estimator IS the product path
		self._single = None
		if self.world == 1 and ops is None and not force_path:
			from ..continuous_processes.gauss_procc import GaussianProcess
			self._single = GaussianProcess(s=s, kernel=self.kernel_object, d=d)
			return
		self.ops = ops if ops is not None else HipLocalOps()
		self.myr, self.myc = self.rank // self.Pc, self.rank % self.Pc      # row-major rank -> (row, col)
		if dist.get_backend() == "nccl" and self.world > 1:
			warnings.warn("DistributedGaussianProcess: the RCCL transport between GPUs has not been exercised by this build's tests "
						  "(schedule verified on gloo and on one GPU with staged collectives) -- compare against the single-GPU "
						  "class on a small problem before relying on it", RuntimeWarning, stacklevel=2)
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
		self._group_names = {}
		for r, g in enumerate(self.row_groups):
			self._group_names[id(g)] = "row%d" % r
		for c, g in enumerate(self.col_groups):
			self._group_names[id(g)] = "col%d" % c
		# RCCL creates a communicator at the FIRST collective of a group.  Do that here, in one fixed order (own process row, then
		# own process column: the row groups are disjoint, so are the column groups -- no cyclic wait), instead of in the middle of
		# the first panel step where ranks reach their groups at different points of the schedule.
		# Transport of the panel broadcasts.  xGMI is point to point (every GPU pair has its own link), so a root that sends its
		# panel to each peer separately -- one batched group of isend / irecv, which RCCL runs concurrently -- keeps 1 ... P-1 links
		# busy at once where a ring broadcast is bound by one link per hop.  "auto" takes the fan-out only on RCCL and only when
		# every visible device pair has peer access.  The DEFAULT is the plain RCCL broadcast ("collective"): neither form has
		# run between real GPUs in the authoring loop, and the first contact should be with the most ordinary collective; the
		# fan-out is an option to measure against it (bench.py --transport fanout / STPY_DIST_TRANSPORT=fanout).
		if self._transport_arg == "fanout":
			self.transport = "fanout"
		elif self._transport_arg == "auto" and dist.get_backend() == "nccl" and self.world > 1 and self._peer_access_everywhere():
			self.transport = "fanout"
		if dist.get_backend() == "nccl" and self.world > 1:
			token = torch.zeros(1, dtype=torch.float32, device=self.ops.device)
			if self.Pc > 1:
				dist.all_reduce(token, group=self.row_groups[self.myr])
			if self.Pr > 1:
				dist.all_reduce(token, group=self.col_groups[self.myc])
			dist.all_reduce(token)
			torch.cuda.synchronize()

	@staticmethod
	def auto_nb_dist(n):
		"""Distribution block when the caller gave none: 2048 from N = 131 072 on, 1024 below."""
		return 2048 if n >= 131072 else 1024

	@staticmethod
	def _peer_access_everywhere():
		try:
			nd = torch.cuda.device_count()
			return nd > 1 and all(torch.cuda.can_device_access_peer(a, b) for a in range(nd) for b in range(nd) if a != b)
		except Exception:          # noqa: BLE001
			return False

	@contextlib.contextmanager
	def _on(self, name, stream):
		"""Everything issued inside runs on `stream` (None on CPU tensors) and is logged under the logical stream `name`."""
		prev, self._lstream = self._lstream, name
		try:
			if stream is not None:
				with torch.cuda.stream(stream):
					yield
			else:
				yield
		finally:
			self._lstream = prev

	# ------------------------------------------------------------------ index arithmetic
	def _rank_of(self, r, c):
		return r * self.Pc + c

	def _first_local_above(self, K, my, P):
		"""index of the first local block (along one axis) whose global index exceeds K"""
		return 0 if K < my else (K - my) // P + 1

	def _count_local_below(self, K, my, P):
		"""number of local blocks (along one axis) whose global index is < K"""
		return 0 if K <= my else (K - my + P - 1) // P

	# collectives: RCCL on device tensors.  Under gloo (debug / single-GPU rehearsal with several ranks sharing one
	# card) device tensors are staged through PINNED host memory with an event on the issuing stream: only that
	# stream is waited for, so the ordering between the main and the side stream is exercised as on RCCL.
	def _staged(self, t):
		return t.is_cuda and dist.get_backend() == "gloo"

	def _to_host(self, t):
		h = torch.empty(t.shape, dtype=t.dtype, device="cpu", pin_memory=True)
		h.copy_(t, non_blocking=True)
		ev = torch.cuda.Event()
		ev.record(torch.cuda.current_stream())
		ev.synchronize()
		return h

	def _from_host(self, t, h):
		t.copy_(h, non_blocking=True)
		self._pinned_keep.append(h)          # pinned source of an asynchronous copy: kept until the end of the fit / predict call

	_pinned_keep = []

	def _count(self, t, kind):
		self.stats[kind] += t.numel() * t.element_size()
		self.stats["collectives"] += 1

	def _log(self, group, op, root, t):
		"""Order audit: what an RCCL communicator needs to see identically on every member -- the sequence of (operation, root,
		bytes) -- plus the logical stream it was issued from.  A mismatch passes on host-staged gloo and hangs on RCCL."""
		if self.audit is not None:
			name = "world" if group is None else self._group_names[id(group)]
			self.audit.append((name, op, int(root), int(t.numel() * t.element_size()), self._lstream))

	def _members(self, group):
		return list(range(self.world)) if group is None else dist.get_process_group_ranks(group)

	def _fanout(self, buf, src, group):
		"""root -> every other member, point to point, as ONE batched group (RCCL: concurrent sends on distinct links)."""
		me = dist.get_rank()
		if me == src:
			ops = [dist.P2POp(dist.isend, buf, peer, group) for peer in self._members(group) if peer != src]
		else:
			ops = [dist.P2POp(dist.irecv, buf, src, group)]
		for w in dist.batch_isend_irecv(ops):
			w.wait()

	def _bcast(self, t, src, group, size):
		if size <= 1:
			return
		self._count(t, "bcast_bytes")
		self._log(group, "bcast/" + self.transport, src, t)
		send = self._fanout if self.transport == "fanout" else (lambda b, s_, g: dist.broadcast(b, src=s_, group=g))
		if self._staged(t):
			h = self._to_host(t)
			send(h, src, group)
			if dist.get_rank() != src:
				self._from_host(t, h)
		else:
			send(t, src, group)

	def _allgather(self, out, mine, group, size):
		"""out[(member index)] <- every member's `mine` (equal sizes); one collective."""
		self._count(out, "bcast_bytes")
		self._log(group, "allgather", -1, mine)
		if self._staged(mine):
			hm, ho = self._to_host(mine), torch.empty(out.shape, dtype=out.dtype, device="cpu", pin_memory=True)
			dist.all_gather_into_tensor(ho.reshape(-1), hm.reshape(-1), group=group)
			self._from_host(out, ho)
		else:
			dist.all_gather_into_tensor(out.reshape(-1), mine.reshape(-1), group=group)

	def _reduce_sum(self, t, dst, group, size):
		if size <= 1:
			return
		self._count(t, "reduce_bytes")
		self._log(group, "reduce", dst, t)
		if self._staged(t):
			h = self._to_host(t)
			dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
			if dist.get_rank() == dst:
				self._from_host(t, h)
		else:
			dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)

	def _allreduce(self, t, op):
		if self.world <= 1:
			return
		self._count(t, "reduce_bytes")
		self._log(None, "allreduce", -1, t)
		if self._staged(t):
			h = self._to_host(t)
			dist.all_reduce(h, op=op)
			self._from_host(t, h)
		else:
			dist.all_reduce(t, op=op)

	def _streams(self, on_gpu, bulk=False):
		"""(main, second stream, two events).  The factorisation's panel look-ahead runs on a HIGH-priority side stream (its
		small kernels must get CU slots among the trailing update's workgroups); the solve's look-ahead is the bulk of the
		work and runs on a normal-priority stream beside the latency-bound chain on the caller's stream.  ONE stream of
		each kind per object -- the caching allocator keeps a pool per stream, so a fresh stream per fit would turn every
		panel buffer of a re-fit into a hipMalloc / hipFree with its device sync."""
		if not on_gpu:
			return None, None, None, None
		if getattr(self, "_side", None) is None:
			self._side = torch.cuda.Stream(priority=-1)
			self._bulk = torch.cuda.Stream()
			self._ev = [torch.cuda.Event() for _ in range(4)]
		if bulk:
			return torch.cuda.current_stream(), self._bulk, self._ev[2], self._ev[3]
		return torch.cuda.current_stream(), self._side, self._ev[0], self._ev[1]

	# ------------------------------------------------------------------ small API mirrors (gauss_procc.py:100-134)
	def add_data_point(self, x, y, Sigma=None):
		"""gauss_procc.py:100-111: concatenate and refit from scratch (every rank holds x, y)."""
		if Sigma is not None:
			raise NotImplementedError("a general noise matrix is not provided on the distributed path")
		if self.x is not None:
			x = torch.cat((self.x, x), dim=0)
			y = torch.cat((self.y, y), dim=0)
		self.fit_gp(x, y)

	add_data = add_data_point

	def fit(self, x=None, y=None):
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

	@property
	def A(self):
		"""K^-1 y is not assembled on the distributed path (prediction needs only z = L^-1 y)."""
		if self._single is not None:
			return self._single.A
		raise NotImplementedError("alpha = K^-1 y is not assembled on the distributed path")

	# ------------------------------------------------------------------ fit
	def fit_gp(self, x, y, Sigma=None, iterative=False, extrapoint=False):
		if Sigma is not None:
			raise NotImplementedError("a general noise matrix is not provided on the distributed path")
		self.x, self.y = x, y
		self.n, self.d = x.shape[0], x.shape[1]
		if self._single is not None:
			self._single.fit_gp(x, y)
			self.fitted = True
			return None
		self.fitted = False
		self._f = None                      # release the previous factor before allocating the next one
		xd = self.ops.to_device(x)
		yd = self.ops.to_device(y).reshape(-1)
		self._f = self._factorize(xd, yd, None)
		self._factor_key = self._hyper_key()
		self.fitted = True
		return None

	def _hyper_key(self):
		from ..continuous_processes.gauss_procc import GaussianProcess
		return GaussianProcess._hyper_key(self, self.kernel_object)

	def _factorize(self, xd, yd, kwargs):
		"""Gram fill + block-cyclic Cholesky + z = L^-1 y for the kernel parameters in ``kwargs`` (None: the stored ones)."""
		n = xd.shape[0]
		if self._nb_dist_arg is None:
			self.NB = self.auto_nb_dist(n)          # (a function of N alone: the same on every rank)
		ops, NB, Pr, Pc, myr, myc = self.ops, self.NB, self.Pr, self.Pc, self.myr, self.myc
		f = _Factor()
		f.n, f.xd, f.kwargs = n, xd, kwargs
		f.NB = NB
		nblk = (n + NB - 1) // NB
		f.nblk = nblk
		nr = (nblk - myr + Pr - 1) // Pr if nblk > myr else 0         # local block rows / cols
		nc = (nblk - myc + Pc - 1) // Pc if nblk > myc else 0
		f.nr, f.nc = nr, nc
		del self._pinned_keep[:]

		# ---- local Gram fill, lower blocks only: local block row i (global I) needs the local block columns j with
		# ---- global J = j*Pc + myc <= I -- a staircase, one launch per local block row (nothing above the diagonal is
		# ---- ever read: the trailing update skips those tiles and the panels start at the diagonal)
		def global_index(nloc, my, P):
			idx = (torch.arange(nloc * NB, device=xd.device) // NB * P + my) * NB + torch.arange(nloc * NB, device=xd.device) % NB
			return idx
		gr = global_index(nr, myr, Pr)
		gc = global_index(nc, myc, Pc)
		Aloc = ops.empty(max(nr * NB, 1), max(nc * NB, 1))
		if nr > 0 and nc > 0:
			xr = xd[gr.clamp(max=n - 1)].contiguous()
			xc = xd[gc.clamp(max=n - 1)].contiguous()
			s2 = float(self.s) ** 2
			for i in range(nr):
				I = i * Pr + myr
				jn = self._count_local_below(I + 1, myc, Pc)            # local block columns with J <= I
				if jn == 0:
					continue
				ops.gram(self.kernel_object, xc[:jn * NB], xr[i * NB:(i + 1) * NB], Aloc[i * NB:(i + 1) * NB, :jn * NB], kwargs)
			# padding (global index >= n): identity block; noise s^2 on the global diagonal
			# (the last local index is host arithmetic: ((nloc-1)*P + my)*NB + NB-1 -- no device read-back)
			if ((nr - 1) * Pr + myr + 1) * NB > n:
				Aloc[gr >= n, :] = 0
			if ((nc - 1) * Pc + myc + 1) * NB > n:
				Aloc[:, gc >= n] = 0
			for i in range(nr):
				I = i * Pr + myr
				if I % Pc == myc:
					j = I // Pc
					dblk = Aloc[i * NB:(i + 1) * NB, j * NB:(j + 1) * NB].diagonal()
					gdiag = I * NB + torch.arange(NB, device=xd.device)
					dblk.add_(torch.where(gdiag < n, torch.full_like(dblk, s2), torch.ones_like(dblk)))
		f.Aloc = Aloc
		f.winv = {}
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
				f.winv[K] = (Lkk, wkk)
				if rows_below > 0:
					panel = Aloc[i0 * NB:, lkc * NB:(lkc + 1) * NB]
					ops.trsm_right_lt(panel, Lkk, wkk)
					prow.copy_(panel)
			if rows_below > 0:
				self._bcast(prow, self._rank_of(myr, kc), self.row_groups[myr], Pc)
			# column operand: L_JK for this rank's local block columns J > K
			# The wanted J (J % Pc == myc, J % Pr == rp, J > K) form an arithmetic progression with stride
			# lcm(Pr, Pc), so source and destination are strided slices: no index tensors, no host sync.
			prog = []
			for rp in range(Pr):
				J0 = next((J for J in range(K + 1, min(nblk, K + 1 + lcm)) if J % Pc == myc and J % Pr == rp), None)
				prog.append(None if J0 is None else (J0, (nblk - 1 - J0) // lcm + 1))
			def src_of(rp):          # (this rank's contribution: blocks of prow)
				J0, cnt = prog[rp]
				a, sa = J0 // Pr - self._first_local_above(K, rp, Pr), lcm // Pr
				return prow.reshape(-1, NB, NB)[a:a + (cnt - 1) * sa + 1:sa]
			def dst_of(rp):
				J0, cnt = prog[rp]
				b, sb = J0 // Pc - j0, lcm // Pc
				return pcol.reshape(-1, NB, NB)[b:b + (cnt - 1) * sb + 1:sb]
			if Pr == 1:
				if prog[0] is not None:
					dst_of(0).copy_(src_of(0))
			elif self.col_exchange == "allgather":
				# ONE collective per step and column group: every member contributes the blocks it holds (padded to the longest
				# progression) and receives everybody's -- on two process rows a simultaneous exchange in both directions of the
				# link instead of two one-way broadcasts in sequence
				cmax = max((pg[1] for pg in prog if pg is not None), default=0)
				if cmax > 0:
					mine = ops.empty(cmax, NB, NB)
					if prog[myr] is not None:
						mine[:prog[myr][1]].copy_(src_of(myr))
					allb = ops.empty(Pr, cmax, NB, NB)
					self._allgather(allb, mine, self.col_groups[myc], Pr)
					for rp in range(Pr):
						if prog[rp] is not None:
							dst_of(rp).copy_(allb[rp, :prog[rp][1]])
			else:
				for rp in range(Pr):
					if prog[rp] is None:
						continue
					cnt = prog[rp][1]
					dst = dst_of(rp)
					if myr == rp:
						src = src_of(rp)
						buf = src if src.is_contiguous() else src.contiguous()
					else:
						buf = dst if dst.is_contiguous() else ops.empty(cnt * NB, NB).reshape(cnt, NB, NB)
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
		main, side, ev_col, ev_panel = self._streams(on_gpu)

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
			with self._on("side", side):
				if on_gpu:
					side.wait_event(ev_col)
				ops.beside_update(True)
				try:
					cur = panel_step(K + 1, (K + 1) % 2)
				finally:
					ops.beside_update(False)
				if on_gpu:
					ev_panel.record(side)
			if on_gpu:
				main.wait_event(ev_panel)
		bad = bad[0]

		self._allreduce(bad, dist.ReduceOp.MAX)
		if int(bad.item()) != 0:
			raise torch.linalg.LinAlgError("distributed potrf: the leading minor of order %d is not positive definite" % int(bad.item()))
		# z = L^-1 y through the same left-looking solve with one right-hand side
		ypad = ops.zeros(1, nblk * NB)
		ypad[0, :n] = yd
		f.zloc = self._solve_rows(f, ypad, None)
		del self._pinned_keep[:]
		return f

	# ------------------------------------------------------------------ X = B L^-T, left-looking, column blocks distributed
	def _solve_rows(self, f, rhs_full, xtest):
		"""
		rhs rows against L.  Either ``rhs_full`` (m x Npad, replicated; used for y) or ``xtest``
		(m x d): then block K of the right-hand side, k(x_K, xtest), is formed by the diagonal owner.
		Returns this rank's column blocks of X: (m, nc*NB).

		Per block K, on process row K % P_r:  S = - sum_{local J < K} X_J L_KJ^T  (accumulated by the subtracting GEMM on
		a zeroed buffer), reduce onto the diagonal owner, owner:  X_K = (S + rhs_K) L_KK^-T, broadcast down its process
		column.  Look-ahead: the terms J <= K-1 of block K+1's sum are formed on the side stream while block K's reduce /
		solve / broadcast chain runs; only the J = K term is added between the chains.  S lives in two preallocated slots.
		"""
		ops, NB, Pr, Pc, myr, myc = self.ops, f.NB, self.Pr, self.Pc, self.myr, self.myc
		Aloc, n, nblk = f.Aloc, f.n, f.nblk
		m = rhs_full.shape[0] if rhs_full is not None else xtest.shape[0]
		Xloc = ops.zeros(m, max(f.nc * NB, 1))
		on_gpu = Xloc.is_cuda
		main, side, ev_a, ev_b = self._streams(on_gpu, bulk=True)
		S = [ops.empty(m, NB), ops.empty(m, NB)]
		XKbuf = [ops.empty(m, NB), ops.empty(m, NB)]          # two slots: block K's broadcast may still be in flight when K+1 is formed

		def partial(K, slot, j_lo, j_hi, first):
			"""S[slot] (-)= X[:, local blocks j_lo..j_hi) L_K,.^T on the current stream (process row K % Pr only)."""
			if first:
				S[slot].zero_()
			if j_hi > j_lo:
				lkr = K // Pr
				ops.gemm_nt(Xloc[:, j_lo * NB:j_hi * NB], Aloc[lkr * NB:(lkr + 1) * NB, j_lo * NB:j_hi * NB], S[slot], 1)

		# prologue: block 0 has no predecessors
		if myr == 0 % Pr:
			partial(0, 0, 0, 0, True)
		for K in range(nblk):
			kr, kc, lkr, lkc = K % Pr, K % Pc, K // Pr, K // Pc
			slot = K % 2
			# look-ahead for block K+1 (terms J <= K-1) on the second stream, behind everything `main` has issued up to the
			# end of step K-1: it reads Xloc blocks < K only (final by then) and writes the OTHER S slot
			if K + 1 < nblk and myr == (K + 1) % Pr:
				jn = self._count_local_below(K, myc, Pc)             # local blocks with J <= K-1
				if on_gpu:
					ev_a.record(main)
				with self._on("bulk", side):
					if on_gpu:
						side.wait_event(ev_a)
					partial(K + 1, (K + 1) % 2, 0, jn, True)
					if on_gpu:
						ev_b.record(side)
			if myr == kr:
				# the one term the look-ahead could not cover: J = K-1 (if this rank holds that block column)
				if K >= 1 and (K - 1) % Pc == myc:
					jl = (K - 1) // Pc
					partial(K, slot, jl, jl + 1, False)
				self._reduce_sum(S[slot], self._rank_of(kr, kc), self.row_groups[kr], Pc)
			if myc == kc:
				XK = XKbuf[slot]
				if myr == kr:
					if rhs_full is not None:
						ops.add_into(S[slot], rhs_full[:, K * NB:(K + 1) * NB])
					else:
						gk = (K * NB + torch.arange(NB, device=Xloc.device))
						xk = f.xd[gk.clamp(max=n - 1)].contiguous()
						if (K + 1) * NB > n:
							# padded columns (global index >= n) carry no kernel value: form the block apart and blank them
							ops.gram(self.kernel_object, xk, xtest, XK, f.kwargs)
							XK[:, gk >= n] = 0
							ops.add_into(S[slot], XK)
						else:
							ops.gram(self.kernel_object, xk, xtest, S[slot], f.kwargs, add=True)      # S += k(xtest, x_K)
					Lkk, wkk = f.winv[K]
					ops.trsm_right_lt(S[slot], Lkk, wkk)
					XK.copy_(S[slot])
				self._bcast(XK, self._rank_of(kr, kc), self.col_groups[kc], Pr)
				Xloc[:, lkc * NB:(lkc + 1) * NB] = XK
			if on_gpu and K + 1 < nblk and myr == (K + 1) % Pr:
				main.wait_event(ev_b)                                # S[(K+1) % 2] holds the look-ahead part
		return Xloc

	# ------------------------------------------------------------------ predict
	def mean_std(self, xtest, full=False, reuse=False):
		"""gauss_procc.py:310-334: chunks of ``max_size`` test points against the resident factor."""
		if self._single is not None:
			self._single.max_size, self._single.clamp_variance = self.max_size, self.clamp_variance
			return self._single.mean_std(xtest, full=full, reuse=reuse)
		if not self.fitted:
			raise RuntimeError("fit_gp first")
		m = xtest.shape[0]
		if m < self.max_size or full:
			return self._mean_std_sub(xtest, full)
		mus, sds = [], []
		for i0 in range(0, m, self.max_size):
			mu, sd = self._mean_std_sub(xtest[i0:i0 + self.max_size], False)
			mus.append(mu)
			sds.append(sd)
		return torch.cat(mus), torch.cat(sds)

	mean_var = mean_std

	def _mean_std_sub(self, xtest, full):
		ops, f = self.ops, self._f
		xt = ops.to_device(xtest)
		m = xt.shape[0]
		Xloc = self._solve_rows(f, None, xt)
		Xl = Xloc[:, :f.nc * f.NB]
		red = ops.empty(2, m)
		ops.row_sums(Xl, f.zloc.reshape(-1), out=red)
		self._allreduce(red, dist.ReduceOp.SUM)
		# every process row holds a replica of its columns: scale = 1 / P_r
		if not full:
			kd = ops.kdiag(self.kernel_object, xt)
			mu, sigma = ops.predict_finish(red[0], red[1], kd, 1.0 / self.Pr, self.clamp_variance)
			del self._pinned_keep[:]
			return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(sigma.reshape(-1, 1), xtest))
		# full covariance K** - X X^T (gauss_procc.py:396-399): rank (0, 0) starts from K**, every rank of process row 0
		# subtracts the product over ITS columns of X, the others contribute zeros; one all-reduce assembles the sum
		cov = ops.zeros(m, m)
		if self.myr == 0 and self.myc == 0:
			ops.gram(self.kernel_object, xt, xt, cov)
		if self.myr == 0 and f.nc > 0:
			ops.gemm_nt(Xl, Xl, cov, 1)
		self._allreduce(cov, dist.ReduceOp.SUM)
		mu, _ = ops.predict_finish(red[0], None, None, 1.0 / self.Pr, False, want_sigma=False)
		del self._pinned_keep[:]
		return (_lib.like_input(mu.reshape(-1, 1), xtest), _lib.like_input(cov, xtest))

	def mean(self, xtest):
		return self.mean_std(xtest)[0]

	def log_marginal(self, kernel=None, X=None, weight=1.0):
		"""gauss_procc.py:497-504 -> :631-638 (== estimator.py:32-40): 1/2 y^T K^-1 y + 1/2 * weight * log det K, shape (1, 1).
		``X``: per-item parameter overrides in the kwargs protocol of kernels.py:138-157 -- the matrix is then refilled and
		refactored with them (as the reference does on every call); with X empty, ``kernel`` the fitted kernel object and
		unchanged hyper-parameters the resident factor is reused."""
		kernel = self.kernel_object if kernel is None else kernel
		if self._single is not None:
			return self._single.log_marginal(kernel, X if X else {}, weight)
		ops = self.ops
		reuse = self.fitted and not X and kernel is self.kernel_object and getattr(self, "_factor_key", None) == self._hyper_key()
		if reuse:
			f = self._f
		else:
			if self.x is None:
				raise AttributeError("log_marginal needs data: call fit_gp first")
			saved = self.kernel_object
			self.kernel_object = kernel
			try:
				f = self._factorize(ops.to_device(self.x), ops.to_device(self.y).reshape(-1), dict(X) if X else None)
			finally:
				self.kernel_object = saved
		acc = ops.zeros(2)
		for K in range(f.nblk):
			if K % self.Pr == self.myr and K % self.Pc == self.myc:
				Lkk, _ = f.winv[K]
				acc[0] += ops.logdet(Lkk)
		if self.myr == 0 and f.nc > 0:
			z = f.zloc[:, :f.nc * f.NB]
			acc[1] = ops.row_sums(z, z.reshape(-1))[1][0]
		self._allreduce(acc, dist.ReduceOp.SUM)
		w = float(weight) if not torch.is_tensor(weight) else float(weight.item())
		val = 0.5 * acc[1] + 0.5 * w * 2.0 * acc[0]
		del self._pinned_keep[:]
		return _lib.like_input(val.reshape(1, 1), self.x)
