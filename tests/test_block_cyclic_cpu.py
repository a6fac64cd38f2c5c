"""
Multi-process CPU tests (gloo, world_size 2 and 4) of the 2-D block-cyclic fit / predict schedule
(stpy_amd/parallel/block_cyclic.py) with the CPU local-ops stand-in injected; results are compared
with the single-process oracle.  Runs without a GPU.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import gp_oracle as O
from tests.conftest import rel_err


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _data(n, d, m, seed=7):
	rng = np.random.RandomState(seed)
	x = rng.uniform(-1, 1, size=(n, d))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(n, 1))
	xt = rng.uniform(-1, 1, size=(m, d))
	return x, y, xt


def _worker(rank, world, port, grid, n, d, m, nb_dist, kernel_name, nu, q, opts=None):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		torch.set_num_threads(1)
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		from tests.cpu_local_ops import CpuLocalOps
		x, y, xt = _data(n, d, m)
		gp = DistributedGaussianProcess(gamma=1.3, s=0.2, kappa=1.1, kernel_name=kernel_name, nu=nu, d=d, grid=grid,
										nb_dist=nb_dist, ops=CpuLocalOps(), **(opts or {}))
		gp.fit_gp(torch.from_numpy(x), torch.from_numpy(y))
		mu, std = gp.mean_std(torch.from_numpy(xt))
		lml = gp.log_marginal()
		if opts and opts.get("audit"):
			# order audit: every member of a communicator must have issued the identical sequence of (op, root, bytes, stream)
			logs = [None] * world
			dist.all_gather_object(logs, (gp.Pr, gp.Pc, list(gp.audit)))
			if rank == 0:
				q.put((mu.numpy(), std.numpy(), lml.numpy(), logs, dict(gp.stats)))
			return
		# the wider estimator surface: full covariance, chunked prediction, kwargs overrides, add_data_point
		mu_f, cov = gp.mean_std(torch.from_numpy(xt[:11]), full=True)
		gp.max_size = 16
		mu_c, std_c = gp.mean_std(torch.from_numpy(xt))
		gp.max_size = 10000
		lml_o = gp.log_marginal(gp.kernel_object, {'0': {'gamma': torch.tensor(0.9, dtype=torch.float64)}}, 0.5)
		lml_again = gp.log_marginal(gp.kernel_object, {}, 1.0)          # stored parameters untouched by the override
		gp.add_data_point(torch.from_numpy(xt[:5]), torch.from_numpy(np.cos(xt[:5].sum(axis=1, keepdims=True))))
		mu_a, std_a = gp.mean_std(torch.from_numpy(xt[5:]))
		if rank == 0:
			q.put((mu.numpy(), std.numpy(), lml.numpy(), mu_f.numpy(), cov.numpy(), mu_c.numpy(), std_c.numpy(), lml_o.numpy(), lml_again.numpy(),
				   mu_a.numpy(), std_a.numpy(), dict(gp.stats)))
		# every rank must hold the same replicated result
		ref = mu.clone()
		dist.broadcast(ref, src=0)
		assert torch.equal(ref, mu)
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize("world,grid,n,nb_dist,kernel_name,nu", [
	(2, (1, 2), 700, 128, "squared_exponential", 1.5),
	(2, (2, 1), 700, 128, "squared_exponential", 1.5),
	(2, None, 512, 256, "matern", 2.5),
	(4, (2, 2), 900, 128, "squared_exponential", 1.5),
	(4, (1, 4), 640, 128, "matern", 1.5),
	(4, (4, 1), 300, 128, "squared_exponential", 1.5),
	(6, (2, 3), 1500, 128, "squared_exponential", 1.5),      # lcm 6: strided column exchange on both sides
	(8, (2, 4), 2100, 128, "squared_exponential", 1.5),      # the 8-GPU default grid
])
def test_block_cyclic_matches_oracle(world, grid, n, nb_dist, kernel_name, nu):
	d, m = 3, 37
	ctx = mp.get_context("spawn")
	q = ctx.SimpleQueue()
	port = _free_port()
	procs = [ctx.Process(target=_worker, args=(r, world, port, grid, n, d, m, nb_dist, kernel_name, nu, q)) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(timeout=300)
		if p.is_alive():          # a hung rank must not stay behind
			p.terminate()
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	mu, std, lml, mu_f, cov, mu_c, std_c, lml_ov, lml_again, mu_a, std_a, stats = q.get()
	x, y, xt = _data(n, d, m)
	params = {"gamma": 1.3, "kappa": 1.1}
	if kernel_name == "matern":
		params["nu"] = nu
	spec = [(kernel_name, params, "-")]
	L, alpha = O.fit(x, y, spec, 0.2)
	mu_o, std_o = O.mean_std(x, L, alpha, xt, spec)
	lml_o = O.log_marginal(x, y, spec, 0.2)
	assert rel_err(mu, mu_o) < 1e-10 and rel_err(std, std_o) < 1e-10
	assert abs(lml[0, 0] - lml_o[0, 0]) / abs(lml_o[0, 0]) < 1e-10
	mu_fo, cov_o = O.mean_cov(x, L, alpha, xt[:11], spec)
	assert cov.shape == (11, 11) and rel_err(cov, cov_o) < 1e-10 and rel_err(mu_f, mu_fo) < 1e-10
	assert rel_err(mu_c, mu_o) < 1e-10 and rel_err(std_c, std_o) < 1e-10          # chunks of 16 test points
	ov = O.log_marginal(x, y, spec, 0.2, overrides={'0': {'gamma': 0.9}}, weight=0.5)
	assert abs(lml_ov[0, 0] - ov[0, 0]) / abs(ov[0, 0]) < 1e-10
	assert abs(lml_again[0, 0] - lml_o[0, 0]) / abs(lml_o[0, 0]) < 1e-10
	x2, y2 = np.concatenate([x, xt[:5]]), np.concatenate([y, np.cos(xt[:5].sum(axis=1, keepdims=True))])
	L2, alpha2 = O.fit(x2, y2, spec, 0.2)
	mu_ao, std_ao = O.mean_std(x2, L2, alpha2, xt[5:], spec)
	assert rel_err(mu_a, mu_ao) < 1e-10 and rel_err(std_a, std_ao) < 1e-9
	assert stats["collectives"] > 0 and stats["bcast_bytes"] > 0


_AG, _FA, _BB, _FB = ({"transport": "collective", "col_exchange": "allgather"}, {"transport": "fanout", "col_exchange": "allgather"},
					  {"transport": "collective", "col_exchange": "bcast"}, {"transport": "fanout", "col_exchange": "bcast"})


@pytest.mark.parametrize("world,grid,n,opts", [
	(2, (1, 2), 520, _AG), (2, (1, 2), 520, _FA), (2, (2, 1), 520, _AG), (2, (2, 1), 520, _FB),
	(4, (2, 2), 900, _AG), (4, (2, 2), 900, _FA), (4, (2, 2), 900, _BB), (4, (2, 2), 900, _FB),
	(6, (2, 3), 1100, _AG), (6, (3, 2), 1100, _FA), (6, (3, 2), 1100, _BB),
	(8, (2, 4), 1700, _AG), (8, (2, 4), 1700, _FA), (8, (2, 4), 1700, _BB), (8, (4, 2), 1300, _AG), (8, (4, 2), 1300, _FB),
], ids=lambda v: ("%s+%s" % (v["transport"], v["col_exchange"])) if isinstance(v, dict) else None)
def test_collective_order_audit(world, grid, n, opts):
	"""What hangs on RCCL and passes on host-staged gloo is a member of a communicator issuing a different sequence of collectives
	than its peers.  Every rank logs, per communicator (process row, process column, world), the sequence of (operation, root,
	bytes, issuing logical stream) of a fit + predict + log-marginal; the sequences of all members of a communicator must be
	identical, for both transports (RCCL broadcast / point-to-point fan-out) and both forms of the column exchange.  The
	result is checked against the oracle as well, so each option is also a parity case."""
	d, m = 3, 29
	ctx = mp.get_context("spawn")
	q = ctx.SimpleQueue()
	port = _free_port()
	o = dict(opts, audit=True)
	procs = [ctx.Process(target=_worker, args=(r, world, port, grid, n, d, m, 128, "squared_exponential", 1.5, q, o)) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(timeout=300)
		if p.is_alive():
			p.terminate()
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	mu, std, lml, logs, stats = q.get()
	Pr, Pc = logs[0][0], logs[0][1]
	assert (Pr, Pc) == tuple(grid)
	members = {"world": list(range(world))}
	for r in range(Pr):
		members["row%d" % r] = [r * Pc + c for c in range(Pc)]
	for c in range(Pc):
		members["col%d" % c] = [r * Pc + c for r in range(Pr)]
	seen = 0
	for name, ranks in members.items():
		seqs = [[e[1:] for e in logs[r][2] if e[0] == name] for r in ranks]
		for r, sq in zip(ranks[1:], seqs[1:]):
			assert sq == seqs[0], "communicator %s: rank %d issued a different sequence than rank %d (first difference at %s)" % (
				name, r, ranks[0], next((i for i, (a, b) in enumerate(zip(sq, seqs[0])) if a != b), min(len(sq), len(seqs[0]))))
		seen += len(seqs[0])
		# no rank logged a communicator it is not a member of
		for r in range(world):
			if r not in ranks:
				assert not [e for e in logs[r][2] if e[0] == name]
	assert seen > 0
	# the look-ahead really issues collectives from both logical streams (fit) -- the ordering the audit is about
	streams = {e[4] for e in logs[0][2]}
	assert "side" in streams and "main" in streams
	kinds = {e[1] for lg in logs for e in lg[2]}
	assert ("allgather" in kinds) == (opts["col_exchange"] == "allgather" and Pr > 1)
	assert any(k.startswith("bcast/" + opts["transport"]) for k in kinds)
	x, y, xt = _data(n, d, m)
	spec = [("squared_exponential", {"gamma": 1.3, "kappa": 1.1}, "-")]
	L, alpha = O.fit(x, y, spec, 0.2)
	mu_o, std_o = O.mean_std(x, L, alpha, xt, spec)
	assert rel_err(mu, mu_o) < 1e-10 and rel_err(std, std_o) < 1e-10
	assert abs(lml[0, 0] - O.log_marginal(x, y, spec, 0.2)[0, 0]) / abs(lml[0, 0]) < 1e-10


def test_auto_nb_dist():
	from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
	assert [DistributedGaussianProcess.auto_nb_dist(n) for n in (4096, 65536, 131071, 131072, 262144)] == [1024, 1024, 1024, 2048, 2048]


def test_default_grid():
	from stpy_amd.parallel.block_cyclic import default_grid
	assert [default_grid(p) for p in (1, 2, 4, 6, 8)] == [(1, 1), (1, 2), (2, 2), (2, 3), (2, 4)]


def _rff_worker(rank, world, port, n, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		from stpy_amd.parallel.row_split import ShardedEmbedding, row_range

		class OracleEmbedding:          # stand-in for RFFEmbedding on a box without a GPU: the oracle's embed on the slab
			def __init__(self, W, m):
				self.W, self.m = W, m

			def get_m(self):
				return self.m

			def embed(self, x):
				return torch.from_numpy(O.rff_embed(x.numpy(), self.W, self.m))
		rng = np.random.RandomState(3)
		W = rng.normal(size=(64, 5)) / 0.7
		x = torch.from_numpy(np.random.RandomState(4).uniform(0, 1, size=(n, 5)))
		sh = ShardedEmbedding(OracleEmbedding(W, 64))
		r0, r1, z = sh.embed(x)
		assert (r0, r1) == row_range(n, rank, world) and z.shape == (r1 - r0, 64)
		full = sh.embed(x, gather=True)
		if rank == 0:
			q.put((full.numpy(), [row_range(n, r, world) for r in range(world)]))
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1000), (4, 700), (3, 128), (8, 262144 // 64)])
def test_rff_row_split(world, n):
	"""SURVEY.md section 8e last row: rows split in tile-aligned contiguous slabs, W replicated, no collective on the data
	path; the slabs tile [0, n) exactly and their concatenation is the single-process embed."""
	ctx = mp.get_context("spawn")
	q = ctx.SimpleQueue()
	port = _free_port()
	procs = [ctx.Process(target=_rff_worker, args=(r, world, port, n, q)) for r in range(world)]
	for p in procs:
		p.start()
	full, ranges = q.get()          # (read before joining: a large put blocks the child until the pipe is drained)
	for p in procs:
		p.join(timeout=120)
		if p.is_alive():
			p.terminate()
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	assert ranges[0][0] == 0 and ranges[-1][1] == n and all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
	assert all(a % 128 == 0 for a, _ in ranges)
	sizes = [b - a for a, b in ranges]
	assert max(sizes) - min(sizes) <= 128 + (128 - n % 128) % 128
	W = np.random.RandomState(3).normal(size=(64, 5)) / 0.7
	x = np.random.RandomState(4).uniform(0, 1, size=(n, 5))
	assert rel_err(full, O.rff_embed(x, W, 64)) < 1e-15          # (BLAS blocks a slab differently from the whole: last-bit differences)
