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


def _worker(rank, world, port, grid, n, d, m, nb_dist, kernel_name, nu, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		torch.set_num_threads(1)
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		from tests.cpu_local_ops import CpuLocalOps
		x, y, xt = _data(n, d, m)
		gp = DistributedGaussianProcess(gamma=1.3, s=0.2, kappa=1.1, kernel_name=kernel_name, nu=nu, d=d, grid=grid,
										nb_dist=nb_dist, ops=CpuLocalOps())
		gp.fit_gp(torch.from_numpy(x), torch.from_numpy(y))
		mu, std = gp.mean_std(torch.from_numpy(xt))
		lml = gp.log_marginal()
		if rank == 0:
			q.put((mu.numpy(), std.numpy(), lml.numpy()))
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
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	mu, std, lml = q.get()
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


def test_default_grid():
	from stpy_amd.parallel.block_cyclic import default_grid
	assert [default_grid(p) for p in (1, 2, 4, 6, 8)] == [(1, 1), (1, 2), (2, 2), (2, 3), (2, 4)]
