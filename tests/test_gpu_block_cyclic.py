"""
The block-cyclic schedule with the REAL HIP local ops: several ranks share the one GPU of the test
box (gloo rendezvous, device tensors staged through the host for the collectives), so the staircase
GEMM, the strided-view potrf/trsm calls and the index arithmetic run together on hardware.
Compared with the single-process GaussianProcess on the same inputs.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _data(n, d, m, seed=11):
	rng = np.random.RandomState(seed)
	x = rng.uniform(-1, 1, size=(n, d))
	y = np.sin(x.sum(axis=1, keepdims=True)) + 0.1 * rng.normal(size=(n, 1))
	xt = rng.uniform(-1, 1, size=(m, d))
	return x, y, xt


def _worker(rank, world, port, grid, n, d, m, nb_dist, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	torch.cuda.set_device(0)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
		x, y, xt = _data(n, d, m)
		gp = DistributedGaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d, grid=grid, nb_dist=nb_dist)
		gp.fit_gp(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
		mu, std = gp.mean_std(torch.from_numpy(xt).cuda())
		lml = gp.log_marginal()
		if rank == 0:
			q.put((mu.cpu().numpy(), std.cpu().numpy(), lml.cpu().numpy()))
	finally:
		dist.destroy_process_group()


@pytest.mark.parametrize("world,grid,n,nb_dist", [(1, (1, 1), 3000, 512), (2, (1, 2), 2500, 256), (4, (2, 2), 3333, 256), (4, (1, 4), 2048, 128)])
def test_block_cyclic_on_gpu(gpu_device, world, grid, n, nb_dist):
	d, m = 8, 300
	ctx = mp.get_context("spawn")
	q = ctx.SimpleQueue()
	port = _free_port()
	procs = [ctx.Process(target=_worker, args=(r, world, port, grid, n, d, m, nb_dist, q)) for r in range(world)]
	for p in procs:
		p.start()
	for p in procs:
		p.join(timeout=300)
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	mu, std, lml = q.get()
	import stpy_amd
	x, y, xt = _data(n, d, m)
	GP = stpy_amd.GaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
	mu1, std1 = GP.mean_std(torch.from_numpy(xt).cuda())
	lml1 = GP.log_marginal(GP.kernel_object, {}, 1.0)
	assert rel_err(mu, mu1.cpu().numpy()) < 1e-9 and rel_err(std, std1.cpu().numpy()) < 1e-9
	assert abs(lml[0, 0] - float(lml1.item())) / abs(float(lml1.item())) < 1e-10
