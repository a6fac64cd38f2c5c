"""
The block-cyclic schedule with the REAL HIP local ops: several ranks share the one GPU of the test
box (gloo rendezvous, device tensors staged through the host for the collectives), so the staircase
GEMM, the strided-view potrf/trsm calls and the index arithmetic run together on hardware.
Compared with the single-process GaussianProcess on the same inputs AND with the CPU oracle (mean, std, full
covariance, chunked prediction, log-marginal with and without kwargs overrides).  The collectives are staged through
pinned host memory with an event on the issuing stream only, so the main / side stream ordering of the schedule is
exercised as it would be on RCCL.
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
		gp = DistributedGaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d, grid=grid, nb_dist=nb_dist,
										force_path=True)
		assert gp._single is None
		gp.fit_gp(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
		mu, std = gp.mean_std(torch.from_numpy(xt).cuda())
		lml = gp.log_marginal()
		mu_f, cov = gp.mean_std(torch.from_numpy(xt[:40]).cuda(), full=True)
		gp.max_size = 128
		mu_c, std_c = gp.mean_std(torch.from_numpy(xt).cuda())
		gp.max_size = 10000
		lml_ov = gp.log_marginal(gp.kernel_object, {'0': {'gamma': torch.tensor(1.5, dtype=torch.float64)}}, 0.5)
		if rank == 0:
			q.put(tuple(t.cpu().numpy() for t in (mu, std, lml, mu_f, cov, mu_c, std_c, lml_ov)))
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
		if p.is_alive():          # a hung rank must not keep holding the GPU
			p.terminate()
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	mu, std, lml, mu_f, cov, mu_c, std_c, lml_ov = q.get()
	import stpy_amd
	x, y, xt = _data(n, d, m)
	GP = stpy_amd.GaussianProcess(gamma=2.0, s=0.1, kappa=1.0, kernel_name="squared_exponential", d=d)
	GP.fit_gp(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
	mu1, std1 = GP.mean_std(torch.from_numpy(xt).cuda())
	lml1 = GP.log_marginal(GP.kernel_object, {}, 1.0)
	assert rel_err(mu, mu1.cpu().numpy()) < 1e-9 and rel_err(std, std1.cpu().numpy()) < 1e-9
	assert abs(lml[0, 0] - float(lml1.item())) / abs(float(lml1.item())) < 1e-10
	# ... and against the CPU oracle: real HIP tile arithmetic + the distributed schedule, not a self-comparison
	spec = [("squared_exponential", {"gamma": 2.0, "kappa": 1.0}, "-")]
	L, alpha = O.fit(x, y, spec, 0.1)
	mu_o, std_o = O.mean_std(x, L, alpha, xt, spec)
	assert rel_err(mu, mu_o) < 1e-8 and rel_err(std, std_o) < 1e-8
	assert rel_err(mu_c, mu_o) < 1e-8 and rel_err(std_c, std_o) < 1e-8
	mu_fo, cov_o = O.mean_cov(x, L, alpha, xt[:40], spec)
	assert rel_err(cov, cov_o) < 1e-8 and rel_err(mu_f, mu_fo) < 1e-8
	lm_o = O.log_marginal(x, y, spec, 0.1)[0, 0]
	assert abs(lml[0, 0] - lm_o) / abs(lm_o) < 1e-8
	ov = O.log_marginal(x, y, spec, 0.1, overrides={'0': {'gamma': 1.5}}, weight=0.5)[0, 0]
	assert abs(lml_ov[0, 0] - ov) / abs(ov) < 1e-8


def _rff_worker(rank, world, port, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	torch.cuda.set_device(0)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		import stpy_amd
		from stpy_amd.parallel.row_split import ShardedEmbedding
		rng = np.random.RandomState(5)
		n, d, m = 8192 + 640, 64, 1024
		x = torch.from_numpy(rng.uniform(0, 1, size=(n, d)).astype(np.float32)).cuda()
		np.random.seed(9)
		emb = stpy_amd.RFFEmbedding(gamma=8.0, m=m, d=d)
		emb.W = emb.W.float().cuda()
		sh = ShardedEmbedding(emb)
		r0, r1, z = sh.embed(x)
		ref = O.rff_embed(x[r0:r1].double().cpu().numpy(), emb.W.double().cpu().numpy(), m)
		err = float(np.abs(z.cpu().numpy() - ref).max() / np.abs(ref).max())
		q.put((rank, r0, r1, err))
	finally:
		dist.destroy_process_group()


def test_rff_row_split_on_gpu(gpu_device):
	"""SURVEY.md section 8e last row with the real embed kernel: three ranks sharing the test box's GPU each embed their slab"""
	world = 3
	ctx = mp.get_context("spawn")
	q = ctx.SimpleQueue()
	port = _free_port()
	procs = [ctx.Process(target=_rff_worker, args=(r, world, port, q)) for r in range(world)]
	for p in procs:
		p.start()
	got = sorted(q.get() for _ in range(world))
	for p in procs:
		p.join(timeout=120)
		if p.is_alive():
			p.terminate()
	assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
	assert got[0][1] == 0 and got[-1][2] == 8192 + 640 and all(got[i][2] == got[i + 1][1] for i in range(world - 1))
	assert all(e < 2e-5 for (_, _, _, e) in got), got


def test_bench_self_launch_two_ranks_gloo():
	"""`python bench.py --gpus 2` started PLAINLY (no WORLD_SIZE): the script launches its two ranks itself (fresh child processes
	through torch.distributed.run, before this process has touched the GPU), relays rank 0's single JSON line and the children's
	exit code.  Rehearsal backend: the ranks share the one GPU of the test box and the collectives are staged through the host."""
	import json
	import subprocess
	import sys
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = dict(os.environ, STPY_BENCH_BACKEND="gloo")
	for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
		env.pop(k, None)
	r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "4096", "--no-cpu-baseline"],
					   capture_output=True, text=True, timeout=600, env=env, cwd=root)
	lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
	assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
	out = json.loads(lines[0])
	assert out["n_gpus"] == 2 and out["value"] is not None and out["value"] > 0 and out["steps"] == 1 and out["warmup"] == 0
	assert out["scaling"] == "strong" and out["higher_is_better"] is False and out["unit"] == "s"
	mg = out["multi_gpu"]
	assert mg["backend"] == "gloo" and mg["grid"] == "1x2" and len(mg["per_rank"]) == 2
	assert mg["selfcheck"]["mu_rel_err_vs_single_gpu_class"] < 1e-8 and mg["selfcheck"]["sigma_rel_err"] < 1e-8
	assert not out["result_check"]["nan"]


def test_bench_more_ranks_than_gpus_prints_one_json_error_line():
	"""`python bench.py --gpus 2` on the RCCL backend with ONE visible GPU: the ranks cannot get a device each; the run must end within
	seconds with exactly one JSON line carrying `value: null` and the reason, and a non-zero exit code (no hang, no bare traceback)."""
	import json
	import subprocess
	import sys
	import torch
	if torch.cuda.device_count() >= 2:
		pytest.skip("this box has a GPU per rank")
	root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
	env = dict(os.environ)
	for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "STPY_BENCH_BACKEND"):
		env.pop(k, None)
	r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--n", "4096", "--no-cpu-baseline", "--no-extra-configs"],
					   capture_output=True, text=True, timeout=300, env=env, cwd=root)
	lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
	assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
	out = json.loads(lines[0])
	assert out["value"] is None and out["n_gpus"] == 2 and "GPUs" in out["error"]
