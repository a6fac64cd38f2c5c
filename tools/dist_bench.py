"""Times DistributedGaussianProcess on the ranks it is launched with (gloo staging when several ranks
share one GPU; world_size 1 measures the per-step overhead of the block-cyclic code path itself).
usage: python tools/dist_bench.py [n] [m] [nb_dist ...]"""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
	n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
	m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
	nbs = [int(v) for v in sys.argv[3:]] or [512, 1024]
	os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
	os.environ.setdefault("MASTER_PORT", "29531")
	os.environ.setdefault("RANK", "0")
	os.environ.setdefault("WORLD_SIZE", "1")
	backend = os.environ.get("STPY_BACKEND", "gloo")
	dist.init_process_group(backend=backend)
	from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
	from stpy_amd import _lib
	for kv in filter(None, os.environ.get("STPY_TUNE", "").split(",")):       # e.g. STPY_TUNE="8=0" (no K=128 volley kernel)
		k, v = kv.split("=")
		_lib.load().stpy_tune(int(k), int(v))
	dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count() if backend == "nccl" else 0)
	torch.cuda.set_device(dev)
	d = 16
	g = torch.Generator().manual_seed(1234)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = (torch.sin(x.sum(1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=g, dtype=torch.float64).to(dev))
	xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	for nb in nbs:
		gp = DistributedGaussianProcess(gamma=d ** 0.5, s=0.1, kernel_name="squared_exponential", d=d, nb_dist=nb)
		for it in range(int(os.environ.get('STPY_ITERS', '2'))):
			torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
			gp.fit_gp(x, y)
			torch.cuda.synchronize(); dist.barrier(); t1 = time.perf_counter()
			mu, std = gp.mean_std(xt)
			torch.cuda.synchronize(); dist.barrier(); t2 = time.perf_counter()
			if dist.get_rank() == 0:
				print("world %d grid %dx%d n %d m %d NB %d it %d: fit %.3f s predict %.3f s total %.3f s  |mu| %.6f" % (
					dist.get_world_size(), gp.Pr, gp.Pc, n, m, nb, it, t1 - t0, t2 - t1, t2 - t0, float(mu.norm())), flush=True)
		del gp
	dist.destroy_process_group()

if __name__ == "__main__":
	main()
