"""The block-cyclic code path on ONE rank (RCCL process group of size 1, force_path=True) against the single-GPU estimator:
fit and predict timed separately -- what the distributed schedule itself costs before any communication exists.
usage: python tools/dist_single_rank.py [n] [nb_dist]"""
import os, sys, time, math
import torch
sys.path.insert(0, ".")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
from stpy_amd import GaussianProcess
from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
nbd = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d, m = 16, 4096
g = torch.Generator().manual_seed(3)
x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
y = torch.sin(x.sum(1, keepdim=True))
xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
def timed(f, reps=2):
	best = 1e9
	for _ in range(reps):
		torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
	return best, r
for name, gp in (("single", GaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d)),
				 ("block-cyclic, 1 rank", DistributedGaussianProcess(gamma=math.sqrt(d), s=0.1, kernel_name="squared_exponential", d=d, nb_dist=nbd, force_path=True))):
	gp.fit_gp(x, y); gp.mean_std(xt)
	tf, _ = timed(lambda: gp.fit_gp(x, y))
	tp, (mu, sd) = timed(lambda: gp.mean_std(xt))
	print("%-22s fit %.4f s  predict %.4f s  (mu norm %.10f)" % (name, tf, tp, float(mu.norm())), flush=True)
dist.destroy_process_group()
