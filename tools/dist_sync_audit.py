"""Lists the host-synchronising torch calls of one DistributedGaussianProcess fit + predict (world size 1,
RCCL or gloo): torch.cuda.set_sync_debug_mode("warn") and the warnings grouped by source line.
usage: STPY_BACKEND=nccl python tools/dist_sync_audit.py [n] [m] [nb_dist]"""
import collections, os, sys, warnings
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
	n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
	m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
	nb = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
	os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29537")
	os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
	dist.init_process_group(backend=os.environ.get("STPY_BACKEND", "gloo"))
	from stpy_amd.parallel.block_cyclic import DistributedGaussianProcess
	dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
	d = 16
	g = torch.Generator().manual_seed(1234)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = torch.sin(x.sum(1, keepdim=True))
	xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	gp = DistributedGaussianProcess(gamma=4.0, s=0.1, kernel_name="squared_exponential", d=d, nb_dist=nb)
	gp.fit_gp(x, y); gp.mean_std(xt)                # warm: allocator, lazy communicators
	torch.cuda.synchronize()
	torch.cuda.set_sync_debug_mode("warn")
	with warnings.catch_warnings(record=True) as w:
		warnings.simplefilter("always")
		gp.fit_gp(x, y)
		nfit = len(w)
		gp.mean_std(xt)
	torch.cuda.set_sync_debug_mode("default")
	where = collections.Counter()
	import traceback
	for i, r in enumerate(w):
		where[("fit" if i < nfit else "predict", "%s:%d" % (os.path.basename(r.filename), r.lineno), str(r.message)[:60])] += 1
	print("block steps: %d" % gp.nblk)
	for k, c in sorted(where.items()):
		print(c, k)
	dist.destroy_process_group()

if __name__ == "__main__":
	main()
