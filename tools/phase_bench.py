"""Phase times of one GaussianProcess step (gram / potrf / alpha / predict pieces) at a given size.
usage: python tools/phase_bench.py [n] [d] [m]"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess, _lib as L

def t(f, reps=5):
	f(); torch.cuda.synchronize()
	best = 1e9
	for _ in range(reps):
		torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize()
		best = min(best, time.perf_counter() - t0)
	return best * 1e3

def main():
	n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
	d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
	m = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
	dev = torch.device("cuda:0")
	g = torch.Generator().manual_seed(1)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = torch.sin(x.sum(1, keepdim=True))
	xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	gp = GaussianProcess(gamma=d ** 0.5, s=0.1, kernel_name="squared_exponential", d=d)
	lib = L.load()
	print("n=%d d=%d m=%d" % (n, d, m))
	print("fit_gp            %.2f ms" % t(lambda: gp.fit_gp(x, y)))
	print("mean_std          %.2f ms" % t(lambda: gp.mean_std(xt)))
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	print("  gram (lower)    %.2f ms" % t(lambda: gp.kernel_object._kernel_into(gp._xd, gp._xd, K, None, 0.01, True)))
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
	# sized for the widest panel tried below (the query depends on nb; a workspace sized for nb = 0 is too small for 1024)
	work = torch.empty(max(int(lib.stpy_potrf_workspace_bytes(L.F64, n, nb)) for nb in (0, 256, 512, 1024)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	def potrf(nb):
		gp.kernel_object._kernel_into(gp._xd, gp._xd, K, None, 0.01, True)
		L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel() * work.element_size(), nb, 0, L.ptr(info), L.stream_ptr()), "potrf")
	for nb in (0, 256, 512, 1024):
		print("  gram+potrf nb=%-4d %.2f ms" % (nb, t(lambda: potrf(nb))))
	X = torch.empty(m, n, dtype=torch.float64, device=dev)
	print("  K* gram         %.2f ms" % t(lambda: gp.kernel_object._kernel_into(gp._xd, xt, X)))
	for nb, ww in ((0, 1), (0, 0), (256, 1), (1024, 1)):
		wb = int(lib.stpy_trsm_workspace_bytes(L.F64, m, n, nb)) if ww else 0
		wk = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
		def trsm():
			gp.kernel_object._kernel_into(gp._xd, xt, X)
			L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(gp._L), gp._L.stride(0), L.ptr(gp._winv), gp._winv.numel(), L.ptr(X), X.stride(0), nb, 0, L.ptr(wk) if ww else None, wk.numel() if ww else 0, L.stream_ptr()), "trsm")
		print("  K* + trsm nb=%-4d work=%d %.2f ms" % (nb, ww, t(trsm)))

if __name__ == "__main__":
	main()
