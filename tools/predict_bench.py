"""Times mean_std alone (the block solve X = K* L^-T dominates) after one fit, for several panel widths,
with and without the K-pass workspace.  usage: python tools/predict_bench.py [n] [m]"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess, _lib as L

def main():
	n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
	m = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
	d = 16
	dev = torch.device("cuda:0")
	g = torch.Generator().manual_seed(1)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = torch.sin(x.sum(1, keepdim=True))
	xt = (torch.rand(m, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	gp = GaussianProcess(gamma=d ** 0.5, s=0.1, kernel_name="squared_exponential", d=d)
	gp.fit_gp(x, y)
	lib = L.load()
	X = torch.empty((m, n), dtype=torch.float64, device=dev)
	# negative nb: the right-looking sweep (stpy_tune key 5)
	combos = [(-512, False, 1024, 2048), (-1024, False, 1024, 2048), (0, False, 1024, 2048), (0, True, 1024, 2048), (512, True, 1024, 2048), (1024, True, 1024, 2048)]
	for nb in ():
		for depth, wgt in ((1024, 2048), (2048, 2048), (1024, 4096), (512, 4096), (2048, 1024)):
			combos.append((nb, True, depth, wgt))
	for nb, use_work, depth, wgt in combos:
		lib.stpy_tune(3, depth); lib.stpy_tune(4, wgt); lib.stpy_tune(5, 1 if nb < 0 else 0); nb = abs(nb)
		if True:
			wb = int(lib.stpy_trsm_workspace_bytes(L.F64, m, n, nb)) if use_work else 0
			wk = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
			ts = []
			for it in range(3):
				gp.kernel_object._kernel_into(gp._xd, xt, X)
				torch.cuda.synchronize(); t0 = time.perf_counter()
				L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(gp._L), gp._L.stride(0), L.ptr(gp._winv), gp._winv.numel(), L.ptr(X), X.stride(0), nb, 0,
											   L.ptr(wk) if use_work else None, wk.numel() if use_work else 0, L.stream_ptr()), "trsm")
				torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
			t = min(ts)
			print("trsm n=%d m=%d nb=%d work=%d depth=%d wg=%d: %.1f ms  %.1f TF" % (n, m, nb, use_work, depth, wgt, t * 1e3, float(m) * n * n / t / 1e12), flush=True)

if __name__ == "__main__":
	main()
