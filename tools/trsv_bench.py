import sys, time, torch
sys.path.insert(0, ".")
from stpy_amd import GaussianProcess, _lib as L
lib = L.load()
dev = torch.device("cuda:0")
for n in (16384, 65536):
	d = 8
	g = torch.Generator().manual_seed(1)
	x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
	y = torch.sin(x.sum(1, keepdim=True))
	gp = GaussianProcess(gamma=d ** 0.5, s=0.1, kernel_name="squared_exponential", d=d)
	gp.fit_gp(x, y)
	yv = y.reshape(-1).clone(); z = torch.empty(n, dtype=torch.float64, device=dev); a = torch.empty(n, dtype=torch.float64, device=dev)
	for tr in (0, 1):
		for mode in (0, 1):          # stpy_tune key 16: 0 = chain of per-block launches, 1 = one dataflow launch
			lib.stpy_tune(16, mode)
			ts = []
			for _ in range(5):
				src = yv.clone()
				torch.cuda.synchronize(); t0 = time.perf_counter()
				L.check(lib.stpy_trsv(L.F64, n, L.ptr(gp._L), gp._L.stride(0), L.ptr(gp._winv), gp._winv.numel(), L.ptr(src), L.ptr(z), tr, L.stream_ptr()), "trsv")
				torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
			print("n=%d trsv trans=%d %s: %.3f ms (%.1f us per 128-block; L streamed at %.2f TB/s)" % (n, tr, "one launch" if mode else "step chain",
				  min(ts) * 1e3, min(ts) * 1e6 / (n / 128), n * n * 4.0 / min(ts) / 1e12), flush=True)
		lib.stpy_tune(16, 1)
	del gp, x, y
	torch.cuda.empty_cache()
