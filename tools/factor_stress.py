"""Repeated factorisations of the same matrix with a full L L^T - K check per 128-block each time: catches intermittent (timing-
dependent) faults of the panel chain that a sampled check misses.   usage: python tools/factor_stress.py [n] [d] [reps] [lib.so]"""
import ctypes, math, sys
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
lib = L.load()
if len(sys.argv) > 4:
	lib = ctypes.CDLL(sys.argv[4])
	for name, (res, args) in L.SIGNATURES.items():
		if hasattr(lib, name):
			fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
il = torch.full((d,), 1.0 / math.sqrt(d), dtype=torch.float64, device=dev)
K0 = torch.empty(n, n, dtype=torch.float64, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
assert lib.stpy_gram(0, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, 0, 0, L.ptr(K0), n, L.ptr(ws), ws.numel(), L.stream_ptr()) == 0
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, 0)), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
I128 = torch.eye(128, dtype=torch.float64, device=dev)
worst, nbad_total = 0.0, 0
for rep in range(reps):
	K = K0.clone()
	assert lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()) == 0
	torch.cuda.synchronize()
	Lf = torch.tril(K)
	full = Lf @ Lf.T - K0
	eb = torch.tril(full.abs().reshape(n // 128, 128, n // 128, 128).amax(dim=(1, 3)))
	idx = torch.nonzero(eb > 1e-11)
	W = winv.reshape(-1, 128, 128)
	wbad = [c for c in range(n // 128) if float((W[c] @ Lf[c * 128:(c + 1) * 128, c * 128:(c + 1) * 128] - I128).abs().max()) > 1e-11]
	worst = max(worst, float(eb.max()))
	nbad_total += idx.shape[0] + len(wbad)
	print("rep %d: info %d  blocks of L L^T - K above 1e-11: %d %s   inverse blocks off: %s   max %.2e" % (rep, int(info.item()), idx.shape[0], idx[:6].tolist(), wbad[:6], float(eb.max())), flush=True)
	del full, Lf, K
print("STRESS %s: %d bad blocks in %d factorisations, worst |L L^T - K| %.2e" % ("FAILED" if nbad_total else "ok", nbad_total, reps, worst))
sys.exit(1 if nbad_total else 0)
