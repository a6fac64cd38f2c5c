import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
for n in (8192, 16384, 24576):
	x = torch.rand(n, 8, dtype=torch.float64, device=dev) * 2 - 1
	il = torch.full((8,), 0.35, dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, 8)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	out = []
	for nb in (0, 128, 256, 384, 512, 640, 768):
		work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, nb)), dtype=torch.uint8, device=dev)
		def f():
			L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 8, L.ptr(x), n, 8, 8, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
			L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), nb, 0, L.ptr(info), L.stream_ptr()), "potrf")
		f(); torch.cuda.synchronize()
		ts = []
		for _ in range(4):
			torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
		out.append("nb %d: %.2f ms" % (nb, min(ts) * 1e3))
	print("n %d  " % n + "  ".join(out), flush=True)
