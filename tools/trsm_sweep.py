"""stpy_trsm_right_lt over a grid of (n, m) for the values of one stpy_tune key (default 5: the block-solve algorithm), one process.
usage: python tools/trsm_sweep.py [values, default 0,4] [key, default 5] [f64only] [m list]"""
import sys, time
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
vals = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "4"])]
KEY = int(sys.argv[2]) if len(sys.argv) > 2 else 5
KEY_DEFAULT = int(lib.stpy_tune_get(KEY))
F64ONLY = len(sys.argv) > 3
MLIST = tuple(int(v) for v in sys.argv[4].split(",")) if len(sys.argv) > 4 else None

def factor(n, dtype):
	code = L.dtype_code(dtype)
	x = torch.rand(n, 8, dtype=dtype, device=dev) * 2 - 1
	il = torch.full((8,), 0.35, dtype=dtype, device=dev)
	K = torch.empty(n, n, dtype=dtype, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(code, n, n, 8)), dtype=torch.uint8, device=dev)
	L.check(lib.stpy_gram(0, code, L.ptr(x), n, 8, L.ptr(x), n, 8, 8, None, L.ptr(il), 1.0, 0.0, 0.01 if dtype == torch.float64 else 0.1, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=dtype, device=dev)
	work = torch.empty(int(lib.stpy_potrf_workspace_bytes(code, n, 0)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	L.check(lib.stpy_potrf(code, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
	assert int(info.item()) == 0
	return K, winv

for dtype in ((torch.float64,) if F64ONLY else (torch.float64, torch.float32)):
	code = L.dtype_code(dtype)
	for n in (4096, 8192, 16384, 32768, 65536):
		if dtype == torch.float32 and n not in (16384, 65536):
			continue
		K, winv = factor(n, dtype)
		for m in (MLIST or ((2048, 4096, 10112) if F64ONLY else (256, 1024, 4096, 10112))):
			B = torch.rand(m, n, dtype=dtype, device=dev)
			X = torch.empty_like(B)
			res, outs = {}, {}
			for rnd in range(3):
				for v in vals:
					lib.stpy_tune(KEY, v)
					tw = torch.empty(int(lib.stpy_trsm_workspace_bytes(code, m, n, 0)), dtype=torch.uint8, device=dev)
					X.copy_(B); torch.cuda.synchronize(); t0 = time.perf_counter()
					L.check(lib.stpy_trsm_right_lt(code, m, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(X), n, 0, 0, L.ptr(tw) if tw.numel() else None, tw.numel(), L.stream_ptr()), "trsm")
					torch.cuda.synchronize(); res.setdefault(v, []).append(time.perf_counter() - t0)
					if rnd == 0:
						outs[v] = X[:64].clone()
			lib.stpy_tune(KEY, KEY_DEFAULT)
			diff = max(float((outs[v] - outs[vals[0]]).norm() / outs[vals[0]].norm()) for v in vals)
			print("%s n %6d m %6d: " % ("f64" if dtype == torch.float64 else "f32", n, m) + "  ".join("[%d] %8.2f ms %5.1f TF" % (v, min(res[v]) * 1e3, float(n) * n * m / min(res[v]) / 1e12) for v in vals) + "   rel diff %.1e" % diff, flush=True)
			del B, X
		del K, winv
		torch.cuda.empty_cache()
