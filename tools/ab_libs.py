"""In-process A/B of two builds of the library (tools/_ab/libstpy_hip_base.so = a baseline build, stpy_amd/libstpy_hip.so =
the tree): the trailing-update GEMM at a few shapes, then the whole fit + mean_std step through the C ABI calls the
estimator makes.  usage: python tools/ab_libs.py"""
import ctypes, os, sys, time, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stpy_amd import _lib as L

def bind(path):
	lib = ctypes.CDLL(path)
	for name, (res, args) in L.SIGNATURES.items():
		if hasattr(lib, name):
			fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
	return lib

libs = {"base": bind(os.path.join(ROOT, "tools", "_ab", "libstpy_hip_base.so")), "tree": bind(os.path.join(ROOT, "stpy_amd", "libstpy_hip.so"))}
dev = torch.device("cuda:0")

def timeit(f, reps=3):
	ts = []
	for _ in range(reps):
		torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
	return min(ts)

for n, k in ((32768, 1024), (32768, 256), (16384, 512)):
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	C = torch.randn(n, n, dtype=torch.float64, device=dev)
	res = {}
	for rnd in range(3):
		for name, lib in libs.items():
			f = lambda: lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr())
			f(); res.setdefault(name, []).append(timeit(f))
	fl = 2.0 * n * n * k * (0.5 + 64.0 / n)
	print("syrk n=%d k=%d: " % (n, k) + "  ".join("%s %.3f ms %.2f TF" % (nm, min(v) * 1e3, fl / min(v) / 1e12) for nm, v in res.items()), flush=True)
	del P, C

def step(lib, n, d, m, x, xt, y):
	code = L.F64
	il = torch.full((d,), 1.0 / math.sqrt(d), dtype=torch.float64, device=dev)
	K = torch.empty(n, n, dtype=torch.float64, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(code, n, n, d)), dtype=torch.uint8, device=dev)
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
	work = torch.empty(int(lib.stpy_potrf_workspace_bytes(code, n, 0)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	X = torch.empty(m, n, dtype=torch.float64, device=dev)
	ws2 = torch.empty(int(lib.stpy_gram_workspace_bytes(code, n, m, d)), dtype=torch.uint8, device=dev)
	tw = torch.empty(int(lib.stpy_trsm_workspace_bytes(code, m, n, 0)), dtype=torch.uint8, device=dev)
	def run():
		assert lib.stpy_gram(0, code, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()) == 0
		assert lib.stpy_potrf(code, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()) == 0
		assert lib.stpy_gram(0, code, L.ptr(x), n, d, L.ptr(xt), m, d, d, None, L.ptr(il), 1.0, 0.0, 0.0, 0, 0, L.ptr(X), n, L.ptr(ws2), ws2.numel(), L.stream_ptr()) == 0
		assert lib.stpy_trsm_right_lt(code, m, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(X), n, 0, 0, L.ptr(tw) if tw.numel() else None, tw.numel(), L.stream_ptr()) == 0
	return run, (K, X)

for n, d, m in ((16384, 8, 4096), (65536, 16, 4096)):
	x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
	xt = torch.rand(m, d, dtype=torch.float64, device=dev) * 2 - 1
	res, chk = {}, {}
	for name, lib in libs.items():
		run, (K, X) = step(lib, n, d, m, x, xt, None)
		run(); torch.cuda.synchronize()
		for rnd in range(3):
			res.setdefault(name, []).append(timeit(run, reps=1))
		chk[name] = (float(X[:, ::7].norm()), float(K.diagonal().sum()))
		del run, K, X
		torch.cuda.empty_cache()
	print("gram+potrf+K*+trsm n=%d m=%d: " % (n, m) + "  ".join("%s %.4f s" % (nm, min(v)) for nm, v in res.items()) + "   |X| %r  |L| %r" % (chk["base"], chk["tree"]), flush=True)
