import ctypes, os, sys, time
import torch
ROOT = "."
sys.path.insert(0, ROOT)
from stpy_amd import _lib as L
def bind(path):
	lib = ctypes.CDLL(path)
	for name, (res, args) in L.SIGNATURES.items():
		if hasattr(lib, name):
			fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
	return lib
libs = {"base": bind("tools/_ab/libstpy_hip_base.so"), "tree": bind("stpy_amd/libstpy_hip.so")}
dev = torch.device("cuda:0")
n = 32768
C = torch.randn(n, n, dtype=torch.float64, device=dev)
for k in (128, 256, 384, 512, 768, 1024, 2048):
	P = torch.randn(n, k, dtype=torch.float64, device=dev)
	res = {}
	for rnd in range(3):
		for name, lib in libs.items():
			f = lambda: lib.stpy_gemm_nt(L.F64, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr())
			f(); torch.cuda.synchronize()
			ts = []
			for _ in range(3):
				torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
			res.setdefault(name, []).append(min(ts))
	print("k %5d: " % k + "  ".join("%s %.3f ms" % (nm, min(v) * 1e3) for nm, v in res.items()) + "   delta %+.3f ms" % ((min(res["tree"]) - min(res["base"])) * 1e3), flush=True)
