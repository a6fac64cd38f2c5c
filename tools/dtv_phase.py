import sys, time, torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
n, m = 65536, 4096
x = torch.rand(n, 16, dtype=torch.float64, device=dev) * 2 - 1
xt = torch.rand(m, 16, dtype=torch.float64, device=dev) * 2 - 1
il = torch.full((16,), 0.25, dtype=torch.float64, device=dev)
K = torch.empty(n, n, dtype=torch.float64, device=dev)
X = torch.empty(m, n, dtype=torch.float64, device=dev)
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, 0)), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, 16)), dtype=torch.uint8, device=dev)
tw = torch.empty(int(lib.stpy_trsm_workspace_bytes(L.F64, m, n, 0)), dtype=torch.uint8, device=dev)
def gram():
	L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 16, L.ptr(x), n, 16, 16, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel() * ws.element_size(), L.stream_ptr()), "gram")
def potrf():
	L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel() * work.element_size(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
def kstar():
	L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 16, L.ptr(xt), m, 16, 16, None, L.ptr(il), 1.0, 0.0, 0.0, 0, 0, L.ptr(X), n, L.ptr(ws), ws.numel() * ws.element_size(), L.stream_ptr()), "gram")
def trsm():
	L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(X), n, 0, 0, L.ptr(tw), tw.numel() * tw.element_size(), L.stream_ptr()), "trsm")
def T(f):
	torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for rnd in range(2):
	for v in (0, 8192, 1):
		lib.stpy_tune(6, v)
		gram(); tp = T(potrf); kstar(); tt = T(trsm)
		print("dtv threshold %5d: potrf %.1f ms  trsm %.1f ms" % (v, tp, tt), flush=True)
# the trailing-update GEMM alone at the matrix size of the factorisation (ldc = 65536), operands with ld = 1024
P = torch.randn(n, 1024, dtype=torch.float64, device=dev)
for rr in (65536, 49152, 32768, 16384):
	for v in (0, 8192, 0, 8192):
		lib.stpy_tune(6, v)
		f = lambda: L.check(lib.stpy_gemm_nt(L.F64, rr, rr, 1024, L.ptr(P), 1024, L.ptr(P), 1024, L.ptr(K), n, 1, 1, L.stream_ptr()), "gemm")
		f(); t = min(T(f) for _ in range(3))
		print("update rows=%d ldc=%d dtv=%d: %.2f ms %.2f TF" % (rr, n, v, t, float(rr) * (rr + 128) * 1024 / t / 1e9), flush=True)
lib.stpy_tune(6, 1024)
import ctypes
def prof(tag):
	ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
	lib.stpy_profile_read(tag, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt))
	return ms.value, cnt.value
for v in (0, 8192, 0, 8192):
	lib.stpy_tune(6, v)
	gram(); torch.cuda.synchronize()
	lib.stpy_profile_enable(1)
	tp = T(potrf)
	lib.stpy_profile_enable(0)
	print("in-situ dtv=%d: potrf %.1f ms | summed launch time: trailing %.1f ms (%d)  panel GEMMs %.1f ms (%d)  diag blocks %.1f ms (%d)" % (
		(v, tp) + prof(0) + prof(1) + prof(3)), flush=True)
lib.stpy_tune(6, 1024)
