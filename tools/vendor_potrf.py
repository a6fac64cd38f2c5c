"""The vendor library's Cholesky and triangular solve (torch.linalg on ROCm = hipSOLVER / rocSOLVER / rocBLAS, or MAGMA)
beside stpy_potrf / stpy_trsm_right_lt on the same matrices -- context for the numbers in DESIGN.md, not a dependency.
usage: python tools/vendor_potrf.py [n ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stpy_amd import _lib as L

def timed(f, reps=2):
	best = 1e9
	for _ in range(reps):
		torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); torch.cuda.synchronize()
		best = min(best, time.perf_counter() - t0)
	return best, r

def main():
	ns = [int(v) for v in sys.argv[1:]] or [16384, 32768]
	lib = L.load()
	dev = torch.device("cuda", 0)
	m, d = 4096, 8
	print("linalg backend:", torch.backends.cuda.preferred_linalg_library())
	for n in ns:
		x = torch.rand(n, d, dtype=torch.float64, device=dev) * 2 - 1
		il = torch.full((d,), 0.35, dtype=torch.float64, device=dev)
		K = torch.empty(n, n, dtype=torch.float64, device=dev)
		ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
		def gram(lower):
			L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, lower, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
		B = torch.rand(m, n, dtype=torch.float64, device=dev)
		# ---- vendor
		gram(0)
		torch.linalg.cholesky(K[:256, :256].clone())          # library initialisation outside the timing
		try:
			tv, Lv = timed(lambda: torch.linalg.cholesky(K), reps=2)
			tsv, Xv = timed(lambda: torch.linalg.solve_triangular(Lv, B.T, upper=False), reps=2)      # L^-1 B^T  (n x m)
		except Exception as exc:                     # (N = 65 536: "invalid configuration argument" inside the library path)
			print("n %6d  vendor path failed: %s" % (n, str(exc).splitlines()[0]), flush=True)
			return
		# ---- ours
		winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
		work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, 0)), dtype=torch.uint8, device=dev)
		info = torch.zeros(1, dtype=torch.int32, device=dev)
		def ours():
			gram(1)
			L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
		tg, _ = timed(lambda: gram(1))
		to, _ = timed(ours, reps=3)
		to -= tg
		X = B.clone()
		tw = torch.empty(int(lib.stpy_trsm_workspace_bytes(L.F64, m, n, 0)), dtype=torch.uint8, device=dev)
		def ours_trsm():
			X.copy_(B)
			L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(X), n, 0, 0, L.ptr(tw), tw.numel(), L.stream_ptr()), "trsm")
		tc, _ = timed(lambda: X.copy_(B))
		ts, _ = timed(ours_trsm, reps=3)
		ts -= tc
		err = float((X - Xv.T).norm() / Xv.norm())
		errL = float((torch.tril(K) - Lv).norm() / Lv.norm())
		fp, ft = n ** 3 / 3, float(n) * n * m
		print("n %6d  potrf: vendor %8.1f ms (%5.1f TF)  stpy %8.1f ms (%5.1f TF)   |  trsm m=%d: vendor %8.1f ms (%5.1f TF)  stpy %8.1f ms (%5.1f TF)   | rel diff L %.1e X %.1e" % (
			n, tv * 1e3, fp / tv / 1e12, to * 1e3, fp / to / 1e12, m, tsv * 1e3, ft / tsv / 1e12, ts * 1e3, ft / ts / 1e12, errL, err), flush=True)
		del K, B, X, Xv, Lv, winv, work, tw
		torch.cuda.empty_cache()

if __name__ == "__main__":
	main()
