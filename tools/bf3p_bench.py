"""The pre-split fp32 update kernel alone (gemm_bf3p.hip) against the on-the-fly split kernel on one trailing-update shape, random and all-zero
operands, and with phases switched off (LAB build: stpy_tune key 1 bit 0 = no operand DMA in the K loop, bit 1 = no fragment reads / MFMAs;
results are wrong then).   usage: STPY_HIP_LIB=lab python tools/bf3p_bench.py [n] [k]"""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
k = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
i64, vp, i32 = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int
lib.stpy_debug_bf3_split.argtypes = [vp, i64, i64, i64, vp, i64, i64, vp]
lib.stpy_debug_bf3_split.restype = i32
lib.stpy_debug_gemm_bf3p.argtypes = [i64, i64, i64, vp, i64, i64, i64, i64, vp, i64, i32, vp]
lib.stpy_debug_gemm_bf3p.restype = i32
C = torch.zeros(n, n, dtype=torch.float32, device=dev)
planes = torch.empty(3 * n * k, dtype=torch.int16, device=dev)
flops = float(n) * n * k          # lower tiles


def timed(fn, reps=4):
	best = 1e9
	for r in range(reps):
		torch.cuda.synchronize()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		e0.record()
		fn()
		e1.record()
		torch.cuda.synchronize()
		if r:
			best = min(best, e0.elapsed_time(e1))
	return best


for data in ("random", "zeros"):
	P = (torch.rand(n, k, dtype=torch.float32, device=dev) - 0.5) if data == "random" else torch.zeros(n, k, dtype=torch.float32, device=dev)
	ts = timed(lambda: L.check(lib.stpy_debug_bf3_split(L.ptr(P), k, n, k, L.ptr(planes), k, n * k, L.stream_ptr()), "split"))
	told = timed(lambda: L.check(lib.stpy_gemm_nt(L.F32, n, n, k, L.ptr(P), k, L.ptr(P), k, L.ptr(C), n, 1, 1, L.stream_ptr()), "gemm"))
	print("%s: split pass %.3f ms (%.2f TB/s of 10 B/value);  on-the-fly kernel %.3f ms = %.1f TF/s" % (data, ts, n * k * 10 / ts / 1e9, told, flops / told / 1e9), flush=True)
	for exp in (0, 1, 2, 3):
		lib.stpy_tune(1, exp)
		t = timed(lambda: L.check(lib.stpy_debug_gemm_bf3p(n, n, k, L.ptr(planes), k, n * k, 0, 0, L.ptr(C), n, 1, L.stream_ptr()), "bf3p"))
		print("   pre-split kernel exp=%d (%s): %.3f ms = %.1f TF/s" % (exp, ["as shipped", "no DMA in the loop", "no reads / MFMAs", "neither: barriers + C tile only"][exp], t, flops / t / 1e9), flush=True)
	lib.stpy_tune(1, 0)
	C.zero_()
