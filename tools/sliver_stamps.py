"""Where does a "sliver" panel-chain GEMM spend its time beside a trailing update?  Diagnostic build only
(tools/_ab/libstpy_hip_stamps.so = the library compiled with -DSTPY_STAMPS: s_memrealtime stamps at the phase boundaries of
gemm_nt_sliver_kernel, written to a buffer of their own).  usage: python tools/sliver_stamps.py"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libstpy_hip_stamps.so")
lib = L.load()
lib.stpy_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.stpy_debug_set_stamp_buffer.restype = None
dev = torch.device("cuda:0")

n_upd, k_upd = 16384, 256
P = torch.randn(n_upd, k_upd, dtype=torch.float64, device=dev)
C = torch.randn(n_upd, n_upd, dtype=torch.float64, device=dev)
m, n = 14336, 256
Lf = torch.eye(n, dtype=torch.float64, device=dev) * 2 + 0.01 * torch.randn(n, n, dtype=torch.float64, device=dev).tril()
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
for b in range(n // 128):
	winv[b * 128 * 128:(b + 1) * 128 * 128] = torch.linalg.inv(Lf[b * 128:(b + 1) * 128, b * 128:(b + 1) * 128]).reshape(-1)
B = torch.randn(m, n, dtype=torch.float64, device=dev)
stamps = torch.zeros(512 * 8, dtype=torch.int64, device=dev)
lib.stpy_debug_set_stamp_buffer(ctypes.c_void_p(stamps.data_ptr()))
side = torch.cuda.Stream(priority=-1)


def chain():
	with torch.cuda.stream(side):
		L.check(lib.stpy_trsm_right_lt(L.F64, m, n, L.ptr(Lf), n, L.ptr(winv), winv.numel(), L.ptr(B), n, 256, L.FLAG_BESIDE_UPDATE, None, 0,
									   ctypes.c_void_p(side.cuda_stream)), "trsm")


def update():
	L.check(lib.stpy_gemm_nt(L.F64, n_upd, n_upd, k_upd, L.ptr(P), k_upd, L.ptr(P), k_upd, L.ptr(C), n_upd, 1, 1, L.stream_ptr()), "gemm")


def report(tag):
	torch.cuda.synchronize()
	s = stamps.cpu().numpy().reshape(512, 8)[: m // 32]
	t = s[:, :5].astype(np.float64) / 100.0          # us (100 MHz)
	base = t[:, 0].min()
	ph = np.diff(t, axis=1)
	print("%s: %d workgroups; launch span %.1f us (first start -> last end)" % (tag, len(t), t[:, 4].max() - base))
	print("   start offsets (us): median %.1f  p90 %.1f  max %.1f" % (np.median(t[:, 0] - base), np.percentile(t[:, 0] - base, 90), (t[:, 0] - base).max()))
	for i, name in enumerate(["prologue (3 tiles + C tile, wait)", "K loop (8 tiles)", "drain + barrier", "stores (+ wait)"]):
		print("   %-36s median %6.2f us   p90 %6.2f   max %6.2f" % (name, np.median(ph[:, i]), np.percentile(ph[:, i], 90), ph[:, i].max()))
	cu = s[:, 5]
	print("   distinct (cu, xcc) ids: %d" % len(set(cu.tolist())))


for rep in range(2):
	stamps.zero_()
	chain()
	report("alone")
for rep in range(2):
	stamps.zero_()
	update()
	torch.cuda._sleep(200000)          # ~100 us: let the update flood the chip first
	chain()
	report("beside the update")
