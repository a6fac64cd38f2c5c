"""Run time of the diagonal-block kernel INSIDE a factorisation (start -> end stamps taken by the kernel itself) against the
duration the kernel trace reports for it (dispatch -> end): the difference is the time the workgroup waited for a CU slot.
Diagnostic build only (tools/_ab/libstpy_hip_stamps.so).  usage: python tools/potf2_stamps.py n"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ab", "libstpy_hip_stamps.so")
lib = L.load()
lib.stpy_debug_set_potf2_buffer.argtypes = [ctypes.c_void_p]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda:0")
x = torch.rand(n, 8, dtype=torch.float64, device=dev) * 2 - 1
il = torch.full((8,), 0.35, dtype=torch.float64, device=dev)
K = torch.empty(n, n, dtype=torch.float64, device=dev)
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, 0)), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, 8)), dtype=torch.uint8, device=dev)
buf = torch.zeros(2 + 8 * 1000 + 8, dtype=torch.int64, device=dev)
OTHERS = len(sys.argv) > 2 and sys.argv[2] == "others"
lib.stpy_debug_set_potf2_buffer(ctypes.c_void_p(buf.data_ptr()))
for rep in range(3):
	buf.zero_()
	if OTHERS:
		buf[1000 * 8 + 2] = 1
	L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 8, L.ptr(x), n, 8, 8, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	e0.record()
	L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
	e1.record()
	torch.cuda.synchronize()
	b = buf.cpu().numpy()
	cnt = int(b[0])
	t = b[2:2 + 8 * cnt].reshape(-1, 8)[:, :5].astype(np.float64) / 100.0
	t = t[np.argsort(t[:, 0])]
	run = t[:, 4] - t[:, 0]
	ph = np.diff(t, axis=1)
	q = cnt // 4
	print("potrf n=%d: %.2f ms; %d diagonal blocks; in-kernel run time: mean %.1f us, first quarter of the factorisation %.1f, last quarter %.1f, max %.1f"
		  % (n, e0.elapsed_time(e1), cnt, run.mean(), run[:q].mean(), run[-q:].mean(), run.max()), flush=True)
	ex = b[2:2 + 8 * cnt].reshape(-1, 8)[:, 5:8].astype(np.float64) / 100.0
	if ex.any():
		if OTHERS:
			print("     step 3 on wave 0: trailing updates %.2f us, write-back share %.2f us, inverse block %.2f us" % tuple(ex.mean(axis=0)), flush=True)
		else:
			print("     step 3 on the critical wave: panel rows %.2f us, update + factor of its sub-block %.2f us, waiting for the other waves %.2f us" % tuple(ex.mean(axis=0)), flush=True)
	for i, name in enumerate(["load block -> LDS", "factor (8 sub-block steps)", "write-back L, diag W", "triangular inverse"]):
		print("     %-28s first quarter %6.1f us   last quarter %6.1f us" % (name, ph[:q, i].mean(), ph[-q:, i].mean()), flush=True)
