"""fp32 factorisation with each panel split once into bf16 planes (route key 32 = 1, gemm_bf3p.hip) against the on-the-fly split of every
tile (32 = 0): time, and the two factors compared bit for bit.   usage: python tools/f32_presplit_ab.py [n1,n2,...] [diag]"""
import sys

import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
lib.stpy_tune(30, 0)          # the few-tile sliver route off: route 0 is then the on-the-fly bf16 split at every size (the bit-for-bit comparison)
dev = torch.device("cuda:0")
ns = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "8192,16384,32768,65536").split(",")]
DIAG = len(sys.argv) > 2 and sys.argv[2] == "diag"          # K = 4 I: every panel is zero below its diagonal block -> all-zero operands in the updates (clock / power probe)
dt, code = torch.float32, L.F32
for n in ns:
	d = 16
	torch.manual_seed(n)
	x = torch.rand(n, d, dtype=dt, device=dev) * 2 - 1
	il = torch.full((d,), 0.25, dtype=dt, device=dev)
	K = torch.empty(n, n, dtype=dt, device=dev)
	winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=dt, device=dev)
	work = torch.empty(int(lib.stpy_potrf_workspace_bytes(code, n, 0)), dtype=torch.uint8, device=dev)
	info = torch.zeros(1, dtype=torch.int32, device=dev)
	ws = torch.empty(int(lib.stpy_gram_workspace_bytes(code, n, n, d)), dtype=torch.uint8, device=dev)

	def run(route):
		lib.stpy_tune(32, route)
		best = 1e9
		for rnd in range(3):
			if DIAG:
				K.zero_()
				K.diagonal().fill_(4.0)
			else:
				L.check(lib.stpy_gram(0, code, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.1, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")
			torch.cuda.synchronize()
			e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			e0.record()
			L.check(lib.stpy_potrf(code, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()), "potrf")
			e1.record()
			torch.cuda.synchronize()
			L.check_async("potrf")
			assert int(info.item()) == 0, int(info.item())
			if rnd:
				best = min(best, e0.elapsed_time(e1))
		return best
	t0 = run(0)
	ref = torch.tril(K).clone() if n <= 32768 else torch.tril(K[:, :8192]).clone()
	t1 = run(1)
	got = torch.tril(K) if n <= 32768 else torch.tril(K[:, :8192])
	same = bool(torch.equal(ref, got))
	md = float((ref - got).abs().max())
	print("potrf f32 n=%6d  on-the-fly %9.3f ms (%5.1f TF/s)   pre-split %9.3f ms (%5.1f TF/s)   bit-identical %s  (max diff %.3g)"
	      % (n, t0, n ** 3 / 3.0 / t0 / 1e9, t1, n ** 3 / 3.0 / t1 / 1e9, same, md), flush=True)
	del K, winv, work, ws, ref, got
	torch.cuda.empty_cache()
lib.stpy_tune(32, 1)
