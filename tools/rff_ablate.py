"""C5-shaped fp32 embed with phases of the kernel switched off (stpy_tune key 1 bits: 1 = no stores, 2 = no MFMA (fp32 kernels), 4 = no operand loads,
8 = no trig / staging (bf16x3 kernel: then the contraction is dead code too, i.e. the store stream alone; that kernel knows 0, 1, 8 only).   usage: python tools/rff_ablate.py [ws] [exp values...]   (ws: pass the workspace = the bf16x3 kernel)"""
import sys, time, math
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
dev = torch.device("cuda:0")
n, d, m = 262144, 64, 32768
x = torch.rand(n, d, dtype=torch.float32, device=dev)
W = torch.randn(m, d, dtype=torch.float32, device=dev) / 8
out = torch.empty(n, m, dtype=torch.float32, device=dev)
args = sys.argv[1:]
use_ws = bool(args) and args[0] == "ws"
args = args[1:] if use_ws else args
wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m)) if use_ws else 0
work = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
for exp in [int(v) for v in args] or ((0, 1, 8, 0) if use_ws else (0, 1, 2, 4, 3, 5, 6, 7, 0)):
	lib.stpy_tune(1, exp)
	ts = []
	for it in range(4):
		torch.cuda.synchronize(); t0 = time.perf_counter()
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(x), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(out), m, 0, L.ptr(work) if use_ws else None, wb, L.stream_ptr()), "rff")
		torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
	print("exp %d (%s%s%s%s): %.2f ms" % (exp, "no-store " if exp & 1 else "", "no-mfma " if exp & 2 else "", "no-load " if exp & 4 else "", "no-trig " if exp & 8 else "", min(ts) * 1e3), flush=True)
lib.stpy_tune(1, 0)
