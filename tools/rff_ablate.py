"""C5-shaped fp32 embed with phases of the tile kernel switched off (stpy_tune key 1 bits: 1 = no stores, 2 = no MFMA, 4 = no operand loads)."""
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
for exp in [int(v) for v in sys.argv[1:]] or (0, 1, 2, 4, 3, 5, 6, 7, 0):
	lib.stpy_tune(1, exp)
	ts = []
	for it in range(4):
		torch.cuda.synchronize(); t0 = time.perf_counter()
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(x), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(out), m, 0, L.stream_ptr()), "rff")
		torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
	print("exp %d (%s%s%s): %.2f ms" % (exp, "no-store " if exp & 1 else "", "no-mfma " if exp & 2 else "", "no-load " if exp & 4 else "", min(ts) * 1e3), flush=True)
lib.stpy_tune(1, 0)
