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
res = {}
keep = None
wb = int(lib.stpy_rff_workspace_bytes(L.F32, n, d, m))
work = torch.empty(wb, dtype=torch.uint8, device=dev)
for rnd in range(4):
	for route in (1, 5, 3, 2, 0):          # 5: the streaming kernel on the bf16 matrix cores (exact three-way split; needs the workspace)
		lib.stpy_tune(9, 1 if route == 5 else route)
		torch.cuda.synchronize(); t0 = time.perf_counter()
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(x), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(out), m, 0, L.ptr(work) if route == 5 else None, wb if route == 5 else 0, L.stream_ptr()), "rff")
		torch.cuda.synchronize(); res.setdefault(route, []).append(time.perf_counter() - t0)
		if rnd == 0:
			keep = out[::4097].clone() if route == 1 else keep
			if route != 1:
				print("route %d vs streaming kernel: max abs diff %.2e (amplitude %.2e)" % (route, float((out[::4097] - keep).abs().max()), math.sqrt(2.0 / m)))
lib.stpy_tune(9, 1)
for route, name in ((1, "streaming (LDS patch)"), (5, "streaming, bf16x3 matrix cores"), (3, "streaming (direct stores)"), (2, "tile"), (0, "gemm epilogue")):
	print("%-26s %s ms" % (name, " ".join("%.2f" % (t * 1e3) for t in res[route])))
