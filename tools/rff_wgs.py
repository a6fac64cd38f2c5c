"""BASELINE config 5 through the streaming kernel with 256 / 512 / 768 / 1024 workgroups (stpy_tune key 19): one, two, three,
four waves per SIMD.   usage: python tools/rff_wgs.py"""
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
route = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # stpy_tune key 9: 1 production, 4 = timing-only bf16-MFMA stand-in
lib.stpy_tune(9, route)
for rnd in range(4):
	for wgs in (256, 512, 768, 1024):
		lib.stpy_tune(19, wgs)
		torch.cuda.synchronize(); t0 = time.perf_counter()
		L.check(lib.stpy_rff_embed(L.F32, L.ptr(x), n, d, d, L.ptr(W), d, m, None, None, math.sqrt(2.0 / m), L.ptr(out), m, 0, None, 0, L.stream_ptr()), "rff")
		torch.cuda.synchronize(); res.setdefault(wgs, []).append(time.perf_counter() - t0)
lib.stpy_tune(19, 0)
lib.stpy_tune(9, 1)
print('route', route)
for wgs, ts in res.items():
	print("%4d workgroups: %s ms" % (wgs, " ".join("%.2f" % (t * 1e3) for t in ts)))
