"""gram + stpy_potrf a few times at one size -- target for rocprofv3 --kernel-trace (chain analysis).
usage: python tools/potrf_only.py n [nb]"""
import sys
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, ".")
from stpy_amd import _lib as L
lib = L.load()
n = int(sys.argv[1]); nb = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for kv in sys.argv[3:]:          # stpy_tune key=value pairs
	k, v = kv.split("=")
	lib.stpy_tune(int(k), int(v))
dev = torch.device("cuda:0")
x = torch.rand(n, 8, dtype=torch.float64, device=dev) * 2 - 1
il = torch.full((8,), 0.35, dtype=torch.float64, device=dev)
K = torch.empty(n, n, dtype=torch.float64, device=dev)
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, nb)), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, 8)), dtype=torch.uint8, device=dev)
for _ in range(3):
	L.check(lib.stpy_gram(0, L.F64, L.ptr(x), n, 8, L.ptr(x), n, 8, 8, None, L.ptr(il), 1.0, 0.0, 0.01, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel() * ws.element_size(), L.stream_ptr()), "gram")
	L.check(lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel() * work.element_size(), nb, 0, L.ptr(info), L.stream_ptr()), "potrf")
	torch.cuda.synchronize()
print("info", int(info.item()))
