"""Accuracy of the factorisation at BASELINE config 2's shape with a given build of the library: ||K alpha - y|| / ||y|| and the
relative error of sampled rows of L L^T against K.   usage: python tools/factor_residual.py path/to/lib.so [n] [d]"""
import ctypes, math, sys
import torch
sys.path.insert(0, ".")
from stpy_amd import _lib as L
path = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lib = ctypes.CDLL(path)
for name, (res, args) in L.SIGNATURES.items():
	if hasattr(lib, name):
		fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1234)
x = (torch.rand(n, d, generator=g, dtype=torch.float64) * 2 - 1).to(dev)
g.manual_seed(1235)
y = (torch.sin(x.cpu().sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=g, dtype=torch.float64)).to(dev).reshape(-1)
il = torch.full((d,), 1.0 / math.sqrt(d), dtype=torch.float64, device=dev)
K = torch.empty(n, n, dtype=torch.float64, device=dev)
ws = torch.empty(int(lib.stpy_gram_workspace_bytes(L.F64, n, n, d)), dtype=torch.uint8, device=dev)
assert lib.stpy_gram(0, L.F64, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01, 0, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()) == 0
K0 = K.clone()
winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=torch.float64, device=dev)
work = torch.empty(int(lib.stpy_potrf_workspace_bytes(L.F64, n, 0)), dtype=torch.uint8, device=dev)
info = torch.zeros(1, dtype=torch.int32, device=dev)
assert lib.stpy_potrf(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), 0, 0, L.ptr(info), L.stream_ptr()) == 0
torch.cuda.synchronize()
print("info", int(info.item()))
Lf = torch.tril(K)
rows = torch.arange(0, n, max(n // 64, 1), device=dev)
rec = Lf[rows] @ Lf.T
print("max |L L^T - K| / max|K| on %d sampled rows: %.3e" % (rows.numel(), float((rec - K0[rows]).abs().max() / K0.abs().max())))
full = Lf @ Lf.T - K0
eb = full.abs().reshape(n // 128, 128, n // 128, 128).amax(dim=(1, 3))
eb = torch.tril(eb)
idx = torch.nonzero(eb > 1e-11)
print("128-blocks (row, col) of L L^T - K above 1e-11: %d of %d; first %s" % (idx.shape[0], (n // 128) * (n // 128 + 1) // 2, [(int(a), int(b), float(eb[a, b])) for a, b in idx[:12].tolist()]))
if idx.shape[0]:
	a, b = idx[0].tolist()
	blk = full[a * 128:(a + 1) * 128, b * 128:(b + 1) * 128].abs()
	rr = torch.nonzero(blk.amax(dim=1) > 1e-11).reshape(-1).tolist(); cc = torch.nonzero(blk.amax(dim=0) > 1e-11).reshape(-1).tolist()
	print("   inside block (%d, %d): bad rows %s  bad cols %s" % (a, b, rr[:40], cc[:40]))
del full
# per 128-block: where is the error?
err = (rec - K0[rows]).abs().max(dim=1).values
print("worst rows:", [(int(rows[i]), float(err[i])) for i in torch.argsort(err, descending=True)[:5].tolist()])
# inverse blocks: W L_cc = I
W = winv.reshape(-1, 128, 128)
for c in (0, 1, n // 256, n // 128 - 1):
	Lcc = Lf[c * 128:(c + 1) * 128, c * 128:(c + 1) * 128]
	print("block %d: |W L - I| %.3e   cond(L_cc) %.3e" % (c, float((W[c] @ Lcc - torch.eye(128, dtype=torch.float64, device=dev)).abs().max()), float(torch.linalg.cond(Lcc))))
I128 = torch.eye(128, dtype=torch.float64, device=dev)
bad = []
for c in range(n // 128):
	Lcc = Lf[c * 128:(c + 1) * 128, c * 128:(c + 1) * 128]
	e = float((W[c] @ Lcc - I128).abs().max())
	up = float(torch.triu(W[c], 1).abs().max())
	if e > 1e-12 or up != 0.0 or not bool(torch.isfinite(W[c]).all()):
		bad.append((c, e, up))
print("W blocks with |W L - I| > 1e-12 or a non-zero upper triangle:", bad[:20], "of", n // 128)
z = torch.empty(n, dtype=torch.float64, device=dev); a = torch.empty(n, dtype=torch.float64, device=dev)
ysc = y.clone()
assert lib.stpy_trsv(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(ysc), L.ptr(z), 0, L.stream_ptr()) == 0
zsc = z.clone()
assert lib.stpy_trsv(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(zsc), L.ptr(a), 1, L.stream_ptr()) == 0
zt = torch.linalg.solve_triangular(Lf, y.reshape(-1, 1), upper=False).reshape(-1)
print("forward solve vs torch: %.3e ; first bad 128-block of z: %s" % (float((z - zt).norm() / zt.norm()), [int(b) for b in torch.nonzero((z - zt).abs().reshape(-1, 128).max(dim=1).values > 1e-9 * zt.abs().max()).reshape(-1)[:8]]))
at = torch.linalg.solve_triangular(Lf.T, zt.reshape(-1, 1), upper=True).reshape(-1)
zs2 = zt.clone(); a2 = torch.empty_like(a)
assert lib.stpy_trsv(L.F64, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(zs2), L.ptr(a2), 1, L.stream_ptr()) == 0
print("backward solve (from torch's z) vs torch: %.3e ; bad blocks %s" % (float((a2 - at).norm() / at.norm()), [int(b) for b in torch.nonzero((a2 - at).abs().reshape(-1, 128).max(dim=1).values > 1e-9 * at.abs().max()).reshape(-1)[:8]]))
r = K0 @ a - y
print("||K alpha - y|| / ||y|| = %.3e" % float(r.norm() / y.norm()))
