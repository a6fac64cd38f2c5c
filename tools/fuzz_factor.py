"""Randomised check of stpy_potrf / stpy_trsm_right_lt / stpy_trsv / stpy_logdet_quad through the C ABI against numpy / scipy:
random orders (1 .. 2600, ragged and aligned), padded leading dimensions (padding poisoned with NaN), panel widths, flags, types and
block-solve routes (stpy_tune key 5), random numbers of right-hand sides.
usage: python tools/fuzz_factor.py [cases] [seed]"""
import sys

import numpy as np
import scipy.linalg as sla
import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
NS = [1, 2, 5, 63, 64, 127, 128, 129, 200, 255, 256, 257, 300, 511, 512, 513, 700, 1023, 1024, 1025, 1280, 1536, 2047, 2048, 2049, 2600]
for case in range(cases):
	dt = torch.float64 if rng.uniform() < 0.7 else torch.float32
	code = L.dtype_code(dt)
	n = int(rng.choice(NS)) if rng.uniform() < 0.8 else int(rng.randint(1, 2000))
	m = int(rng.choice([1, 2, 8, 9, 33, 128, 200, 513, 2048, 2100]))
	nb = int(rng.choice([0, 0, 128, 256, 512, 1024]))
	flags = int(rng.choice([0, 0, 1]))
	pad = int(rng.choice([0, 0, 2, 4, 16, 36]))
	route = int(rng.choice([0, 0, 1, 3, 4, 5]))
	desc = "case %d: %s n=%d m=%d nb=%d flags=%d pad=%d route=%d" % (case, str(dt)[6:], n, m, nb, flags, pad, route)
	G = rng.normal(size=(n, max(n // 2, 1)))
	K = G @ G.T / max(n // 2, 1) + np.eye(n) * (0.5 if dt == torch.float64 else 1.0)
	Kq = K.astype(np.float32 if dt == torch.float32 else np.float64).astype(np.float64)
	Lref = np.linalg.cholesky(Kq)
	tol = 1e-11 if dt == torch.float64 else 5e-4
	try:
		Kd = torch.full((n, n + pad), float("nan"), dtype=dt, device=dev); Kd[:, :n] = torch.from_numpy(Kq).to(dev).to(dt)
		winv = torch.empty((int(lib.stpy_potrf_winv_elems(n)),), dtype=dt, device=dev)
		work = torch.empty((max(int(lib.stpy_potrf_workspace_bytes(code, n, nb)), 1),), dtype=torch.uint8, device=dev)
		info = torch.full((1,), -5, dtype=torch.int32, device=dev)
		L.check(lib.stpy_potrf(code, n, L.ptr(Kd), n + pad, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), nb, flags, L.ptr(info), L.stream_ptr()), "potrf")
		assert int(info.item()) == 0, ("info", int(info.item()))
		Lg = np.tril(Kd[:, :n].double().cpu().numpy())
		e = np.linalg.norm(Lg - Lref) / np.linalg.norm(Lref)
		assert e < tol, ("factor", e)
		# block solve X = B L^-T
		B = rng.normal(size=(m, n))
		Bq = B.astype(np.float32 if dt == torch.float32 else np.float64).astype(np.float64)
		Bd = torch.full((m, n + pad), float("nan"), dtype=dt, device=dev); Bd[:, :n] = torch.from_numpy(Bq).to(dev).to(dt)
		lib.stpy_tune(5, route)
		wb = int(lib.stpy_trsm_workspace_bytes(code, m, n, nb))
		wk = torch.empty(max(wb, 1), dtype=torch.uint8, device=dev)
		L.check(lib.stpy_trsm_right_lt(code, m, n, L.ptr(Kd), n + pad, L.ptr(winv), winv.numel(), L.ptr(Bd), n + pad, nb, flags, L.ptr(wk) if wb else None, wb, L.stream_ptr()), "trsm")
		lib.stpy_tune(5, 0)
		Xref = sla.solve_triangular(Lref, Bq.T, lower=True).T
		e = np.linalg.norm(Bd[:, :n].double().cpu().numpy() - Xref) / np.linalg.norm(Xref)
		assert e < tol * 10, ("trsm", e)
		# vector solves
		y = rng.normal(size=n)
		yq = y.astype(np.float32 if dt == torch.float32 else np.float64).astype(np.float64)
		yd = torch.from_numpy(yq).to(dev).to(dt); zd = torch.empty(n, dtype=dt, device=dev); ad = torch.empty(n, dtype=dt, device=dev)
		L.check(lib.stpy_trsv(code, n, L.ptr(Kd), n + pad, L.ptr(winv), winv.numel(), L.ptr(yd), L.ptr(zd), 0, L.stream_ptr()), "trsv")
		zref = sla.solve_triangular(Lref, yq, lower=True)
		e = np.linalg.norm(zd.double().cpu().numpy() - zref) / np.linalg.norm(zref)
		assert e < tol * 10, ("trsv forward", e)
		zs = zd.clone()
		L.check(lib.stpy_trsv(code, n, L.ptr(Kd), n + pad, L.ptr(winv), winv.numel(), L.ptr(zs), L.ptr(ad), 1, L.stream_ptr()), "trsv")
		aref = sla.solve_triangular(Lref.T, zd.double().cpu().numpy(), lower=False)
		e = np.linalg.norm(ad.double().cpu().numpy() - aref) / np.linalg.norm(aref)
		assert e < tol * 10, ("trsv backward", e)
		assert lib.stpy_async_status(L.stream_ptr()) == 0
		out2 = torch.empty(2, dtype=dt, device=dev)
		L.check(lib.stpy_logdet_quad(code, n, L.ptr(Kd), n + pad, L.ptr(zd), L.ptr(out2), L.stream_ptr()), "logdet_quad")
		ld_ref, q_ref = float(np.log(np.diag(Lref)).sum()), float(zref @ zref)
		got = out2.double().cpu().numpy()
		assert abs(got[0] - ld_ref) < (1e-10 if dt == torch.float64 else 2e-3) * max(1.0, abs(ld_ref)) and abs(got[1] - q_ref) < tol * 100 * max(1.0, q_ref), ("logdet / quad", got, ld_ref, q_ref)
	except Exception as ex:          # noqa: BLE001
		lib.stpy_tune(5, 0)
		print("FAILED", desc, "->", type(ex).__name__, ex, flush=True)
		sys.exit(1)
	if case % 15 == 0:
		print("ok", desc, flush=True)
print("all %d factor / solve cases passed" % cases)
