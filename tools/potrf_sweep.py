"""Interleaved A/B of stpy_potrf policies in ONE process (same buffers, same clocks): stpy_tune key / value pairs against
the defaults, per matrix order.   usage: python tools/potrf_sweep.py "n1,n2,.." "key=v1|v2|..;key=..." [nb] [f32]
e.g.  python tools/potrf_sweep.py 8192,16384,32768 "10=0|6144|1000000;7=0|8192" """
import itertools
import sys
import time

import torch

sys.path.insert(0, ".")
from stpy_amd import _lib as L

lib = L.load()
dev = torch.device("cuda:0")


def main():
	ns = [int(v) for v in sys.argv[1].split(",")]
	axes = []
	for part in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(";")):
		k, vs = part.split("=")
		axes.append((int(k), [int(v) for v in vs.split("|")]))
	nb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
	dt = torch.float32 if len(sys.argv) > 4 and sys.argv[4] == "f32" else torch.float64
	code = L.dtype_code(dt)
	esz = 4 if dt == torch.float32 else 8
	defaults = {k: int(lib.stpy_tune_get(k)) for k, _ in axes}
	for n in ns:
		d = 16
		x = torch.rand(n, d, dtype=dt, device=dev) * 2 - 1
		il = torch.full((d,), 0.25, dtype=dt, device=dev)
		K = torch.empty(n, n, dtype=dt, device=dev)
		winv = torch.empty(int(lib.stpy_potrf_winv_elems(n)), dtype=dt, device=dev)
		work = torch.empty(max(int(lib.stpy_potrf_workspace_bytes(code, n, nb)), 2 * n * 4096 * esz), dtype=torch.uint8, device=dev)          # (widest panel any policy picks)
		info = torch.zeros(1, dtype=torch.int32, device=dev)
		ws = torch.empty(int(lib.stpy_gram_workspace_bytes(code, n, n, d)), dtype=torch.uint8, device=dev)

		def gram():
			L.check(lib.stpy_gram(0, code, L.ptr(x), n, d, L.ptr(x), n, d, d, None, L.ptr(il), 1.0, 0.0, 0.01 if esz == 8 else 0.1, 1, 0, L.ptr(K), n, L.ptr(ws), ws.numel(), L.stream_ptr()), "gram")

		def potrf():
			L.check(lib.stpy_potrf(code, n, L.ptr(K), n, L.ptr(winv), winv.numel(), L.ptr(work), work.numel(), nb, 0, L.ptr(info), L.stream_ptr()), "potrf")
		combos = list(itertools.product(*[vs for _, vs in axes])) or [()]
		best = {c: 1e9 for c in combos}
		for rnd in range(4):
			for c in combos:
				for (k, _), v in zip(axes, c):
					lib.stpy_tune(k, v)
				gram()
				torch.cuda.synchronize()
				e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
				e0.record()
				potrf()
				e1.record()
				torch.cuda.synchronize()
				if rnd > 0:
					best[c] = min(best[c], e0.elapsed_time(e1))
		for k, v in defaults.items():
			lib.stpy_tune(k, v)
		assert int(info.item()) == 0
		for c in combos:
			tag = " ".join("k%d=%d" % (k, v) for (k, _), v in zip(axes, c)) or "defaults"
			print("potrf n=%6d nb=%d  %-28s %9.3f ms  %6.1f TF/s" % (n, nb, tag, best[c], n ** 3 / 3.0 / best[c] / 1e9), flush=True)
		del K, winv, work, ws
		torch.cuda.empty_cache()


if __name__ == "__main__":
	main()
