"""KernelizedFeatures.fit_gp + mean_std at the bench's KF shape (N = 262 144, d = 64, m = 8192, fp32): target for rocprofv3 --kernel-trace --stats
(where the 0.1 s go: the normal equations' SYRK per slab, the split pass, the embed, the m x m factorisation, the prediction).
usage: python tools/kf_trace.py [reps]"""
import math
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from stpy_amd import RFFEmbedding
from stpy_amd.continuous_processes.kernelized_features import KernelizedFeatures

dev = torch.device("cuda:0")
n, d, m, M = 262144, 64, 8192, 4096
xk = torch.rand(n, d, generator=torch.Generator().manual_seed(1238), dtype=torch.float32).to(dev)
yk = torch.sin(xk[:, :4].sum(dim=1, keepdim=True)) + 0.1 * torch.randn(n, 1, generator=torch.Generator().manual_seed(1239), dtype=torch.float32).to(dev)
xtk = torch.rand(M, d, generator=torch.Generator().manual_seed(1240), dtype=torch.float32).to(dev)
np.random.seed(1238)
emb = RFFEmbedding(gamma=math.sqrt(d), m=m, d=d)
emb.W = emb.W.float()
kf = KernelizedFeatures(embedding=emb, m=m, s=1.0, lam=1.0, d=d)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	kf.fit_gp(xk, yk)
	torch.cuda.synchronize()
	t1 = time.perf_counter()
	mu, sd = kf.mean_std(xtk)
	torch.cuda.synchronize()
	t2 = time.perf_counter()
	print("fit %.4f s  mean_std %.4f s" % (t1 - t0, t2 - t1), flush=True)
