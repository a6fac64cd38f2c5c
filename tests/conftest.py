import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
	sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
	config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
	with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
		return {k: f[k] for k in f.files}


def rel_err(a, b):
	a = np.asarray(a, dtype=np.float64)
	b = np.asarray(b, dtype=np.float64)
	den = np.linalg.norm(b.ravel())
	return np.linalg.norm((a - b).ravel()) / (den if den > 0 else 1.0)


@pytest.fixture(scope="session")
def gpu_device():
	import torch
	if not torch.cuda.is_available():
		pytest.skip("no GPU")
	return torch.device("cuda:0")
