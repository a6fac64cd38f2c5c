"""Test-grid builders used by every stpy script (reference: stpy/helpers/helper.py:27-59, :125-140)."""
import numpy as np
import torch


def cartesian(arrays, out=None, dtype=None):
	"""helper.py:27-59: cartesian product of 1-D arrays, first array varying slowest."""
	arrays = [np.asarray(x).reshape(-1) for x in arrays]
	if dtype is None:
		dtype = arrays[0].dtype
	mesh = np.meshgrid(*arrays, indexing="ij")
	res = np.stack([m.reshape(-1) for m in mesh], axis=1).astype(dtype)
	if out is not None:
		out[...] = res
		return out
	return res


def interval(n, d, L_infinity_ball=1, offset=None):
	"""helper.py:125-136."""
	if offset is None:
		arrays = [np.linspace(-L_infinity_ball, L_infinity_ball, n) for i in range(d)]
	else:
		arrays = [np.linspace(offset[i][0], offset[i][1], n) for i in range(d)]
	return cartesian(arrays)


def interval_torch(n, d, L_infinity_ball=1, offset=None):
	"""helper.py:139-140."""
	return torch.from_numpy(interval(n, d, L_infinity_ball=L_infinity_ball, offset=offset))
