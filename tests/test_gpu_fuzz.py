"""Randomised parity sweep (tools/fuzz_parity.py) as part of the GPU suite: 80 random (size, dimension, kernel family, hyper-parameter)
cases through the drop-in classes against the CPU oracle -- ragged and tiny sizes, tile / panel edges, sums and products of kernels,
full covariance, add_data_point; then 40 random RFF embeds (both types, biased and plain, every kernel route)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_random_parity_sweep():
	r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "80", "11", "40"], cwd=ROOT, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
	assert "all 80 cases passed" in r.stdout and "all 40 rff cases passed" in r.stdout


@pytest.mark.gpu
def test_random_gemm_sweep():
	"""tools/fuzz_gemm.py: 150 random (m, n, k, leading dimensions, mode, lower-only, type) products through stpy_gemm_nt against fp64
	matmul; NaN-poisoned operand padding, untouched output padding and untouched tiles above the diagonal"""
	r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gemm.py"), "150", "3"], cwd=ROOT, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
	assert "all 150 gemm cases passed" in r.stdout


@pytest.mark.gpu
def test_random_factor_solve_sweep():
	"""tools/fuzz_factor.py: 60 random (order, right-hand sides, panel width, flags, padding, block-solve route, type) runs of
	stpy_potrf -> stpy_trsm_right_lt -> stpy_trsv (both ways) -> stpy_logdet_quad against numpy / scipy"""
	r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_factor.py"), "60", "4"], cwd=ROOT, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
	assert "all 60 factor / solve cases passed" in r.stdout


@pytest.mark.gpu
def test_factor_stress_every_block_every_time():
	"""tools/factor_stress.py: eight factorisations of the same N = 8192 matrix, L L^T - K and W L - I checked for EVERY 128-block each
	time.  Round 4's MFMA diagonal-block kernel once produced a wrong 4-row group in ~3 of 128 diagonal blocks, only beside the
	trailing update's MFMA traffic and not in every run (two fp64 MFMAs in flight with partially overlapping destination registers);
	sampled-row checks and the small-size kernel tests all passed."""
	r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "factor_stress.py"), "8192", "8", "8"], cwd=ROOT, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0 and "STRESS ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
