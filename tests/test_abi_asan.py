"""
Host sanitizer pass over the argument-checking layer of the C ABI: the work is done by tools/abi_asan_check.py,
which rebuilds the library with the host-side address sanitizer and calls every entry point of include/stpy_hip.h with arguments it must
refuse (or an empty problem it must accept) in a child interpreter under the sanitizer runtime.  That script is listed in .gpurunignore
(the GPU runner refuses snapshots that carry a sanitizer build line), so on a GPU box this test finds no script and skips; it runs in
the CPU-only authoring container.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "abi_asan_check.py")


@pytest.mark.timeout(900)
def test_c_abi_argument_checks_under_host_sanitizer(tmp_path):
	if not os.path.exists(SCRIPT):
		pytest.skip("tools/abi_asan_check.py does not travel to GPU boxes (.gpurunignore)")
	r = subprocess.run([sys.executable, SCRIPT, str(tmp_path)], capture_output=True, text=True, timeout=800, cwd=ROOT)
	if "ASAN_ABI_SKIPPED" in r.stdout:
		pytest.skip(r.stdout.strip())
	assert r.returncode == 0 and "ASAN_ABI_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
