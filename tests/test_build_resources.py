"""
Build-level guards that need no GPU: the C-ABI library exports every symbol of include/stpy_hip.h,
and the MFMA GEMM instantiations keep their accumulators in registers (no scratch, no VGPR
spills) -- a rolled loop over the accumulator array once sent them all to scratch and made one
instantiation 5x slower without failing any numerical test.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stpy_amd", "csrc")


def test_header_symbols_exported():
	from stpy_amd import _lib
	lib = _lib.load()
	header = open(os.path.join(ROOT, "include", "stpy_hip.h")).read()
	declared = set(re.findall(r"\b(stpy_[a-z0-9_]+)\s*\(", header))
	assert declared, "no declarations parsed"
	for name in declared:
		assert hasattr(lib, name), "libstpy_hip.so lacks %s" % name
	assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
	assert lib.stpy_version().startswith(b"stpy_hip")
	# ... and NOTHING else: the dynamic symbol table of the product library is the header (no internal C++ symbols, no debug
	# hooks, no kernel handles -- csrc/exports.map), and the experiment knobs are not in it (lab build only)
	out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
	exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
	assert exported == declared, sorted(exported ^ declared)
	if not _lib.LIB_PATH.endswith("_lab.so"):
		assert b"lab" not in lib.stpy_version()
		for key in (0, 1, 2, 6, 11, 12, 18, 20, 21, 22):
			assert lib.stpy_tune_get(key) == -1, "experiment knob %d is compiled into the product library" % key
		for key in (5, 8, 9, 16, 17, 26, 28):
			assert lib.stpy_tune_get(key) >= 0


def test_no_cpu_fallback_without_gpu():
	import torch
	if torch.cuda.is_available():
		pytest.skip("GPU present")
	import stpy_amd
	from stpy_amd._lib import StpyHipError
	with pytest.raises(StpyHipError):
		stpy_amd.KernelFunction(d=2).kernel(torch.zeros(3, 2).double(), torch.zeros(3, 2).double())
	with pytest.raises(StpyHipError):
		stpy_amd.GaussianProcess(d=2).fit_gp(torch.zeros(3, 2).double(), torch.zeros(3, 1).double())


def test_product_does_not_import_oracle():
	for dirpath, _, files in os.walk(os.path.join(ROOT, "stpy_amd")):
		for f in files:
			if f.endswith(".py"):
				src = open(os.path.join(dirpath, f)).read()
				assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), "%s imports the oracle" % f
				assert "/root/reference" not in src


@pytest.mark.parametrize("src", ["gemm.hip", "potrf.hip"])
def test_mfma_kernels_stay_in_registers(src, tmp_path):
	out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c", os.path.join(CSRC, src),
						  "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True).stderr
	blocks = re.split(r"remark: Function Name: ", out)[1:]
	seen = 0
	for b in blocks:
		name = b.split()[0]
		if "sliver_kernel" in name:
			# kernels meant to run BESIDE two trailing-update workgroups of a CU: what those leave over is 512 - 2 * 224 = 64
			# VGPRs per SIMD lane and 160 - 2 * 32 = 96 KiB of LDS (static part; the dynamic part is checked at the launch site)
			tot = int(re.search(r"\bVGPRs: (\d+)", b).group(1)) + int(re.search(r"\bAGPRs: (\d+)", b).group(1))
			assert tot <= 64, (name, tot)
			assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)) <= 96 * 1024
			assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) <= (0 if "gemm_nt" in name else 64), name
			seen += 1
			continue
		if "potf2_trtri_flow_kernel" in name:
			# round 4's fp64 diagonal-block kernel: same 128-VGPR cap as the kernel it replaces (two of its waves per SIMD beside ONE
			# update workgroup); a few values of the one-off load phase and of the per-wave inverse cases go to scratch (136 B/lane
			# when this was written), none of them in the critical wave's factor code -- bounded so that it cannot grow unnoticed
			assert int(re.search(r"\bVGPRs: (\d+)", b).group(1)) <= 128 and int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) <= 192, name
			seen += 1
			continue
		if "gemm_nt_kernel" not in name and "potf2_trtri_mfma_kernel" not in name and "gemm_nt_dtv_kernel" not in name and "gemm_nt_k128_kernel" not in name:
			continue
		seen += 1
		scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
		vspill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
		vgprs = int(re.search(r"\bVGPRs: (\d+)", b).group(1))
		if "gemm_nt_kernel" in name:
			# plain / accumulate instantiations (EPI = 0: the Cholesky and solve work-horses) must be spill
			# free; the fused-epilogue instantiations (EPI = 2, 3, 4) may spill epilogue temporaries AFTER
			# the K loop, but not so much that the accumulators themselves are in scratch (>= 512 B)
			if "ELi0EEEv" in name:
				assert vspill == 0 and scratch == 0, (name, scratch, vspill)
			else:
				assert scratch < 400, (name, scratch, vspill)
			assert vgprs <= 256
		elif "gemm_nt_k128_kernel" in name:
			assert vspill == 0 and scratch == 0 and vgprs <= 128, (name, vgprs, scratch)
		elif "gemm_nt_dtv_kernel" in name:
			# direct-to-VGPR GEMM: a spill there also means hipcc reloads before the loop and waits for them inside it,
			# which drains the hand-counted load queue
			assert vspill == 0 and scratch == 0, (name, scratch, vspill)
			# 512 - 232 = 280 registers per SIMD must stay for the diagonal-block kernel's two waves; and TWO workgroups of this
			# kernel must leave 64 VGPRs for a "sliver" workgroup beside them: allocation granule 8, so at most 224
			assert vgprs <= 224
		else:
			# diagonal-block kernel: capped at 128 VGPRs so that it fits beside one GEMM workgroup (see gemm.hip); a few
			# loop-invariant addresses may go to scratch, the row / accumulator arrays may not
			assert vgprs <= 136 and scratch <= 64, (name, vgprs, scratch, vspill)
	assert seen >= 2


def test_rff_tile_kernel_fits_three_per_cu(tmp_path):
	"""rff_tile_f32_kernel relies on three co-resident workgroups per CU: <= 53 KiB of LDS, <= 168 VGPRs, no scratch."""
	out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c", os.path.join(CSRC, "rff.hip"),
						  "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True).stderr
	allb = re.split(r"remark: Function Name: ", out)[1:]
	stream = [b for b in allb if "rff_stream_f32_kernel" in b.split()[0]]
	assert len(stream) == 1          # persistent streaming kernel: two waves per SIMD (<= 256 VGPRs), nothing in scratch
	assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", stream[0]).group(1)) == 0
	assert int(re.search(r"\bVGPRs: (\d+)", stream[0]).group(1)) <= 256
	assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", stream[0]).group(1)) >= 2
	# the bf16-matrix-core form (production instantiation <0>): two waves per SIMD, both staging patches of two workgroups inside
	# one CU's LDS; it sits AT the 256-VGPR limit -- the row-block prologue and the first two tiles may spill, the steady-state
	# tile loop may not (checked on the assembly below)
	split = [b for b in allb if "rff_stream_bf16x3_kernelILi0" in b.split()[0]]
	assert len(split) == 1
	assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", split[0]).group(1)) >= 2
	assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", split[0]).group(1)) <= 80 * 1024
	assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", split[0]).group(1)) <= 200
	asm = tmp_path / "rff.s"
	subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-S", "--cuda-device-only",
					os.path.join(CSRC, "rff.hip"), "-o", str(asm)], capture_output=True, text=True, check=True)
	src = asm.read_text()
	body = src[src.index("_ZN4stpy24rff_stream_bf16x3_kernelILi0"):]
	body = body[body.index(":\n"):body.index(".amdhsa_kernel")]
	steady = 0
	for blk in re.split(r"\n\.LBB\d+_\d+:", body):
		if blk.count("v_mfma_f32_16x16x32_bf16") == 96 and blk.count("global_store_dwordx4") == 8:          # a steady-state tile: all MFMAs + the spread stores
			steady += 1
			assert "scratch_" not in blk
	assert steady == 2
	blocks = [b for b in allb if "rff_tile_f32_kernel" in b.split()[0]]
	assert len(blocks) == 2
	for b in blocks:
		assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0
		assert int(re.search(r"\bVGPRs: (\d+)", b).group(1)) <= 168
		assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)) <= 53 * 1024
		assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1)) >= 3


def test_dtv_kernel_fragments_are_not_copied(tmp_path):
	"""gemm_nt_dtv_kernel loads A fragments with inline asm and guards them with hand-counted waits.  The compiler believes
	the asm's output is valid at once; if it ever copied a fragment register between the load and its wait, the copy
	would read stale data.  The subtracting instantiations contain no other reason for a register move inside the MFMA
	range (the overwriting ones zero their accumulators there), so: none at all."""
	asm = tmp_path / "gemm.s"
	subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-S", "--cuda-device-only",
					os.path.join(CSRC, "gemm.hip"), "-o", str(asm)], capture_output=True, text=True, check=True)
	src = asm.read_text()
	names = [n for n in re.findall(r"^(_ZN4stpy18gemm_nt_dtv_kernel\w+):", src, re.M) if "Li1EEEv" in n]          # ACC = 1: the subtracting form
	assert len(names) == 2
	for name in names:
		body = src[src.index(name + ":"):].split(".Lfunc_end")[0].split("\n")
		mf = [k for k, l in enumerate(body) if "v_mfma" in l]
		moves = [l.strip() for l in body[mf[0]:mf[-1]] if re.match(r"\s*v_(mov|accvgpr|pk_mov)", l)]
		assert not moves, (name, moves[:4])


def test_bench_always_prints_one_json_line():
	"""`python bench.py --gpus 8` on a box without eight GPUs (this container has none): the self-launched ranks fail, and the
	one line the driver reads is still there -- value null and the reason -- with a non-zero exit code."""
	import json
	import sys
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
	lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
	assert len(lines) == 1, r.stdout[-2000:]
	out = json.loads(lines[0])
	import torch
	if not torch.cuda.is_available() or torch.cuda.device_count() < 8:
		assert r.returncode != 0 and out["value"] is None and out["n_gpus"] == 8 and out["error"]


def test_round3_kernels_keep_their_occupancy(tmp_path):
	"""gemm_nt_bf3_kernel (fp32 on the bf16 matrix cores): two workgroups per CU -- <= 256 VGPRs with BOTH accumulator sets (the
	two-level accumulation) in registers, no scratch, 48 KiB of LDS.  gram_fill_f64_kernel: the squared-exponential and linear forms
	fit four workgroups per CU (<= 128 VGPRs, no scratch, <= 40 KiB of LDS); the Matern forms three."""
	out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c", os.path.join(CSRC, "gemm.hip"),
						  "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True).stderr
	blocks = {b.split()[0]: b for b in re.split(r"remark: Function Name: ", out)[1:]}
	get = lambda b, key: int(re.search(key + r": (\d+)", b).group(1))
	bf3 = [b for n, b in blocks.items() if "gemm_nt_bf3_kernel" in n]
	assert len(bf3) == 3
	for b in bf3:
		assert get(b, r"\bVGPRs") <= 256 and get(b, r"ScratchSize \[bytes/lane\]") == 0 and get(b, r"Occupancy \[waves/SIMD\]") >= 2
		assert get(b, r"LDS Size \[bytes/block\]") <= 48 * 1024
	fills = {n: b for n, b in blocks.items() if "gram_fill_f64_kernel" in n}
	assert len(fills) == 4
	for n, b in fills.items():
		fast = "ILi0E" in n or "ILi4E" in n          # STPY_K_SE = 0, STPY_K_LINEAR = 4
		assert get(b, r"Occupancy \[waves/SIMD\]") >= (4 if fast else 3), n
		assert get(b, r"ScratchSize \[bytes/lane\]") == 0, n
		assert get(b, r"LDS Size \[bytes/block\]") <= 40 * 1024, n


def test_presplit_update_kernel_resources(tmp_path):
	"""gemm_bf3p_kernel (round 4: fp32 trailing updates from a panel split once into bf16 planes): one 512-thread workgroup per CU = two
	waves per SIMD, so <= 256 VGPRs with both accumulator sets and all 24 fragments of a K step in registers, no scratch; its 144 KiB of
	LDS are dynamic (checked at the launch site), so the static size must be 0."""
	out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-c", os.path.join(CSRC, "gemm_bf3p.hip"),
						  "-o", str(tmp_path / "x.o"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, check=True).stderr
	blocks = {b.split()[0]: b for b in re.split(r"remark: Function Name: ", out)[1:]}
	get = lambda b, key: int(re.search(key + r": (\d+)", b).group(1))
	ks = [b for n, b in blocks.items() if "gemm_bf3p_kernel" in n]
	assert len(ks) == 3                  # C = A B^T, C -= A B^T, C += A B^T
	for k in ks:
		assert get(k, r"\bVGPRs") + get(k, r"\bAGPRs") <= 256 and get(k, r"ScratchSize \[bytes/lane\]") == 0 and get(k, r"VGPRs Spill") == 0
		assert get(k, r"LDS Size \[bytes/block\]") == 0
	sp = [b for n, b in blocks.items() if "bf3_split_kernel" in n]
	assert len(sp) == 1 and get(sp[0], r"ScratchSize \[bytes/lane\]") == 0


def test_diag_block_kernel_has_no_overlapping_mfma_destinations(tmp_path):
	"""potf2_trtri_mfma_kernel<double>: two v_mfma_f64_16x16x4 with a constant C (results of which only element 0 is used) must not
	be given partially overlapping destination tuples -- hipcc packs them that way when the unused elements are dead, and the
	hardware does not order the element writes of two such instructions in flight (wrong rows in ~3 of 128 diagonal blocks beside
	other MFMA traffic; tools/factor_stress.py).  The kernel pins both results whole with an asm keep-alive; this checks the ISA."""
	asm = tmp_path / "potrf.s"
	subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-comment", "-S", "--cuda-device-only",
					os.path.join(CSRC, "potrf.hip"), "-o", str(asm)], check=True, capture_output=True)
	lines = asm.read_text().splitlines()
	for sym in ("_ZN4stpy23potf2_trtri_mfma_kernelIdEE", "_ZN4stpy23potf2_trtri_flow_kernelIdEE", "_ZN4stpy23potf2_trtri_flow_kernelIfEE"):          # (the flow kernel computes in fp64 for both matrix types)
		_check_mfma_pairs(lines, sym)


def _check_mfma_pairs(lines, sym):
	start = next(i for i, l in enumerate(lines) if l.startswith(sym))
	end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
	pat = re.compile(r"v_mfma_f64_16x16x4_f64 v\[(\d+):(\d+)\], v\[\d+:\d+\], v\[\d+:\d+\], (\S+)")
	mf = []
	for ln, l in enumerate(lines[start:end]):
		m = pat.search(l)
		if m:
			mf.append((ln, int(m.group(1)), int(m.group(2)), m.group(3)))
	assert len(mf) > 20
	# "in flight together": the second issues within a few instructions of the first (an fp64 16x16x4 occupies the pipe for 64 cycles)
	body = lines[start:end]
	for (la, a0, a1, _), (lb, b0, b1, cb) in zip(mf, mf[1:]):
		# (a pair the compiler separated by wait states -- a true dependency it knows about -- is not in flight together)
		if cb == "0" and (a0, a1) != (b0, b1) and lb - la <= 12 and not any("s_nop" in l for l in body[la:lb]):
			assert b1 < a0 or b0 > a1, "MFMAs %d instructions apart write overlapping tuples v[%d:%d] / v[%d:%d]" % (lb - la, a0, a1, b0, b1)
