"""
Host AddressSanitizer pass over the argument-checking layer of the C ABI (sanitizers run on the CPU build only
-- GPU ASan / XNACK are not available on the pool, and the GPU runner refuses any snapshot that carries a sanitizer build line, which is
why this script is listed in .gpurunignore and lives under tools/: it never travels to a GPU box).  The library is rebuilt with the
address sanitizer on the HOST side (device code untouched), loaded in a child interpreter under the ASan runtime, and every entry point
of include/stpy_hip.h is called with arguments it must refuse -- null pointers, impossible dimensions, undersized workspaces, unknown
dtypes / modes -- or with an empty problem it must accept without looking at any pointer.  No GPU is needed: every such call returns
before the first HIP call.  Exit code 0 and the line ASAN_ABI_OK <calls> on success.
usage: python tools/abi_asan_check.py [build dir]
"""
import os
import subprocess
import sys
import tempfile
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stpy_amd", "csrc")
SRCS = ["api", "gemm", "gemm_bf3p", "potrf", "solve", "gram", "rff", "reduce"]

CHILD = r'''
import ctypes, sys
sys.path.insert(0, %(root)r)
from stpy_amd import _lib as L          # signatures only: the product library is NOT loaded
lib = ctypes.CDLL(%(so)r)
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
P = ctypes.c_void_p(0x1000)             # a non-null pointer that must never be dereferenced by a refused call
N = None
calls = 0
def neg(rc, what):
    global calls
    calls += 1
    assert rc < 0, (what, rc)
    assert lib.stpy_last_error_string(), what
def zero(rc, what):
    global calls
    calls += 1
    assert rc == 0, (what, rc)
assert lib.stpy_version().startswith(b"stpy_hip")
# ---- Gram
neg(lib.stpy_gram(0, 0, N, 4, 4, P, 4, 4, 2, N, P, 1.0, 0.0, 0.0, 0, 0, P, 4, N, 0, N), "gram null a")
neg(lib.stpy_gram(0, 0, P, 4, 4, P, 4, 4, 0, N, P, 1.0, 0.0, 0.0, 0, 0, P, 4, N, 0, N), "gram d = 0")
neg(lib.stpy_gram(0, 0, P, 4, 1, P, 4, 4, 2, N, P, 1.0, 0.0, 0.0, 0, 0, P, 4, N, 0, N), "gram lda < d")
neg(lib.stpy_gram(0, 0, P, 4, 4, P, 4, 4, 2, N, P, 1.0, 0.0, 0.0, 0, 0, P, 3, N, 0, N), "gram ldo < n")
neg(lib.stpy_gram(0, 0, P, 256, 4, P, 256, 4, 2, N, P, 1.0, 0.0, 0.0, 0, 0, P, 256, P, 16, N), "gram undersized workspace")
zero(lib.stpy_gram(0, 0, N, 0, 4, N, 4, 4, 2, N, N, 1.0, 0.0, 0.0, 0, 0, N, 4, N, 0, N), "gram empty")
assert lib.stpy_gram_workspace_bytes(0, 256, 256, 4) > 0
neg(lib.stpy_gram_diag(0, 0, N, 4, 4, 2, N, P, 1.0, 0.0, 0, P, N), "gram_diag null")
zero(lib.stpy_gram_diag(0, 0, N, 0, 4, 2, N, N, 1.0, 0.0, 0, N, N), "gram_diag empty")
# ---- factorisation and solves
assert lib.stpy_potrf_winv_elems(130) == 2 * 128 * 128 and lib.stpy_potrf_workspace_bytes(0, 1024, 0) > 0
neg(lib.stpy_potrf(0, 256, N, 256, P, 1 << 20, P, 1 << 30, 0, 0, P, N), "potrf null A")
neg(lib.stpy_potrf(0, 256, P, 100, P, 1 << 20, P, 1 << 30, 0, 0, P, N), "potrf lda < n")
neg(lib.stpy_potrf(0, 256, P, 256, P, 10, P, 1 << 30, 0, 0, P, N), "potrf winv too small")
neg(lib.stpy_potrf(0, 256, P, 256, P, 1 << 20, P, 10, 0, 0, P, N), "potrf workspace too small")
neg(lib.stpy_potrf(7, 256, P, 256, P, 1 << 20, P, 1 << 30, 0, 0, P, N), "potrf unknown dtype")
neg(lib.stpy_trsm_right_lt(0, 8, 256, N, 256, P, 1 << 20, P, 256, 0, 0, N, 0, N), "trsm null L")
neg(lib.stpy_trsm_right_lt(0, 8, 256, P, 256, P, 10, P, 256, 0, 0, N, 0, N), "trsm winv too small")
neg(lib.stpy_trsm_right_lt(0, 8, 256, P, 200, P, 1 << 20, P, 256, 0, 0, N, 0, N), "trsm ldl < n")
neg(lib.stpy_potri(0, 256, N, 256, P, 1 << 20, P, 256, P, 1 << 30, N), "potri null")
neg(lib.stpy_potri(0, 256, P, 256, P, 1 << 20, P, 256, P, 8, N), "potri workspace too small")
neg(lib.stpy_lml_weight(0, 0, N, 256, 4, 2, N, P, 1.0, 1.0, P, P, 256, P, 256, P, 1 << 30, N), "lml_weight null x")
neg(lib.stpy_trsv(0, 256, P, 256, P, 1 << 20, P, P, 0, N), "trsv y aliases out")
neg(lib.stpy_trsv(0, 256, P, 256, P, 10, P, ctypes.c_void_p(0x2000), 0, N), "trsv winv too small")
neg(lib.stpy_predict(0, 4, 8, N, 8, P, P, P, P, 0, N), "predict null X")
neg(lib.stpy_predict(0, 4, 8, P, 4, P, P, P, P, 0, N), "predict ldx < n")
zero(lib.stpy_predict(0, 0, 8, N, 8, N, N, N, N, 0, N), "predict empty")
neg(lib.stpy_predict_finish(0, 4, P, N, N, 1.0, P, 0, N), "predict_finish sigma without sumsq")
zero(lib.stpy_predict_finish(0, 0, N, N, N, 1.0, N, 0, N), "predict_finish empty")
neg(lib.stpy_combine(0, 4, 4, N, 4, P, 4, 0, 0.0, N), "combine null")
neg(lib.stpy_combine(0, 4, 4, P, 4, P, 4, 9, 0.0, N), "combine unknown op")
neg(lib.stpy_combine(0, 4, 4, P, 2, P, 4, 0, 0.0, N), "combine ldo < n")
neg(lib.stpy_logdet_quad(0, 4, N, 4, P, P, N), "logdet null")
neg(lib.stpy_logdet_quad(0, 4, P, 2, P, P, N), "logdet ldl < n")
# ---- products
neg(lib.stpy_gemm_nt(0, 4, 4, 4, N, 4, P, 4, P, 4, 0, 0, N), "gemm null A")
neg(lib.stpy_gemm_nt(0, 4, 4, 4, P, 2, P, 4, P, 4, 0, 0, N), "gemm lda < k")
neg(lib.stpy_gemm_nt(0, 256, 256, 16, P, 16, P, 16, P, 1 << 25, 0, 0, N), "gemm ldc >= 2^25")
neg(lib.stpy_gemm_nt(9, 4, 4, 4, P, 4, P, 4, P, 4, 0, 0, N), "gemm unknown dtype")
neg(lib.stpy_syrk(1, 4096, 64, N, 64, P, 4096, 0, N, 0, N), "syrk null A")
neg(lib.stpy_syrk(1, 4096, 64, P, 32, P, 4096, 0, N, 0, N), "syrk lda < k")
neg(lib.stpy_syrk(1, 4096, 64, P, 64, P, 4096, 3, N, 0, N), "syrk unknown mode")
neg(lib.stpy_syrk(1, 4096, 64, P, 64, P, 4096, 0, P, 16, N), "syrk undersized workspace")
zero(lib.stpy_syrk(1, 0, 64, N, 64, N, 1, 0, N, 0, N), "syrk empty")
assert lib.stpy_syrk_workspace_bytes(1, 4096, 64) == 3 * 4096 * 64 * 2 and lib.stpy_syrk_workspace_bytes(0, 4096, 64) == 0 and lib.stpy_syrk_workspace_bytes(1, 1000, 64) == 0
assert lib.stpy_gemm_nt_splitk_passes(128, 128, 65536) >= 1
neg(lib.stpy_gemm_nt_splitk(0, 128, 128, 4096, P, 4096, P, 4096, P, 128, 0, 4, P, 8, N), "splitk workspace too small")
neg(lib.stpy_gemm_nt_bc(0, 256, 256, 128, P, 128, P, 128, P, 256, 1, 100, 1, 1, 0, 0, 0, 0, N), "gemm_bc block not a multiple of 128")
neg(lib.stpy_symmetrize_lower(0, 4, N, 4, N), "symmetrize null")
neg(lib.stpy_tril(0, 4, N, 4, N), "tril null")
neg(lib.stpy_tril(0, 4, P, 2, N), "tril lda < n")
zero(lib.stpy_tril(0, 0, N, 0, N), "tril empty")
neg(lib.stpy_trace_dot(0, 4, P, 4, N, N, N, N), "trace_dot null out")
neg(lib.stpy_trace_dot(0, 4, P, 2, N, N, P, N), "trace_dot lda < n")
neg(lib.stpy_scaled_points_t(0, N, 4, 4, 2, N, P, P, 4, 1, N), "scaled_points null x")
neg(lib.stpy_scaled_points_t(0, P, 4, 4, 2, N, P, P, 3, 1, N), "scaled_points ldo < n")
zero(lib.stpy_scaled_points_t(0, N, 0, 4, 2, N, N, N, 4, 1, N), "scaled_points empty")
neg(lib.stpy_lml_grad_reduce(0, P, 4, 4, 2, N, P, P, 2, N, P, N), "lml_grad_reduce ldp < d + 1")
neg(lib.stpy_lml_grad_reduce(0, N, 4, 4, 2, N, P, P, 3, N, P, N), "lml_grad_reduce null x")
neg(lib.stpy_lml_grad_cov_reduce(0, P, 4, 4, 2, N, N, 2, 2, P, 3, P, N), "lml_grad_cov_reduce null z")
neg(lib.stpy_lml_grad_cov_reduce(0, P, 4, 4, 2, N, P, 2, 2, P, 2, P, N), "lml_grad_cov_reduce ldp < p + 1")
# ---- RFF
assert lib.stpy_rff_workspace_bytes(1, 262144, 64, 32768) > 0 and lib.stpy_rff_workspace_bytes(0, 100, 5, 64) == 0
neg(lib.stpy_rff_embed(1, N, 16, 4, 4, P, 4, 8, N, N, 1.0, P, 8, 0, N, 0, N), "rff null x")
neg(lib.stpy_rff_embed(1, P, 16, 2, 4, P, 4, 8, N, N, 1.0, P, 8, 0, N, 0, N), "rff ldx < d")
neg(lib.stpy_rff_embed(1, P, 262144, 64, 64, P, 64, 32768, N, N, 1.0, P, 32768, 0, P, 16, N), "rff undersized workspace")
zero(lib.stpy_rff_embed(1, N, 0, 4, 4, N, 4, 8, N, N, 1.0, N, 8, 0, N, 0, N), "rff empty")
# ---- switches and profiler (no device needed)
for key in (5, 8, 9, 16, 17, 26, 28, 30, 32):
    v = lib.stpy_tune_get(key); assert v >= 0; lib.stpy_tune(key, v)
assert lib.stpy_tune_get(12345) == -1
lib.stpy_tune(12345, 1)
lib.stpy_profile_enable(0)
print("ASAN_ABI_OK", calls)
'''




def main():
	import torch
	if torch.cuda.is_available():
		print("ASAN_ABI_SKIPPED: placeholder pointers are only safe where no call can reach a device")
		return 0
	rt = subprocess.run(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
	if not rt or not os.path.exists(rt):
		print("ASAN_ABI_SKIPPED: no shared ASan runtime in this toolchain")
		return 0
	tmp = sys.argv[1] if len(sys.argv) > 1 else tempfile.mkdtemp(prefix="stpy_asan_")
	flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-comment", "-fvisibility=hidden", '-DSTPY_SRC_HASH="asan"',
			 "-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"]
	procs = [subprocess.Popen(["hipcc"] + flags + ["-c", os.path.join(CSRC, s + ".hip"), "-o", os.path.join(tmp, s + ".o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
			 for s in SRCS]
	for p in procs:
		out, _ = p.communicate()
		if p.returncode != 0:
			print(out[-3000:])
			return 1
	so = os.path.join(tmp, "libstpy_hip_asan.so")
	subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
					"-o", so] + [os.path.join(tmp, s + ".o") for s in SRCS], check=True, capture_output=True)
	env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1")
	r = subprocess.run([sys.executable, "-c", textwrap.dedent(CHILD) % {"root": ROOT, "so": so}], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
	sys.stdout.write(r.stdout)
	sys.stderr.write(r.stderr[-4000:])
	return 0 if (r.returncode == 0 and "ASAN_ABI_OK" in r.stdout and "AddressSanitizer" not in r.stderr) else 1


if __name__ == "__main__":
	sys.exit(main())
