"""
ctypes binding of libstpy_hip.so (include/stpy_hip.h).

The shared library is built in-tree by ``__graft_entry__.build()`` / ``make -C stpy_amd/csrc``.
There is no CPU fallback: if the library is missing, or no ROCm device is visible, the product
path raises -- a silently different code path would void every parity claim.
PyTorch is used for device memory, streams and (multi-GPU) torch.distributed only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# STPY_HIP_LIB=lab selects the lab build (make -C stpy_amd/csrc EXPERIMENTS=1: the product kernels plus the experiment knobs and the
# measured-and-dropped variants tools/ compares against); unset = the product library.  Same C ABI either way.
LIB_PATH = os.path.join(_HERE, "libstpy_hip_lab.so" if os.environ.get("STPY_HIP_LIB", "") == "lab" else "libstpy_hip.so")

F64, F32 = 0, 1
K_SE, K_MATERN12, K_MATERN32, K_MATERN52, K_LINEAR, K_POLY = 0, 1, 2, 3, 4, 5
OUT_SET, OUT_ADD, OUT_MUL = 0, 1, 2
IB = 128
FLAG_BESIDE_UPDATE = 1

_c = ctypes
_vp, _i64, _i32, _dbl = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_double

# name -> (restype, argtypes); mirrors include/stpy_hip.h one to one
SIGNATURES = {
	"stpy_version": (_c.c_char_p, []),
	"stpy_last_error_string": (_c.c_char_p, []),
	"stpy_gram": (_i32, [_i32, _i32, _vp, _i64, _i64, _vp, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _dbl, _i32, _i32, _vp, _i64, _vp, _i64, _vp]),
	"stpy_gram_workspace_bytes": (_i64, [_i32, _i64, _i64, _i32]),
	"stpy_gram_diag": (_i32, [_i32, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _i32, _vp, _vp]),
	"stpy_potrf_workspace_bytes": (_i64, [_i32, _i64, _i32]),
	"stpy_potrf_winv_elems": (_i64, [_i64]),
	"stpy_potrf": (_i32, [_i32, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _vp]),
	"stpy_trsm_workspace_bytes": (_i64, [_i32, _i64, _i64, _i32]),
	"stpy_trsm_right_lt": (_i32, [_i32, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
	"stpy_potri": (_i32, [_i32, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
	"stpy_lml_weight": (_i32, [_i32, _i32, _vp, _i64, _i64, _i32, _vp, _vp, _dbl, _dbl, _vp, _vp, _i64, _vp, _i64, _vp, _i64, _vp]),
	"stpy_predict_finish": (_i32, [_i32, _i64, _vp, _vp, _vp, _dbl, _vp, _i32, _vp]),
	"stpy_combine": (_i32, [_i32, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _dbl, _vp]),
	"stpy_trsv": (_i32, [_i32, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp]),
	"stpy_predict": (_i32, [_i32, _i64, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
	"stpy_logdet_quad": (_i32, [_i32, _i64, _vp, _i64, _vp, _vp, _vp]),
	"stpy_gemm_nt": (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
	"stpy_syrk_workspace_bytes": (_i64, [_i32, _i64, _i64]),
	"stpy_syrk": (_i32, [_i32, _i64, _i64, _vp, _i64, _vp, _i64, _i32, _vp, _i64, _vp]),
	"stpy_gemm_nt_splitk_passes": (_i32, [_i64, _i64, _i64]),
	"stpy_gemm_nt_splitk": (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
	"stpy_gemm_nt_bc": (_i32, [_i32, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
	"stpy_symmetrize_lower": (_i32, [_i32, _i64, _vp, _i64, _vp]),
	"stpy_tril": (_i32, [_i32, _i64, _vp, _i64, _vp]),
	"stpy_trace_dot": (_i32, [_i32, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
	"stpy_scaled_points_t": (_i32, [_i32, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _i32, _vp]),
	"stpy_lml_grad_reduce": (_i32, [_i32, _vp, _i64, _i64, _i32, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
	"stpy_lml_grad_cov_reduce": (_i32, [_i32, _vp, _i64, _i64, _i32, _vp, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
	"stpy_rff_workspace_bytes": (_i64, [_i32, _i64, _i32, _i64]),
	"stpy_rff_embed": (_i32, [_i32, _vp, _i64, _i64, _i32, _vp, _i64, _i64, _vp, _vp, _dbl, _vp, _i64, _i32, _vp, _i64, _vp]),
	"stpy_profile_enable": (None, [_i32]),
	"stpy_async_status": (_i32, [_vp]),
	"stpy_profile_read_union": (_i32, [_i32, _c.POINTER(_dbl), _c.POINTER(_dbl), _c.POINTER(_i64)]),
	"stpy_tune": (None, [_i32, _i32]),
	"stpy_tune_get": (_i32, [_i32]),
	"stpy_profile_read": (_i32, [_i32, _c.POINTER(_dbl), _c.POINTER(_dbl), _c.POINTER(_i64)]),
}

_lib = None


class StpyHipError(RuntimeError):
	pass


def load():
	"""Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise StpyHipError(
			"libstpy_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
			"or `make -C stpy_amd/csrc`. stpy_amd has no CPU fallback." % LIB_PATH)
	lib = ctypes.CDLL(LIB_PATH)
	for name, (res, args) in SIGNATURES.items():
		fn = getattr(lib, name)          # AttributeError here = header/library mismatch
		fn.restype = res
		fn.argtypes = args
	_lib = lib
	return lib


def check(rc, what):
	if rc != 0:
		msg = load().stpy_last_error_string().decode("utf-8", "replace")
		raise StpyHipError("%s failed (rc=%d): %s" % (what, rc, msg))


def check_async(what, stream=None):
	"""Read (and clear) the sticky device error word of the calling stream -- waits for the stream.  Raises StpyHipError when a
	hand-off wait of the one-launch vector solve gave up (its output is NaN from the affected block on)."""
	rc = load().stpy_async_status(stream_ptr() if stream is None else stream)
	if rc != 0:
		raise StpyHipError("%s: device-side failure reported by stpy_async_status (code %d%s)" % (
			what, rc, ": a hand-off wait of the one-launch vector solve timed out, its result is NaN" if rc == 1 else ""))


def device():
	"""The ROCm device this process computes on (one process per GPU: LOCAL_RANK picks it)."""
	if not torch.cuda.is_available():
		raise StpyHipError("stpy_amd needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path.")
	return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
	return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt):
	if dt == torch.float64:
		return F64
	if dt == torch.float32:
		return F32
	raise StpyHipError("unsupported dtype %s (float64 or float32)" % dt)


def ptr(t):
	return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def ld(t):
	"""Leading dimension (elements) of a row-major 2-D tensor.  A one-row tensor reports whatever stride(0) its history left behind
	(``x.T.contiguous()`` of an (n, 1) tensor keeps stride 1): there the row length is the only valid answer."""
	return t.stride(0) if t.shape[0] > 1 else max(int(t.shape[1]), 1)


def to_device(t, dtype=None):
	"""Caller tensor (CPU or GPU, torch or numpy) -> contiguous 2-D/1-D tensor on this process's GPU."""
	if not torch.is_tensor(t):
		t = torch.as_tensor(t)
	dev = device()
	if dtype is None:
		dtype = t.dtype if t.dtype in (torch.float32, torch.float64) else torch.float64
	t = t.to(device=dev, dtype=dtype)
	if t.dim() == 2 and t.stride(1) != 1:
		t = t.contiguous()
	elif t.dim() != 2:
		t = t.contiguous()
	return t


def like_input(result, ref):
	"""Results live where the caller's inputs live (CPU in -> CPU out), as in the reference."""
	if torch.is_tensor(ref) and ref.is_cuda:
		return result
	return result.cpu()
