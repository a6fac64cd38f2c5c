// api.hip -- the extern "C" surface of libstpy_hip.so (see include/stpy_hip.h).
#include <stdarg.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <utility>
#include <vector>

#include "common.h"

namespace stpy {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(g_err, sizeof(g_err), fmt, ap);
	va_end(ap);
}

int check_launch(const char* what)
{
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		set_error("%s: launch failed: %s", what, hipGetErrorString(e));
		return -1000 - (int)e;
	}
	return 0;
}

// ---- launch profiler -------------------------------------------------------------------------
// (the record table is shared by every host thread that launches while the profiler is on: slots are handed out and
// read under a mutex; the events themselves are recorded on the launching thread's stream)
struct ProfRec { hipEvent_t e0, e1; double flops; int tag; };
static std::mutex g_prof_mutex;
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static std::atomic<bool> g_prof_on{false};

ProfScope::ProfScope(int tag, double flops, hipStream_t st_) : slot(-1), st(st_)
{
	if (!g_prof_on.load(std::memory_order_relaxed)) return;
	hipEvent_t e0;
	{
		std::lock_guard<std::mutex> lock(g_prof_mutex);
		if (g_prof_used == g_prof.size()) {
			ProfRec r;
			if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
			g_prof.push_back(r);
		}
		slot = (int)g_prof_used++;
		g_prof[slot].flops = flops;
		g_prof[slot].tag = tag;
		e0 = g_prof[slot].e0;
	}
	(void)hipEventRecord(e0, st);
}

ProfScope::~ProfScope()
{
	if (slot < 0) return;
	hipEvent_t e1;
	{
		std::lock_guard<std::mutex> lock(g_prof_mutex);
		e1 = g_prof[slot].e1;
	}
	(void)hipEventRecord(e1, st);
}

}  // namespace stpy

using namespace stpy;

#define DISPATCH(dtype, CALL64, CALL32)                                   \
	do {                                                                  \
		if ((dtype) == STPY_F64) return CALL64;                           \
		if ((dtype) == STPY_F32) return CALL32;                           \
		set_error("unknown dtype %d (0 = float64, 1 = float32)", dtype);  \
		return -2;                                                        \
	} while (0)

extern "C" {

#ifndef STPY_SRC_HASH
#define STPY_SRC_HASH "unhashed"
#endif
// "stpy_hip <version> (gfx950) src <hash of the sources this build was compiled from>[ lab]": bench.py pairs a committed PMC profile
// with a run only when the two strings agree
const char* stpy_version(void) { return "stpy_hip 0.3 (gfx950) src " STPY_SRC_HASH
#if STPY_LAB
	" lab"
#endif
	; }
const char* stpy_last_error_string(void) { return g_err; }

// a workspace that is too small is refused here: the kernels cannot check it and would write past its end
#define WINV_CHECK(fn, have, n_) do { const int64_t need_ = stpy_potrf_winv_elems(n_); if ((have) < need_) { \
	set_error(fn ": winv holds %lld elements, %lld needed (stpy_potrf_winv_elems(n))", (long long)(have), (long long)need_); return -21; } } while (0)
#define WORK_CHECK(fn, have, need) do { const int64_t need_ = (need); if ((have) < need_) { \
	set_error(fn ": workspace of %lld bytes, %lld needed (see the *_workspace_bytes query for these arguments)", (long long)(have), (long long)need_); return -20; } } while (0)

int stpy_gram(int kind, int dtype, const void* a, int64_t n, int64_t lda, const void* b, int64_t q, int64_t ldb,
              int d, const int32_t* cols, const void* inv_ls, double kappa, double offset, double diag_add,
              int lower_only, int combine, void* out, int64_t ldo, void* work, int64_t work_bytes, void* stream)
{
	if (n <= 0 || q <= 0) return 0;          // empty problem: nothing to write (empty tensors have null data pointers)
	if (!a || !b || !out || !inv_ls) { set_error("stpy_gram: null pointer"); return -3; }
	if (work) WORK_CHECK("stpy_gram", work_bytes, stpy_gram_workspace_bytes(dtype, n, q, d));
	if (d <= 0 || ldo < n) { set_error("stpy_gram: bad dimensions d=%d ldo=%lld n=%lld", d, (long long)ldo, (long long)n); return -9; }
	// (with a column subset the rows must hold the largest selected column, which lives on the device: the caller's
	// contract; without one the first d columns are read)
	if (lda < 1 || ldb < 1 || (!cols && (lda < d || ldb < d))) { set_error("stpy_gram: leading dimensions lda=%lld ldb=%lld below d=%d", (long long)lda, (long long)ldb, d); return -5; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         gram<double>(kind, (const double*)a, n, lda, (const double*)b, q, ldb, d, cols, (const double*)inv_ls, kappa, offset, diag_add, lower_only, combine, (double*)out, ldo, work, st),
	         gram<float>(kind, (const float*)a, n, lda, (const float*)b, q, ldb, d, cols, (const float*)inv_ls, kappa, offset, diag_add, lower_only, combine, (float*)out, ldo, work, st));
}

int64_t stpy_gram_workspace_bytes(int dtype, int64_t n, int64_t q, int d)
{
	return gram_workspace_bytes(n, q, d, dtype == STPY_F64 ? 8 : 4);
}

int stpy_gram_diag(int kind, int dtype, const void* x, int64_t m, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                   double kappa, double offset, int combine, void* out, void* stream)
{
	if (m <= 0) return 0;
	if (!x || !out || !inv_ls) { set_error("stpy_gram_diag: null pointer"); return -3; }
	if (d < 0 || ldx < 1 || (!cols && ldx < d)) { set_error("stpy_gram_diag: bad dimensions"); return -5; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         gram_diag<double>(kind, (const double*)x, m, ldx, d, cols, (const double*)inv_ls, kappa, offset, combine, (double*)out, st),
	         gram_diag<float>(kind, (const float*)x, m, ldx, d, cols, (const float*)inv_ls, kappa, offset, combine, (float*)out, st));
}

int64_t stpy_potrf_workspace_bytes(int dtype, int64_t n, int nb)
{
	if (nb <= 0) nb = potrf_auto_nb(n);
	return potrf_workspace_bytes(n, nb, dtype == STPY_F64 ? 8 : 4);     /* two panel workspaces (look-ahead); fp32: + the panel's bf16 planes */
}

int64_t stpy_potrf_winv_elems(int64_t n) { return ((n + IB - 1) / IB) * (int64_t)IB * IB; }

int stpy_potrf(int dtype, int64_t n, void* A, int64_t lda, void* winv, int64_t winv_elems, void* work, int64_t work_bytes, int nb, int flags,
               int32_t* info_dev, void* stream)
{
	if (!A || !winv || !work || !info_dev) { set_error("stpy_potrf: null pointer"); return -3; }
	if (n <= 0 || lda < n) { set_error("stpy_potrf: bad dimensions n=%lld lda=%lld", (long long)n, (long long)lda); return -2; }
	WINV_CHECK("stpy_potrf", winv_elems, n);
	WORK_CHECK("stpy_potrf", work_bytes, stpy_potrf_workspace_bytes(dtype, n, nb));
	if (flags & ~STPY_FLAG_BESIDE_UPDATE) { set_error("stpy_potrf: unknown flag bits 0x%x", flags); return -10; }
	hipStream_t st = (hipStream_t)stream;
	const int gf = (flags & STPY_FLAG_BESIDE_UPDATE) ? GEMM_BESIDE : 0;
	DISPATCH(dtype,
	         potrf<double>(n, (double*)A, lda, (double*)winv, (double*)work, nb, info_dev, st, gf),
	         potrf<float>(n, (float*)A, lda, (float*)winv, (float*)work, nb, info_dev, st, gf));
}

int64_t stpy_trsm_workspace_bytes(int dtype, int64_t m, int64_t n, int nb)
{
	if (m <= 0 || n <= 0) return 0;
	if (nb <= 0) nb = trsm_auto_nb(m);
	if (trsm_is_recursive(dtype == STPY_F32 ? 4 : 8, m, false)) return 0;      // recursive form: no workspace
	if (n <= nb || (n < 32768 && g_trsm_right_looking != 2)) return 0;       // the right-looking sweep serves these sizes and needs none
	return (int64_t)TRSM_MAX_PASSES * m * nb * (int64_t)(dtype == STPY_F32 ? 4 : 8);
}

int stpy_trsm_right_lt(int dtype, int64_t m, int64_t n, const void* L, int64_t ldl, const void* winv, int64_t winv_elems, void* B, int64_t ldb, int nb, int flags,
                       void* work, int64_t work_bytes, void* stream)
{
	if (m <= 0 || n <= 0) return 0;
	if (!L || !winv || !B) { set_error("stpy_trsm_right_lt: null pointer"); return -4; }
	WINV_CHECK("stpy_trsm_right_lt", winv_elems, n);
	if (flags & ~STPY_FLAG_BESIDE_UPDATE) { set_error("stpy_trsm_right_lt: unknown flag bits 0x%x", flags); return -11; }
	const int gf = (flags & STPY_FLAG_BESIDE_UPDATE) ? GEMM_BESIDE : 0;
	if (work) WORK_CHECK("stpy_trsm_right_lt", work_bytes, stpy_trsm_workspace_bytes(dtype, m, n, nb));
	if (m < 0 || n <= 0 || ldl < n || ldb < n) { set_error("stpy_trsm_right_lt: bad dimensions"); return -2; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         trsm_right_lt<double>(m, n, (const double*)L, ldl, (const double*)winv, (double*)B, ldb, nb, st, false, (double*)work, gf),
	         trsm_right_lt<float>(m, n, (const float*)L, ldl, (const float*)winv, (float*)B, ldb, nb, st, false, (float*)work, gf));
}

int stpy_potri(int dtype, int64_t n, const void* L, int64_t ldl, const void* winv, int64_t winv_elems, void* Kinv, int64_t ldk, void* work, int64_t work_bytes, void* stream)
{
	if (!L || !winv || !Kinv || !work) { set_error("stpy_potri: null pointer"); return -3; }
	if (n <= 0 || ldl < n || ldk < n) { set_error("stpy_potri: bad dimensions"); return -2; }
	WINV_CHECK("stpy_potri", winv_elems, n);
	WORK_CHECK("stpy_potri", work_bytes, n * n * (int64_t)(dtype == STPY_F32 ? 4 : 8));
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         potri_lower<double>(n, (const double*)L, ldl, (const double*)winv, (double*)Kinv, ldk, (double*)work, st),
	         potri_lower<float>(n, (const float*)L, ldl, (const float*)winv, (float*)Kinv, ldk, (float*)work, st));
}

int stpy_lml_weight(int kind, int dtype, const void* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                    double kappa, double weight, const void* alpha, const void* Kinv, int64_t ldk, void* H, int64_t ldh, void* work, int64_t work_bytes, void* stream)
{
	if (!x || !inv_ls || !alpha || !H || !work) { set_error("stpy_lml_weight: null pointer"); return -3; }
	if (Kinv == H) Kinv = nullptr;          // in place
	if (Kinv && ldk < n) { set_error("stpy_lml_weight: ldk=%lld < n=%lld", (long long)ldk, (long long)n); return -13; }
	if (n <= 0 || d <= 0 || ldh < n || ldx < 1 || (!cols && ldx < d)) { set_error("stpy_lml_weight: bad dimensions n=%lld d=%d ldx=%lld ldh=%lld", (long long)n, d, (long long)ldx, (long long)ldh); return -5; }
	WORK_CHECK("stpy_lml_weight", work_bytes, stpy_gram_workspace_bytes(dtype, n, n, d));
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         lml_weight<double>(kind, (const double*)x, n, ldx, d, cols, (const double*)inv_ls, kappa, weight, (const double*)alpha, (const double*)Kinv, ldk, (double*)H, ldh, work, st),
	         lml_weight<float>(kind, (const float*)x, n, ldx, d, cols, (const float*)inv_ls, kappa, weight, (const float*)alpha, (const float*)Kinv, ldk, (float*)H, ldh, work, st));
}

int stpy_trsv(int dtype, int64_t n, const void* L, int64_t ldl, const void* winv, int64_t winv_elems, void* y, void* out, int trans, void* stream)
{
	if (!L || !winv || !y || !out || y == out) { set_error("stpy_trsv: null or aliased pointer (y is scratch, out must differ)"); return -3; }
	if (n <= 0 || ldl < n) { set_error("stpy_trsv: bad dimensions"); return -2; }
	WINV_CHECK("stpy_trsv", winv_elems, n);
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         trsv<double>(n, (const double*)L, ldl, (const double*)winv, (double*)y, (double*)out, trans, st),
	         trsv<float>(n, (const float*)L, ldl, (const float*)winv, (float*)y, (float*)out, trans, st));
}

int stpy_predict(int dtype, int64_t m, int64_t n, const void* X, int64_t ldx, const void* z, const void* kdiag,
                 void* mu, void* sigma, int clamp, void* stream)
{
	if (m <= 0) return 0;
	if (!X || !z || (sigma && !kdiag && clamp != 2)) { set_error("stpy_predict: null pointer"); return -4; }
	if (n < 0 || ldx < n) { set_error("stpy_predict: ldx=%lld < n=%lld", (long long)ldx, (long long)n); return -5; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         predict<double>(m, n, (const double*)X, ldx, (const double*)z, (const double*)kdiag, (double*)mu, (double*)sigma, clamp, st),
	         predict<float>(m, n, (const float*)X, ldx, (const float*)z, (const float*)kdiag, (float*)mu, (float*)sigma, clamp, st));
}

int stpy_predict_finish(int dtype, int64_t m, void* mu, const void* sumsq, const void* kdiag, double scale, void* sigma, int clamp, void* stream)
{
	if (m <= 0) return 0;
	if (sigma && (!sumsq || !kdiag)) { set_error("stpy_predict_finish: sigma needs sumsq and kdiag"); return -4; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         predict_finish<double>(m, (double*)mu, (const double*)sumsq, (const double*)kdiag, scale, (double*)sigma, clamp, st),
	         predict_finish<float>(m, (float*)mu, (const float*)sumsq, (const float*)kdiag, scale, (float*)sigma, clamp, st));
}

int stpy_combine(int dtype, int64_t m, int64_t n, void* out, int64_t ldo, const void* src, int64_t lds, int combine, double diag_add, void* stream)
{
	if (m <= 0 || n <= 0) return 0;
	if (!out || !src) { set_error("stpy_combine: null pointer"); return -4; }
	if (ldo < n || lds < n) { set_error("stpy_combine: leading dimensions ldo=%lld lds=%lld below n=%lld", (long long)ldo, (long long)lds, (long long)n); return -5; }
	if (combine < STPY_OUT_SET || combine > STPY_OUT_MUL) { set_error("stpy_combine: unknown combine %d", combine); return -8; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         combine_into<double>(m, n, (double*)out, ldo, (const double*)src, lds, combine, diag_add, st),
	         combine_into<float>(m, n, (float*)out, ldo, (const float*)src, lds, combine, diag_add, st));
}

int stpy_logdet_quad(int dtype, int64_t n, const void* L, int64_t ldl, const void* z, void* out2, void* stream)
{
	if (!L || !out2) { set_error("stpy_logdet_quad: null pointer"); return -3; }
	if (n <= 0 || ldl < n) { set_error("stpy_logdet_quad: bad dimensions"); return -2; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         logdet_quad<double>(n, (const double*)L, ldl, (const double*)z, (double*)out2, st),
	         logdet_quad<float>(n, (const float*)L, ldl, (const float*)z, (float*)out2, st));
}

int stpy_gemm_nt(int dtype, int64_t m, int64_t n, int64_t k, const void* A, int64_t lda, const void* B, int64_t ldb,
                 void* C, int64_t ldc, int mode, int lower_only, void* stream)
{
	if (m <= 0 || n <= 0) return 0;
	if (!A || !B || !C) { set_error("stpy_gemm_nt: null pointer"); return -5; }
	if (k < 0 || lda < k || ldb < k || ldc < n) { set_error("stpy_gemm_nt: leading dimensions lda=%lld ldb=%lld (k=%lld) ldc=%lld (n=%lld)", (long long)lda, (long long)ldb, (long long)k, (long long)ldc, (long long)n); return -6; }
	if (mode < 0 || mode > 2) { set_error("stpy_gemm_nt: mode %d (0: C = A B^T, 1: C -= A B^T, 2: C += A B^T)", mode); return -11; }
	if (mode == 2) mode = 5;          // (internal numbering: 2-4 are the fused-epilogue forms)
	hipStream_t st = (hipStream_t)stream;
	ProfScope ps(TAG_GEMM_API, (lower_only && m == n) ? (double)m * (double)n * (double)k : 2.0 * (double)m * (double)n * (double)k, st);
	DISPATCH(dtype,
	         gemm_nt<double>(m, n, k, (const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, (double*)nullptr, 0, mode, lower_only, st),
	         gemm_nt<float>(m, n, k, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, (float*)nullptr, 0, mode, lower_only, st));
}

int64_t stpy_syrk_workspace_bytes(int dtype, int64_t n, int64_t k)
{
	return (dtype == STPY_F32 && g_gemm_bf3 > 0) ? syrk_planes_workspace_bytes(n, k) : 0;
}

int stpy_syrk(int dtype, int64_t n, int64_t k, const void* A, int64_t lda, void* C, int64_t ldc, int mode, void* work, int64_t work_bytes, void* stream)
{
	if (n <= 0) return 0;
	if (!A || !C) { set_error("stpy_syrk: null pointer"); return -4; }
	if (k < 0 || lda < k || ldc < n) { set_error("stpy_syrk: leading dimensions lda=%lld (k=%lld) ldc=%lld (n=%lld)", (long long)lda, (long long)k, (long long)ldc, (long long)n); return -5; }
	if (mode < 0 || mode > 2) { set_error("stpy_syrk: mode %d (0: C = A A^T, 1: C -= A A^T, 2: C += A A^T)", mode); return -8; }
	const int imode = mode == 2 ? 5 : mode;
	hipStream_t st = (hipStream_t)stream;
	ProfScope ps(TAG_GEMM_API, (double)n * (double)n * (double)k, st);
	const int64_t need = stpy_syrk_workspace_bytes(dtype, n, k);
	if (work && need > 0) {
		WORK_CHECK("stpy_syrk", work_bytes, need);
		if (lda % 4 == 0 && (((uintptr_t)A | (uintptr_t)work) & 15) == 0 && ldc < (1 << 24))
			return syrk_planes(n, k, (const float*)A, lda, (float*)C, ldc, imode, work, st);
	}
	DISPATCH(dtype,
	         gemm_nt<double>(n, n, k, (const double*)A, lda, (const double*)A, lda, (double*)C, ldc, (double*)nullptr, 0, imode, 1, st),
	         gemm_nt<float>(n, n, k, (const float*)A, lda, (const float*)A, lda, (float*)C, ldc, (float*)nullptr, 0, imode, 1, st));
}

int stpy_gemm_nt_splitk_passes(int64_t m, int64_t n, int64_t k)
{
	return gemm_splitk_plan(m, n, k);
}

int stpy_gemm_nt_splitk(int dtype, int64_t m, int64_t n, int64_t k, const void* A, int64_t lda, const void* B, int64_t ldb,
                        void* C, int64_t ldc, int mode, int passes, void* work, int64_t work_bytes, void* stream)
{
	if (m <= 0 || n <= 0) return 0;
	if (!A || !B || !C) { set_error("stpy_gemm_nt_splitk: null pointer"); return -5; }
	if (k < 0 || lda < k || ldb < k || ldc < n) { set_error("stpy_gemm_nt_splitk: leading dimensions"); return -6; }
	if (passes > 1 && !work) { set_error("stpy_gemm_nt_splitk: %d passes need a workspace of passes*m*n elements", passes); return -5; }
	if (passes > 1) WORK_CHECK("stpy_gemm_nt_splitk", work_bytes, (int64_t)passes * m * n * (int64_t)(dtype == STPY_F32 ? 4 : 8));
	if (mode != 0 && mode != 1) { set_error("stpy_gemm_nt_splitk: mode must be 0 or 1"); return -11; }
	hipStream_t st = (hipStream_t)stream;
	ProfScope ps(TAG_GEMM_API, 2.0 * (double)m * (double)n * (double)k, st);
	DISPATCH(dtype,
	         gemm_nt<double>(m, n, k, (const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, (double*)nullptr, 0, mode, 0, st, nullptr, nullptr, nullptr, passes, (double*)work),
	         gemm_nt<float>(m, n, k, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, (float*)nullptr, 0, mode, 0, st, nullptr, nullptr, nullptr, passes, (float*)work));
}

int stpy_gemm_nt_bc(int dtype, int64_t m, int64_t n, int64_t k, const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int64_t ldc, int mode, int nb_dist, int pr, int pc, int myr, int myc, int i0, int j0, void* stream)
{
	if (m <= 0 || n <= 0) return 0;
	if (!A || !B || !C) { set_error("stpy_gemm_nt_bc: null pointer"); return -5; }
	if (k < 0 || lda < k || ldb < k || ldc < n) { set_error("stpy_gemm_nt_bc: leading dimensions"); return -6; }
	if (pr <= 0 || pc <= 0 || myr < 0 || myr >= pr || myc < 0 || myc >= pc || i0 < 0 || j0 < 0) { set_error("stpy_gemm_nt_bc: bad process-grid arguments"); return -13; }
	hipStream_t st = (hipStream_t)stream;
	BlockCyclic bc{nb_dist, pr, pc, myr, myc, i0, j0};
	// algorithmic flops: only the distribution blocks on or below the global diagonal are computed
	double elems = (double)m * (double)n;
	if (g_prof_on.load() && nb_dist >= 128 && nb_dist % 128 == 0) {
		elems = 0;
		for (int64_t bi = 0; bi * nb_dist < m; ++bi) {
			const int64_t I = (bi + i0) * pr + myr, rows = std::min<int64_t>(nb_dist, m - bi * nb_dist);
			for (int64_t bj = 0; bj * nb_dist < n; ++bj)
				if (I >= (bj + j0) * pc + myc) elems += (double)rows * (double)std::min<int64_t>(nb_dist, n - bj * nb_dist);
		}
	}
	ProfScope ps(TAG_GEMM_API, 2.0 * elems * (double)k, st);
	DISPATCH(dtype,
	         gemm_nt<double>(m, n, k, (const double*)A, lda, (const double*)B, ldb, (double*)C, ldc, (double*)nullptr, 0, mode, 0, st, &bc),
	         gemm_nt<float>(m, n, k, (const float*)A, lda, (const float*)B, ldb, (float*)C, ldc, (float*)nullptr, 0, mode, 0, st, &bc));
}

int stpy_symmetrize_lower(int dtype, int64_t n, void* A, int64_t lda, void* stream)
{
	if (!A) { set_error("stpy_symmetrize_lower: null pointer"); return -3; }
	if (n <= 0 || lda < n) { set_error("stpy_symmetrize_lower: bad dimensions"); return -2; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype, symmetrize_lower<double>(n, (double*)A, lda, st), symmetrize_lower<float>(n, (float*)A, lda, st));
}

int stpy_tril(int dtype, int64_t n, void* A, int64_t lda, void* stream)
{
	if (n <= 0) return 0;
	if (!A) { set_error("stpy_tril: null pointer"); return -3; }
	if (lda < n) { set_error("stpy_tril: lda=%lld < n=%lld", (long long)lda, (long long)n); return -4; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype, tril<double>(n, (double*)A, lda, st), tril<float>(n, (float*)A, lda, st));
}

int stpy_trace_dot(int dtype, int64_t n, const void* A, int64_t lda, const void* u, const void* v, void* out2, void* stream)
{
	if (!out2 || (u && !v)) { set_error("stpy_trace_dot: null pointer"); return -7; }
	if (n < 0 || (A && lda < n)) { set_error("stpy_trace_dot: bad dimensions"); return -4; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         trace_dot<double>(n, (const double*)A, lda, (const double*)u, (const double*)v, (double*)out2, st),
	         trace_dot<float>(n, (const float*)A, lda, (const float*)u, (const float*)v, (float*)out2, st));
}

int stpy_scaled_points_t(int dtype, const void* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                         void* out, int64_t ldo, int ones_row, void* stream)
{
	if (n <= 0) return 0;
	if (!x || !inv_ls || !out) { set_error("stpy_scaled_points_t: null pointer"); return -2; }
	if (d < 0 || d > 65534 || ldx < 1 || (!cols && ldx < d) || ldo < n) { set_error("stpy_scaled_points_t: bad dimensions"); return -5; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         scaled_points_t<double>((const double*)x, n, ldx, d, cols, (const double*)inv_ls, (double*)out, ldo, ones_row, st),
	         scaled_points_t<float>((const float*)x, n, ldx, d, cols, (const float*)inv_ls, (float*)out, ldo, ones_row, st));
}

int stpy_lml_grad_reduce(int dtype, const void* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                         const void* P, int64_t ldp, const int32_t* pidx, void* acc, void* stream)
{
	if (n <= 0 || d <= 0) return 0;
	if (!x || !inv_ls || !P || !acc) { set_error("stpy_lml_grad_reduce: null pointer"); return -2; }
	if (ldx < 1 || (!cols && ldx < d) || ldp < d + 1) { set_error("stpy_lml_grad_reduce: bad dimensions"); return -9; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         lml_grad_reduce<double>((const double*)x, n, ldx, d, cols, (const double*)inv_ls, (const double*)P, ldp, pidx, (double*)acc, st),
	         lml_grad_reduce<float>((const float*)x, n, ldx, d, cols, (const float*)inv_ls, (const float*)P, ldp, pidx, (float*)acc, st));
}

int stpy_lml_grad_cov_reduce(int dtype, const void* x, int64_t n, int64_t ldx, int dg, const int32_t* cols,
                             const void* z, int64_t ldz, int p, const void* P, int64_t ldp, void* out, void* stream)
{
	if (n <= 0 || dg <= 0 || p <= 0) return 0;
	if (!x || !z || !P || !out) { set_error("stpy_lml_grad_cov_reduce: null pointer"); return -2; }
	if (ldx < 1 || (!cols && ldx < dg) || ldz < p || ldp < p + 1) { set_error("stpy_lml_grad_cov_reduce: bad dimensions"); return -8; }
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         lml_grad_cov_reduce<double>((const double*)x, n, ldx, dg, cols, (const double*)z, ldz, p, (const double*)P, ldp, (double*)out, st),
	         lml_grad_cov_reduce<float>((const float*)x, n, ldx, dg, cols, (const float*)z, ldz, p, (const float*)P, ldp, (float*)out, st));
}

int64_t stpy_rff_workspace_bytes(int dtype, int64_t n, int d, int64_t m)
{
	if (n <= 0 || m <= 0 || d <= 0) return 0;
	return rff_workspace_bytes(dtype == STPY_F32 ? 4 : 8, n, d, m);
}

int stpy_rff_embed(int dtype, const void* x, int64_t n, int64_t ldx, int d, const void* W, int64_t ldw, int64_t m,
                   const void* bias, const void* feat_scale, double scale, void* out, int64_t ldo, int transposed,
                   void* work, int64_t work_bytes, void* stream)
{
	if (n <= 0 || m <= 0) return 0;
	if (!x || !W || !out) { set_error("stpy_rff_embed: null pointer"); return -2; }
	if (d <= 0 || ldx < d || ldw < d || ldo < (transposed ? n : m)) { set_error("stpy_rff_embed: bad dimensions"); return -5; }
	if (work) WORK_CHECK("stpy_rff_embed", work_bytes, stpy_rff_workspace_bytes(dtype, n, d, m));
	hipStream_t st = (hipStream_t)stream;
	DISPATCH(dtype,
	         rff_embed<double>((const double*)x, n, ldx, d, (const double*)W, ldw, m, (const double*)bias, (const double*)feat_scale, scale, (double*)out, ldo, transposed, work, work_bytes, st),
	         rff_embed<float>((const float*)x, n, ldx, d, (const float*)W, ldw, m, (const float*)bias, (const float*)feat_scale, scale, (float*)out, ldo, transposed, work, work_bytes, st));
}

/* Route switches (every build; include/stpy_hip.h lists them) and, in the lab build only, the experiment knobs of common.h. */
#if STPY_LAB
}  // extern "C"
namespace stpy {
#define STPY_KNOB_DEFINE(key, name, dflt) int name = dflt;
STPY_KNOB_LIST(STPY_KNOB_DEFINE)
#undef STPY_KNOB_DEFINE
}
extern "C" {
#endif

void stpy_tune(int key, int value)
{
	switch (key) {
	case 5: g_trsm_right_looking = value; return;
	case 8: g_gemm_k128 = value; return;
	case 9: if (STPY_LAB || value != 3) g_rff_tile = value; return;          // (3 = the direct-store streaming variant: lab build only)
	case 16: g_trsv_flow = value; return;
	case 17: g_trsm_strip = (value == 1 || value == 128 || value == 256 || value == 512 || value == 1024) ? value : 0; return;
	case 26: g_gemm_bf3 = value; return;
	case 28: g_gram_fill = value; return;
	case 30: g_gemm_sliver_tiles = value; return;
	case 32: g_potrf_presplit = value; return;
	default: break;
	}
#if STPY_LAB
	if (key == 3) { g_trsm_pass_depth = value > 0 ? value : 1024; return; }
	if (key == 4) { g_trsm_wg_target = value > 0 ? value : 2048; return; }
#define STPY_KNOB_SET(k, name, dflt) if (key == k) { name = value; return; }
	STPY_KNOB_LIST(STPY_KNOB_SET)
#undef STPY_KNOB_SET
#endif
}

/* current value of a switch; -1: no such key in this build */
int stpy_tune_get(int key)
{
	switch (key) {
	case 5: return g_trsm_right_looking;
	case 8: return g_gemm_k128;
	case 9: return g_rff_tile;
	case 16: return g_trsv_flow;
	case 17: return g_trsm_strip;
	case 26: return g_gemm_bf3;
	case 28: return g_gram_fill;
	case 30: return g_gemm_sliver_tiles;
	case 32: return g_potrf_presplit;
	default: break;
	}
#if STPY_LAB
#define STPY_KNOB_GET(k, name, dflt) if (key == k) return name;
	STPY_KNOB_LIST(STPY_KNOB_GET)
#undef STPY_KNOB_GET
#endif
	return -1;
}

/* Errors that only the device can detect after the call has returned (today: a hand-off wait of the one-launch vector solve that
 * gave up -- its output is then NaN).  Waits for `stream`, returns 0 or a positive code (1 = stpy_trsv hand-off timed out) and
 * clears the word; < 0: the query itself failed. */
int stpy_async_status(void* stream)
{
	int status = 0;
	const int rc = trsv_async_status((hipStream_t)stream, &status);
	return rc ? rc : status;
}

/* profiler: enable != 0 starts a fresh recording; read() waits for the recorded events */
void stpy_profile_enable(int enable)
{
	std::lock_guard<std::mutex> lock(g_prof_mutex);
	if (enable) g_prof_used = 0;
	g_prof_on.store(enable != 0);
}

int stpy_profile_read(int tag, double* total_ms, double* total_flops, int64_t* launches)
{
	double ms = 0, fl = 0;
	int64_t cnt = 0;
	std::lock_guard<std::mutex> lock(g_prof_mutex);
	for (size_t i = 0; i < g_prof_used; ++i) {
		if (g_prof[i].tag != tag) continue;
		if (hipEventSynchronize(g_prof[i].e1) != hipSuccess) { set_error("stpy_profile_read: event sync failed"); return -1; }
		float t = 0;
		if (hipEventElapsedTime(&t, g_prof[i].e0, g_prof[i].e1) != hipSuccess) { set_error("stpy_profile_read: elapsed failed"); return -1; }
		ms += t; fl += g_prof[i].flops; ++cnt;
	}
	if (total_ms) *total_ms = ms;
	if (total_flops) *total_flops = fl;
	if (launches) *launches = cnt;
	return 0;
}

/* Same records, but launches that overlap in time (look-ahead side stream) are not double counted:
 * busy_ms = length of the union of the [start, end] intervals of all launches whose tag bit is set
 * in tagmask (bit t = tag t). */
int stpy_profile_read_union(int tagmask, double* busy_ms, double* total_flops, int64_t* launches)
{
	std::vector<std::pair<float, float>> iv;
	double fl = 0;
	int64_t cnt = 0;
	hipEvent_t base = nullptr;
	std::lock_guard<std::mutex> lock(g_prof_mutex);
	for (size_t i = 0; i < g_prof_used; ++i) {
		if (!((tagmask >> g_prof[i].tag) & 1)) continue;
		if (hipEventSynchronize(g_prof[i].e1) != hipSuccess) { set_error("stpy_profile_read_union: event sync failed"); return -1; }
		if (!base) base = g_prof[i].e0;
		float t0 = 0, t1 = 0;
		if (hipEventElapsedTime(&t0, base, g_prof[i].e0) != hipSuccess || hipEventElapsedTime(&t1, base, g_prof[i].e1) != hipSuccess) {
			set_error("stpy_profile_read_union: elapsed failed"); return -1;
		}
		iv.emplace_back(t0, t1);
		fl += g_prof[i].flops; ++cnt;
	}
	std::sort(iv.begin(), iv.end());
	double busy = 0;
	float cs = 0, ce = -1;
	for (auto& p : iv) {
		if (ce < cs || p.first > ce) { if (ce >= cs) busy += ce - cs; cs = p.first; ce = p.second; }
		else if (p.second > ce) ce = p.second;
	}
	if (ce >= cs && !iv.empty()) busy += ce - cs;
	if (busy_ms) *busy_ms = busy;
	if (total_flops) *total_flops = fl;
	if (launches) *launches = cnt;
	return 0;
}

}  // extern "C"
