// Shared declarations of libstpy_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/stpy_hip.h"

namespace stpy {

constexpr int IB = 128;          // inner (diagonal) block of the factorisation / solves
constexpr int POTRF_DEFAULT_NB = 1024, TRSM_DEFAULT_NB = 512;
// ---- switches ---------------------------------------------------------------------------------------------------------
// ROUTE switches (stpy_tune keys 5, 8, 9, 16, 17, 26, 28, 30; every build): which of the SHIPPED kernels serves a call where the library
// normally decides by size -- tests/ use them to reach every shipped path at small sizes.  Process-wide, read at launch time.
extern int g_trsm_right_looking, g_gemm_k128, g_rff_tile, g_trsv_flow, g_trsm_strip, g_gemm_bf3, g_gram_fill, g_gemm_sliver_tiles, g_potrf_presplit;
// EXPERIMENT knobs: compile-time constants in the product library (the measured defaults); variables behind stpy_tune only in
// the lab build (make EXPERIMENTS=1 -> libstpy_hip_lab.so, used by tools/).  The kernels and code paths that only a non-default
// value reaches are compiled under #if STPY_LAB, so the product library does not carry them.  Where each default comes from is
// documented at the place that reads it (gemm.hip, potrf.hip, solve.hip, rff.hip).
#ifdef STPY_EXPERIMENTS
#define STPY_LAB 1
#define STPY_KNOB(key, name, dflt) extern int name;
#else
#define STPY_LAB 0
#define STPY_KNOB(key, name, dflt) constexpr int name = dflt;
#endif
#define STPY_KNOB_LIST(X) \
	X(0, g_gemm_stagger, 40000) X(1, g_gemm_exp, 0) X(2, g_potf2_scalar, 0) X(3, g_trsm_pass_depth, 1024) X(4, g_trsm_wg_target, 2048) \
	X(6, g_gemm_dtv, 1) X(7, g_potrf_diag_first_below, 0) X(10, g_potrf_beside_min, 0) X(11, g_potf2_sliver, 0) \
	X(12, g_potrf_reserve_below, 0) X(13, g_potrf_reserve_above, 2048) X(14, g_potrf_nb256_upto, 2048) X(15, g_potrf_nb512_upto, 16384) \
	X(18, g_potrf_strip, 0) X(19, g_rff_wgs, 0) X(20, g_gemm_tri_diag_last, 0) X(21, g_potrf_serial_below, 3200) X(22, g_trsv_fault_ticket, 0) X(23, g_potrf_nb1024_upto, 32768) X(24, g_potrf_first_nb, 0) X(25, g_potrf_nb2048_upto, 1 << 30) X(29, g_potf2_flow, 1) X(31, g_potrf_serial_band, 0) X(33, g_potrf_planes_min_tiles, 800)
STPY_KNOB_LIST(STPY_KNOB)
constexpr int g_gemm_dtv_min_k = 64;
// panel width when the caller passes nb = 0: narrower panels shorten the latency-bound panel chain, which a
// small trailing matrix cannot hide; the thresholds are documented with their measurements in potrf.hip.  Above 32 768 remaining
// rows the panels are 2048 wide (round 3): half as many trailing-update launches, whose per-launch fixed cost (C-tile traffic and
// the partly filled last round, ~0.8 ms at 32 768 rows) is the larger part of the gap between the kernel's K-loop slope and what a
// K = 1024 update delivers -- potrf N = 65 536: 1358.7 -> 1341.9 ms and 1385.2 -> 1368.6 ms on two boxes (tools/potrf_sweep.py "23=...")
inline int potrf_auto_nb(int64_t n) { return n <= g_potrf_nb256_upto ? 256 : (n <= g_potrf_nb512_upto ? 512 : (n <= g_potrf_nb1024_upto ? POTRF_DEFAULT_NB : (n <= g_potrf_nb2048_upto ? 2 * POTRF_DEFAULT_NB : 4 * POTRF_DEFAULT_NB))); }

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// ---- look-ahead side stream (highest priority) + events used by potrf and the block solves.  One set per
// ---- (device, caller stream), created on first use and kept for the life of the process: two host threads that
// ---- drive two streams never share a side stream or an event, and calls on ONE stream are ordered by that stream.
struct LookAhead {
	hipStream_t side = nullptr;
	hipEvent_t col_ready = nullptr, panel_done = nullptr, trail_done = nullptr;
	// "reserved" mode of potrf: update stream masked off one CU per XCD, diagonal-block stream masked onto those CUs
	hipStream_t upd = nullptr, diag = nullptr;          // created on first use (potrf.hip: lookahead_reserved_streams)
	bool reserved_tried = false;
	hipEvent_t ev_diag = nullptr, ev_gemm = nullptr, ev_mode = nullptr;
	// 64 bytes of device memory: ticket / published-count / error words of the one-launch vector solve (solve.hip)
	void* trsv_sync = nullptr;
};
int lookahead_acquire(hipStream_t caller, LookAhead** out);

// per-call behaviour flags of stpy_potrf / stpy_trsm_right_lt (include/stpy_hip.h: STPY_FLAG_*), passed down to the GEMM launcher
constexpr int GEMM_BESIDE = 1;       // = STPY_FLAG_BESIDE_UPDATE: the launch runs while a trailing update floods the chip -- take the kernels
                                     // that fit into what two update workgroups leave over on a CU (never the 128 KiB one-volley kernel)

// ---- optional launch profiler (stpy_profile_*): HIP events recorded on the launch stream around
// ---- every tagged kernel, so bench.py can report the dominant kernel's live average duration.
enum { TAG_SYRK = 0, TAG_PANEL_GEMM = 1, TAG_TRSM_GEMM = 2, TAG_POTF2 = 3, TAG_GEMM_API = 4, TAG_COUNT = 5 };
struct ProfScope {
	int slot;
	hipStream_t st;
	ProfScope(int tag, double flops, hipStream_t st);
	~ProfScope();
};

// ---- MFMA 16x16x4 traits: the only thing that differs between f64 and f32 is the builtin and
// ---- the C/D row map (cdna_hip_programming.md section 3: f64 row = (lane>>4) + 4*reg,
// ---- f32 row = 4*(lane>>4) + reg); A/B: lane l holds A[l&15][k=l>>4], B[k=l>>4][l&15].
template <typename T> struct Mfma;
template <> struct Mfma<double> {
	typedef double v4 __attribute__((ext_vector_type(4)));
	typedef double v2 __attribute__((ext_vector_type(2)));
	static __device__ __forceinline__ v4 mma(double a, double b, v4 c) {
		return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	}
	// c - a b: the f64 MFMA encodes neg:[a,b,c] in the blgp field, so the subtraction is free (no VALU negation
	// in the K loop, no VALU -> MFMA wait states)
	static constexpr bool HAS_NEG = true;
	static __device__ __forceinline__ v4 mms(double a, double b, v4 c) {
		return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 1);
	}
	static __device__ __forceinline__ int crow(int lane, int i) { return (lane >> 4) + 4 * i; }
};
template <> struct Mfma<float> {
	typedef float v4 __attribute__((ext_vector_type(4)));
	typedef float v2 __attribute__((ext_vector_type(2)));
	static __device__ __forceinline__ v4 mma(float a, float b, v4 c) {
		return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
	}
	static constexpr bool HAS_NEG = false;           // (blgp is a lane-group broadcast pattern for f32, not a negation)
	static __device__ __forceinline__ v4 mms(float a, float b, v4 c) {
		return __builtin_amdgcn_mfma_f32_16x16x4f32(-a, b, c, 0, 0, 0);
	}
	static __device__ __forceinline__ int crow(int lane, int i) { return (lane >> 4) * 4 + i; }
};

// ---- internal launchers (all enqueue on `st`, return 0 or a negative error code) ----
struct BlockCyclic { int nb_dist, pr, pc, myr, myc, i0, j0; };
template <typename T> struct RffEpilogue { int half; T scale; const T* bias; int by_row; const T* fscale; };
template <typename T> struct GramEpilogue { int kind, combine; T kappa, offset, diag_add; const T* na; const T* nb; const T* alpha; T weight; };
template <typename T>
int gemm_nt(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B, int64_t ldb,
            T* C, int64_t ldc, T* C2, int64_t ldc2, int mode, int lower_only, hipStream_t st, const BlockCyclic* bc = nullptr,
            const RffEpilogue<T>* rff = nullptr, const GramEpilogue<T>* gr = nullptr, int ksplit = 1, T* split_work = nullptr, int gflags = 0);
int gemm_splitk_plan(int64_t m, int64_t n, int64_t k);
// fp32 trailing updates from a panel split once into three bf16 planes (gemm_bf3p.hip): the split pass, and C -= A B^T with A / B = plane rows arow.. / brow..
int bf3_split(const float* X, int64_t ldx, int64_t rows, int64_t cols, unsigned short* pl, int64_t ldp, int64_t pstride, int64_t prow0, hipStream_t st);
int gemm_nt_bf3p(int64_t m, int64_t n, int64_t k, const unsigned short* pl, int64_t ldp, int64_t pstride, int64_t arow, int64_t brow,
                 float* C, int64_t ldc, int tri, hipStream_t st, int mode = 1, int nch = 1, float* ws = nullptr);
int64_t syrk_planes_workspace_bytes(int64_t n, int64_t k);
int syrk_planes(int64_t n, int64_t k, const float* A, int64_t lda, float* C, int64_t ldc, int mode, void* work, hipStream_t st);
// bytes of the factorisation's workspace: two panel buffers (look-ahead) and, for fp32, the three bf16 planes of one panel
inline int64_t potrf_workspace_bytes(int64_t n, int nb, int esize) { return 2 * n * (int64_t)nb * esize + (esize == 4 ? 3 * n * (int64_t)nb * 2 : 0); }
// dedicated fp64 Gram fill (gemm.hip): 1 = taken, 0 = not this kernel's shape (the caller falls back to the GEMM epilogue), < 0 = error
int gram_fill_f64(int kind, const double* as, const double* bs, const double* na, const double* nb, int dpad, int64_t n, int64_t q,
                  double kappa, double offset, double diag_add, int lower_only, int combine, double* out, int64_t ldo, hipStream_t st);      // recommended number of K passes for a product with few output tiles
template <typename T>
int potf2_trtri(T* A, int64_t lda, int nbk, T* W, T* P2, int64_t ldp2, int32_t* info, int block_row0, hipStream_t st, bool beside = false);
template <typename T>
int potrf(int64_t n, T* A, int64_t lda, T* winv, T* work, int nb, int32_t* info, hipStream_t st, int gflags = 0);
template <typename T>
int trsm_right_lt(int64_t m, int64_t n, const T* L, int64_t ldl, const T* winv, T* B, int64_t ldb, int nb, hipStream_t st, bool upper_rhs = false, T* work = nullptr, int gflags = 0);
int trsm_auto_nb(int64_t m);
constexpr int TRSM_MAX_PASSES = 16;      // split-K of the long left-looking products (needs the workspace)
// which form stpy_trsm_right_lt takes (shared by the solve and by stpy_trsm_workspace_bytes): the recursive one needs no workspace
inline bool trsm_is_recursive(size_t elem, int64_t m, bool upper_rhs)
{
	if (g_trsm_right_looking >= 3) return true;
	if (g_trsm_right_looking != 0 || upper_rhs) return false;
	return m >= 2048 || g_trsm_strip > 0;
}
// X <- X L_D^-T for a diagonal block of 1..8 128-blocks in one launch (solve.hip); X2: optional second copy of the result
template <typename T>
int trsm_strip(int64_t m, const T* Ld, int64_t ldl, const T* W, T* X, int64_t ldx, T* X2, int64_t ldx2, int64_t w, hipStream_t st);
bool trsm_strip_ok(size_t elem, const void* L, int64_t ldl, const void* winv);
template <typename T>
int potri_lower(int64_t n, const T* L, int64_t ldl, const T* winv, T* Kinv, int64_t ldk, T* work, hipStream_t st);
template <typename T>
int lml_weight(int kind, const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, double kappa, double weight,
               const T* alpha, const T* Hsrc, int64_t ldhs, T* H, int64_t ldh, void* work, hipStream_t st);
template <typename T>
int combine_into(int64_t m, int64_t n, T* out, int64_t ldo, const T* src, int64_t lds, int combine, double diag_add, hipStream_t st);
template <typename T>
int predict_finish(int64_t m, T* mu, const T* sumsq, const T* kdiag, double scale, T* sigma, int clamp, hipStream_t st);
template <typename T>
int trsv(int64_t n, const T* L, int64_t ldl, const T* winv, T* y, T* out, int trans, hipStream_t st);
int trsv_async_status(hipStream_t st, int* status);      // solve.hip: sticky device error word of the one-launch vector solves on `st`
template <typename T>
int predict(int64_t m, int64_t n, const T* X, int64_t ldx, const T* z, const T* kdiag, T* mu, T* sigma, int clamp, hipStream_t st);
template <typename T>
int logdet_quad(int64_t n, const T* L, int64_t ldl, const T* z, T* out2, hipStream_t st);
template <typename T>
int symmetrize_lower(int64_t n, T* A, int64_t lda, hipStream_t st);
// reduce.hip: small reductions / layout helpers of the evidence gradient, the samplers and the scalar summaries
template <typename T>
int tril(int64_t n, T* A, int64_t lda, hipStream_t st);
template <typename T>
int trace_dot(int64_t n, const T* A, int64_t lda, const T* u, const T* v, T* out2, hipStream_t st);
template <typename T>
int scaled_points_t(const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, T* out, int64_t ldo, int ones_row, hipStream_t st);
template <typename T>
int lml_grad_cov_reduce(const T* x, int64_t n, int64_t ldx, int dg, const int32_t* cols, const T* z, int64_t ldz, int pdim, const T* P, int64_t ldp,
                        T* out, hipStream_t st);
template <typename T>
int lml_grad_reduce(const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, const T* P, int64_t ldp,
                    const int32_t* pidx, T* acc, hipStream_t st);
template <typename T>
int gram(int kind, const T* a, int64_t n, int64_t lda, const T* b, int64_t q, int64_t ldb, int d,
         const int32_t* cols, const T* inv_ls, double kappa, double offset, double diag_add,
         int lower_only, int combine, T* out, int64_t ldo, void* work, hipStream_t st);
int64_t gram_workspace_bytes(int64_t n, int64_t q, int d, size_t esz);
template <typename T>
int gram_diag(int kind, const T* x, int64_t m, int64_t ldx, int d, const int32_t* cols, const T* inv_ls,
              double kappa, double offset, int combine, T* out, hipStream_t st);
template <typename T>
int rff_embed(const T* x, int64_t n, int64_t ldx, int d, const T* W, int64_t ldw, int64_t m,
              const T* bias, const T* feat_scale, double scale, T* out, int64_t ldo, int transposed, void* work, int64_t work_bytes, hipStream_t st);
int64_t rff_workspace_bytes(int elem, int64_t n, int d, int64_t m);

}  // namespace stpy
