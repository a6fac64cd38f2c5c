// solve.hip -- everything downstream of the factor: B L^-T for blocks of right-hand sides (MFMA
// GEMMs with the cached inverse diagonal blocks), L^-1 y / L^-T y for single vectors (HBM-bound
// streaming of L), the fused prediction epilogue and the log-marginal reductions.
#include <atomic>
#include <type_traits>

#include <cstddef>
#include "common.h"

namespace stpy {

// ------------------------------------------------------------------------------------------
// B <- B L^-T,  B: m x n (rows = right-hand sides).  Same two-level blocking as potrf:
// left-looking over 128-column blocks inside an nb-wide panel, right-looking between panels.
// ------------------------------------------------------------------------------------------
// upper_rhs: B is upper triangular on entry (the identity, for the inverse factor) and stays so: rows
// below column block c are zero in that block, so every product is restricted to the rows above --
// n^3/3 flops instead of n^3.
// the 128-column blocks of one nb-wide panel, left-looking (small, latency-bound launches)
template <typename T>
static int solve_panel(int64_t m, int64_t n, int64_t k, int64_t kb, const T* L, int64_t ldl, const T* winv, T* B, int64_t ldb, hipStream_t st, bool upper_rhs, int gflags)
{
	int rc;
	for (int64_t c = k; c < k + kb; c += IB) {
		const int64_t cb = (n - c < IB) ? (n - c) : IB;
		const int64_t jj = c - k;
		const int64_t mc = upper_rhs ? ((c + cb < m) ? c + cb : m) : m;       // rows that can be non-zero in this block column
		if (jj > 0) {   // B[:, c:c+cb] -= B[:, k:c] L[c:c+cb, k:c]^T
			ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mc * (double)cb * (double)jj, st);
			rc = gemm_nt<T>(mc, cb, jj, B + k, ldb, L + c * ldl + k, ldl, B + c, ldb, (T*)nullptr, 0, 1, 0, st, nullptr, nullptr, nullptr, 1, nullptr, gflags);
			if (rc) return rc;
		}
		// B[:, c:c+cb] <- B[:, c:c+cb] inverse(L_cc)^T   (one column tile => safe in place)
		{
			ProfScope ps(TAG_TRSM_GEMM, (double)mc * (double)cb * (double)cb, st);
			rc = gemm_nt<T>(mc, cb, cb, B + c, ldb, winv + (c / IB) * IB * IB, IB, B + c, ldb, (T*)nullptr, 0, 0, 0, st, nullptr, nullptr, nullptr, 1, nullptr, gflags);
		}
		if (rc) return rc;
	}
	return 0;
}

// ------------------------------------------------------------------------------------------
// X <- X L_D^-T for an NBLK x NBLK block D of 128-blocks on the diagonal of L, as ONE launch ("strip" kernel).
// Used as the leaf of the recursive block solve and for the rows below the diagonal block of a Cholesky panel.
// Through the recursion a leaf is 3 launches per 256 columns plus the small products of the levels above it -- 32-128 tiles
// each on a chip with 512 workgroup slots: at n = 16 384, M = 4096 the levels below 1024 columns took ~5 of 24 ms for 2 % of
// the flops.  Here a workgroup owns 16 rows for the whole block:
//   * its 16 x (128 NBLK) strip lives in the MFMA accumulators (wave w: the 32-column quarter w of every 128-block);
//   * block j: the strip's block (already carrying - sum_{i<j} X_i L_ji^T) goes through LDS to become the A operand of
//     X_j = . W_j^T (W_j = inverse(L_jj), cached), X_j is stored, goes through LDS again and updates the later blocks
//     acc_jj -= X_j L_jj,j^T;
//   * the B operands (W_j, L_jj,j: 128 x 128, shared by every workgroup, L2-resident) are read straight into registers one
//     K tile ahead -- no LDS staging, no barrier inside a product.
// Rows beyond m are read as zero and not stored.  X2 (optional): a second copy of the result (the Cholesky panel workspace).
// ------------------------------------------------------------------------------------------
constexpr int TS_ROWS = 16;
template <typename T, int NBLK>
__global__ __launch_bounds__(256)
void trsm_strip_kernel(const T* __restrict__ Ld, int64_t ldl, const T* __restrict__ W, T* __restrict__ X, int64_t ldx, T* __restrict__ X2, int64_t ldx2, int m)
{
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	constexpr int CE = 16 / (int)sizeof(T);                  // elements of a 16-byte chunk: 2 doubles / 4 floats
	constexpr int NCH = 4 / CE;                              // chunks per lane and K tile of 16: each lane supplies 4 of the 16 k's
	constexpr int TS_LD = IB + 16 / (int)sizeof(T) * 1;      // LDS row stride 130 doubles / 132 floats: 16 rows -> 16 different 4-bank groups
	typedef T chunk __attribute__((ext_vector_type(CE)));
	__shared__ __attribute__((aligned(16))) T tile[TS_ROWS * TS_LD];
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int r16 = lane & 15, g = lane >> 4;
	const int64_t row0 = (int64_t)blockIdx.x * TS_ROWS;
	T* const xs = X + row0 * ldx;          // this workgroup's strip (the pointers are already at the block's first column)
	T* const xs2 = X2 ? X2 + row0 * ldx2 : nullptr;
	const int rows_left = m - (int)row0;   // rows of this strip that exist

	// accumulators: acc[j][t][i] = row crow(lane, i), column 128 j + 32 wave + 16 t + r16
	v4 acc[NBLK][2];
#pragma unroll
	for (int j = 0; j < NBLK; ++j)
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int r = MM::crow(lane, i);
				acc[j][t][i] = r < rows_left ? xs[(int64_t)r * ldx + 128 * j + 32 * wave + 16 * t + r16] : T(0);
			}

	// C fragment -> LDS tile (16 x 128, K-contiguous rows): the A operand of the next product.  NEG: the tile holds -c (fp32 has
	// no negating MFMA; the subtracting products then accumulate (-X_j) L^T)
	auto to_tile = [&](const v4 (&c)[2], bool neg) {
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int i = 0; i < 4; ++i) tile[MM::crow(lane, i) * TS_LD + 32 * wave + 16 * t + r16] = neg ? -c[t][i] : c[t][i];
	};
	// out[t] (+|-)= tile (16 x 128) * Bblk[32 wave + 16 t + r16][0:128]^T ; Bblk row-major with leading dimension ldb
	auto product = [&](auto sub_tag, v4 (&out)[2], const T* Bblk, int64_t ldb) {
		constexpr bool SUBT = decltype(sub_tag)::value && MM::HAS_NEG;
		const T* brow0 = Bblk + (int64_t)(32 * wave + r16) * ldb + NCH * g * CE;          // lane group g: chunks NCH g .. of every K tile
		const T* brow1 = brow0 + 16 * ldb;
		const T* arow = tile + r16 * TS_LD + NCH * g * CE;
		chunk b0[2][NCH], b1[2][NCH];          // [tile][chunk] for K tile kt (b0) and kt + 1 (b1)
		auto loadb = [&](chunk (&b)[2][NCH], int kt) {
#pragma unroll
			for (int h = 0; h < NCH; ++h) { b[0][h] = *(const chunk*)(brow0 + kt * 16 + CE * h); b[1][h] = *(const chunk*)(brow1 + kt * 16 + CE * h); }
		};
		auto step = [&](const chunk (&b)[2][NCH], int kt) {
#pragma unroll
			for (int h = 0; h < NCH; ++h) {
				const chunk a = *(const chunk*)(arow + kt * 16 + CE * h);
#pragma unroll
				for (int s2 = 0; s2 < CE; ++s2)
#pragma unroll
					for (int t = 0; t < 2; ++t) out[t] = SUBT ? MM::mms(a[s2], b[t][h][s2], out[t]) : MM::mma(a[s2], b[t][h][s2], out[t]);
			}
		};
		loadb(b0, 0);
#pragma unroll
		for (int kt = 0; kt < 8; kt += 2) {
			loadb(b1, kt + 1);
			step(b0, kt);
			if (kt + 2 < 8) loadb(b0, kt + 2);
			step(b1, kt + 1);
		}
	};

#pragma unroll
	for (int j = 0; j < NBLK; ++j) {
		to_tile(acc[j], false);
		__syncthreads();
		v4 xj[2] = {v4{0, 0, 0, 0}, v4{0, 0, 0, 0}};
		product(std::false_type{}, xj, W + (int64_t)j * IB * IB, IB);          // X_j = (.) inverse(L_jj)^T
#pragma unroll
		for (int t = 0; t < 2; ++t)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int r = MM::crow(lane, i);
				if (r < rows_left) {
					xs[(int64_t)r * ldx + 128 * j + 32 * wave + 16 * t + r16] = xj[t][i];
					if (xs2) xs2[(int64_t)r * ldx2 + 128 * j + 32 * wave + 16 * t + r16] = xj[t][i];
				}
			}
		if (j + 1 < NBLK) {
			__syncthreads();                     // everybody has read the tile
			to_tile(xj, !MM::HAS_NEG);
			__syncthreads();
#pragma unroll
			for (int jj = j + 1; jj < NBLK; ++jj)
				product(std::true_type{}, acc[jj], Ld + (int64_t)(128 * jj) * ldl + 128 * j, ldl);          // acc_jj -= X_j L_jj,j^T
			__syncthreads();
		}
	}
}

int g_trsm_strip = 1;            // stpy_tune key 17: leaf width of the recursive block solve handled by trsm_strip_kernel (0 = off; 512; 1024;
                                 // 1 = the default, 512 since round 4; 128 / 256 / 512 / 1024 force a width)

// Ld: the diagonal block's first element (L + c0 * ldl + c0); W: its first inverse 128-block; X / X2 at the block's first column.
template <typename T>
int trsm_strip(int64_t m, const T* Ld, int64_t ldl, const T* W, T* X, int64_t ldx, T* X2, int64_t ldx2, int64_t w, hipStream_t st)
{
	if (m <= 0) return 0;
	if (w % IB != 0 || w > 8 * IB || m > (int64_t)TS_ROWS * 0x7fffff00 / 256 * 256 || ldl % (16 / sizeof(T)) != 0 ||
	    (((uintptr_t)Ld | (uintptr_t)W) & 15) != 0) { set_error("trsm strip: unsupported shape / alignment"); return -2; }
	const dim3 grid((unsigned)((m + TS_ROWS - 1) / TS_ROWS)), block(256);
	switch ((int)(w / IB)) {
#define STPY_STRIP(NB) case NB: hipLaunchKernelGGL((trsm_strip_kernel<T, NB>), grid, block, 0, st, Ld, ldl, W, X, ldx, X2, ldx2, (int)m); break;
		STPY_STRIP(1) STPY_STRIP(2) STPY_STRIP(3) STPY_STRIP(4) STPY_STRIP(5) STPY_STRIP(6) STPY_STRIP(7) STPY_STRIP(8)
#undef STPY_STRIP
	}
	return check_launch("trsm (strip)");
}
template int trsm_strip<double>(int64_t, const double*, int64_t, const double*, double*, int64_t, double*, int64_t, int64_t, hipStream_t);
template int trsm_strip<float>(int64_t, const float*, int64_t, const float*, float*, int64_t, float*, int64_t, int64_t, hipStream_t);
bool trsm_strip_ok(size_t elem, const void* L, int64_t ldl, const void* winv)
{
	return ldl % (int64_t)(16 / elem) == 0 && (((uintptr_t)L | (uintptr_t)winv) & 15) == 0;
}

#define HIPCHK_S(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("trsm: %s failed: %s", #x, hipGetErrorString(e_)); return -1000 - (int)e_; } } while (0)

// Without a workspace (and for A/B timing under stpy_tune key 5): right-looking between panels with one panel of look-ahead (same scheme as potrf): after panel k is
// solved, the update of the NEXT panel's columns goes first, then the next panel's latency-bound
// 128-blocks run on the side stream while the caller's stream updates the remaining columns.
template <typename T>
static int trsm_right_looking(int64_t m, int64_t n, const T* L, int64_t ldl, const T* winv, T* B, int64_t ldb, int nb, hipStream_t st, bool upper_rhs, int gflags)
{
	if (nb <= 0) nb = TRSM_DEFAULT_NB;
	if (nb % IB != 0) { set_error("trsm: nb must be a multiple of %d", IB); return -9; }
	LookAhead* la = nullptr;
	int rc = lookahead_acquire(st, &la);
	if (rc) return rc;
	rc = solve_panel<T>(m, n, 0, (n < nb) ? n : nb, L, ldl, winv, B, ldb, st, upper_rhs, gflags);
	if (rc) return rc;
	for (int64_t k = 0; k + nb < n; k += nb) {
		const int64_t r = k + nb;
		const int64_t nkb = (n - r < nb) ? (n - r) : nb;
		const int64_t mr = upper_rhs ? ((r < m) ? r : m) : m;                 // X[:, k:k+nb] is zero below row k+nb
		{   // next panel's columns: B[:, r:r+nkb] -= B[:, k:r] L[r:r+nkb, k:r]^T
			ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mr * (double)nkb * (double)nb, st);
			rc = gemm_nt<T>(mr, nkb, nb, B + k, ldb, L + r * ldl + k, ldl, B + r, ldb, (T*)nullptr, 0, 1, 0, st);
			if (rc) return rc;
		}
		HIPCHK_S(hipEventRecord(la->col_ready, st));
		HIPCHK_S(hipStreamWaitEvent(la->side, la->col_ready, 0));
		rc = solve_panel<T>(m, n, r, nkb, L, ldl, winv, B, ldb, la->side, upper_rhs, gflags);
		if (rc) return rc;
		HIPCHK_S(hipEventRecord(la->panel_done, la->side));
		if (r + nkb < n) {  // the rest: B[:, r+nkb:] -= B[:, k:r] L[r+nkb:, k:r]^T
			const int64_t r2 = r + nkb;
			ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mr * (double)(n - r2) * (double)nb, st);
			rc = gemm_nt<T>(mr, n - r2, nb, B + k, ldb, L + r2 * ldl + k, ldl, B + r2, ldb, (T*)nullptr, 0, 1, 0, st);
			if (rc) return rc;
		}
		HIPCHK_S(hipStreamWaitEvent(st, la->panel_done, 0));
	}
	return 0;
}


// Recursive form (stpy_tune key 5 = 3): columns [c0, c0+w) split at h; solve the left part, subtract its
// contribution from the right part in ONE product (m x (w-h) x h: long K and many tiles for all but the
// deepest levels, half of all flops in the top-level product alone), solve the right part.  Everything is
// in order on the caller's stream; the leaves are single 128-blocks.
template <typename T>
static int trsm_recursive(int64_t m, int64_t n, int64_t c0, int64_t w, const T* L, int64_t ldl, const T* winv, T* B, int64_t ldb, hipStream_t st, bool upper_rhs, int64_t leaf, int gflags)
{
	// (round 4: 512 at every row count -- with the products between the leaves on the sliver kernel (gemm.hip, route key 30) the 1024-column
	// strips of round 2, 140 us each on m / 16 workgroups, no longer pay: tools/trsm_sweep.py 1,128,256,512 17: n = 4096 / 8192 / 16 384, M = 256:
	// 0.74 -> 0.61, 1.65 -> 1.39, 3.91 -> 3.37 ms; M = 4096: 1.40 -> 1.34, 4.80 -> 4.68, 16.98 -> 16.73; 256 = 512 within 1 %, 128 = the old default)
	const int64_t strip_w = g_trsm_strip == 1 ? 512 : g_trsm_strip;
	// (not beside a trailing update: the strip kernel's 120-202 VGPRs do not fit next to two update workgroups; there the
	// 64-VGPR sliver products of solve_panel stay ahead -- potrf panels with the strip: 35.5 against 34.8 ms at N = 16 384)
	if (strip_w > 0 && !(gflags & GEMM_BESIDE) && w <= strip_w && w % IB == 0 && !upper_rhs && c0 + w <= n && m < (1 << 30) && trsm_strip_ok(sizeof(T), L, ldl, winv)) {
		ProfScope ps(TAG_TRSM_GEMM, (double)m * (double)w * (double)w, st);
		return trsm_strip<T>(m, L + c0 * ldl + c0, ldl, winv + (c0 / IB) * IB * IB, B + c0, ldb, (T*)nullptr, 0, w, st);
	}
	if (w <= leaf) return solve_panel<T>(m, n, c0, w, L, ldl, winv, B, ldb, st, upper_rhs, gflags);
	int64_t h = IB;
	while (h * 2 < w) h *= 2;                                 // largest power-of-two multiple of 128 below w
	int rc = trsm_recursive<T>(m, n, c0, h, L, ldl, winv, B, ldb, st, upper_rhs, leaf, gflags);
	if (rc) return rc;
	const int64_t mr = upper_rhs ? ((c0 + h < m) ? c0 + h : m) : m;      // X[:, c0:c0+h) is zero from row c0+h on
	{
		ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mr * (double)(w - h) * (double)h, st);
		rc = gemm_nt<T>(mr, w - h, h, B + c0, ldb, L + (c0 + h) * ldl + c0, ldl, B + c0 + h, ldb, (T*)nullptr, 0, 1, 0, st);
		if (rc) return rc;
	}
	return trsm_recursive<T>(m, n, c0 + h, w - h, L, ldl, winv, B, ldb, st, upper_rhs, leaf, gflags);
}

// LEFT-looking between panels: panel p (columns [k, k+nb)) first receives every earlier panel's
// contribution in ONE long-K product,  B[:, k:k+nb] -= X[:, 0:k) L[k:k+nb, 0:k)^T,  and is then solved.
// The output of that product is read and written once per panel (a right-looking sweep re-reads
// the whole remaining right-hand side once per panel, at K = nb per visit), and a long K range
// amortises the per-tile prologue/epilogue (70 TFLOP/s fp64 on these shapes in isolation,
// tools/quick_bench.py leftlook).  In place the whole solve gains less: 299 ms against 311 ms for the
// right-looking sweep at N = 65536, M = 4096 (tools/predict_bench.py) -- 59 vs 57 TFLOP/s.
// Look-ahead: the product is split at the previous panel.  G1(p+1), the part over columns [0, k) that
// needs only panels <= p-1, runs on the caller's stream while the side stream finishes panel p:
// G2(p) (the K = nb slice over panel p-1) and the latency-bound 128-blocks S(p).
// G1 is cut into K passes (workspace) so that it runs as several rounds of workgroups and the side
// stream's kernels get CU slots in between.
int g_trsm_right_looking = 0;
// g_trsm_pass_depth = 1024: least K depth of one pass;  g_trsm_wg_target = 2048: workgroups a long product is cut into (four rounds of two per CU)

int trsm_auto_nb(int64_t)
{
	return TRSM_DEFAULT_NB;
}

template <typename T>
int trsm_right_lt(int64_t m, int64_t n, const T* L, int64_t ldl, const T* winv, T* B, int64_t ldb, int nb, hipStream_t st, bool upper_rhs, T* work, int gflags)
{
	// no workspace: the long products cannot be cut into K passes, and one round of workgroups that
	// holds every CU slot would keep the next panel's small kernels out -- the right-looking sweep
	// is the faster form then (311 vs 335 ms at N = 65536, M = 4096; 299 ms with the workspace)
	// ... and below n = 32768 in any case (28.1 vs 29.9 ms at n = 16384, 10.0 vs 11.0 at 8192, equal at 32768)
	// (stpy_tune key 5: 1 forces the right-looking sweep, 2 the left-looking form at any n -- tests, A/B timing)
	// Default from 2048 right-hand sides on: the recursive form (tools/trsm_sweep.py, fp64, M = 4096: 27.1 -> 24.7 ms at
	// n = 16384, 83.9 -> 80.4 at 32768, 288 -> 284 at 65536; M = 10112: 705 -> 650 ms at 65536 = 66.9 TFLOP/s).  With fewer
	// rows its lower levels have too few tiles per product and the panel forms below, which overlap the small kernels with
	// a large product on a second stream (and split K), stay ahead (n = 65536, M = 1024: 97 vs 107 ms).
	// Round 2: with the leaves of the recursion as one strip launch each (trsm_strip_kernel, fp64, rows a multiple of 16) the
	// recursive form is ahead at every row count (tools/trsm_sweep.py 0,4 5 f64only 256,512,1024,1536: n = 16 384, M = 256
	// 8.8 -> 6.4 ms, M = 1024 13.8 -> 10.5 ms; n = 65 536, M = 1024 91 -> 87 ms), so fp64 takes it whenever the strips apply.
	// (not for the triangular right-hand side of the inverse: value + gradient at N = 32768 0.615 s with the sweep, 0.677 s recursive)
	// (tried and dropped: slabs of 1024 right-hand-side rows, each running this recursion on a stream of its own, to fill the chip
	// in the deep levels whose products have few tiles.  One process, tools/trsm_sweep.py: n = 16 384, M = 4096 25.0 ms on one
	// stream, 35.4 ms with two slabs, 42.6 ms with four; the same ordering at every size up to n = 65 536.)
	if (trsm_is_recursive(sizeof(T), m, upper_rhs))
		return trsm_recursive<T>(m, n, 0, n, L, ldl, winv, B, ldb, st, upper_rhs, g_trsm_right_looking >= 3 ? (int64_t)IB << (g_trsm_right_looking - 3) : 2 * IB, gflags);
	if (g_trsm_right_looking == 1 || !work || (n < 32768 && g_trsm_right_looking != 2)) return trsm_right_looking<T>(m, n, L, ldl, winv, B, ldb, nb, st, upper_rhs, gflags);
	if (nb <= 0) nb = trsm_auto_nb(m);
	if (nb % IB != 0) { set_error("trsm: nb must be a multiple of %d", IB); return -9; }
	int rc = solve_panel<T>(m, n, 0, (n < nb) ? n : nb, L, ldl, winv, B, ldb, st, upper_rhs, gflags);
	if (rc || n <= nb) return rc;
	LookAhead* la = nullptr;
	rc = lookahead_acquire(st, &la);
	if (rc) return rc;
	hipStream_t side = la->side;
	HIPCHK_S(hipEventRecord(la->col_ready, st));            // "G1(1)" is empty: panel 1 only waits for S(0)
	for (int64_t k = nb; k < n; k += nb) {
		const int64_t w = (n - k < nb) ? (n - k) : nb;
		const int64_t mr = upper_rhs ? ((k < m) ? k : m) : m;   // X[:, 0:k) is zero from row k on
		// ---- side stream: G2(p) over panel p-1, then the 128-blocks of panel p
		HIPCHK_S(hipStreamWaitEvent(side, la->col_ready, 0));   // G1(p) (issued one iteration ago) has landed
		{
			ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mr * (double)w * (double)nb, side);
			rc = gemm_nt<T>(mr, w, nb, B + (k - nb), ldb, L + k * ldl + (k - nb), ldl, B + k, ldb, (T*)nullptr, 0, 1, 0, side);
			if (rc) return rc;
		}
		rc = solve_panel<T>(m, n, k, w, L, ldl, winv, B, ldb, side, upper_rhs, gflags);
		if (rc) return rc;
		// ---- caller's stream: G1(p+1) over columns [0, k): needs S(p-1), not S(p)
		if (k > nb) HIPCHK_S(hipStreamWaitEvent(st, la->panel_done, 0));
		const int64_t k1 = k + nb;
		if (k1 < n) {
			const int64_t w1 = (n - k1 < nb) ? (n - k1) : nb;
			ProfScope ps(TAG_TRSM_GEMM, 2.0 * (double)mr * (double)w1 * (double)k, st);
			// K passes (workspace given): several rounds of shorter workgroups instead of one round that
			// holds every CU slot for the whole product -- the side stream's kernels get in between rounds
			int passes = 1;
			if (work) {
				const int64_t tiles = ((mr + IB - 1) / IB) * ((w1 + IB - 1) / IB);
				passes = (int)(k / g_trsm_pass_depth);
				if (passes > g_trsm_wg_target / tiles) passes = (int)(g_trsm_wg_target / tiles);
				if (passes > TRSM_MAX_PASSES) passes = TRSM_MAX_PASSES;
				if (passes < 1) passes = 1;
			}
			rc = gemm_nt<T>(mr, w1, k, B, ldb, L + k1 * ldl, ldl, B + k1, ldb, (T*)nullptr, 0, 1, 0, st, nullptr, nullptr, nullptr, passes, work);
			if (rc) return rc;
			HIPCHK_S(hipEventRecord(la->col_ready, st));
		}
		HIPCHK_S(hipEventRecord(la->panel_done, side));       // S(p); recorded AFTER the wait above, which refers to S(p-1)
	}
	HIPCHK_S(hipStreamWaitEvent(st, la->panel_done, 0));
	return 0;
}

// ------------------------------------------------------------------------------------------
// K^-1 = L^-T L^-1 (lower triangle) from the factor: work <- I, work <- work L^-T (= L^-T, upper
// triangular, rows restricted as above), Kinv <- work work^T on the lower tiles with the K range of
// every tile row starting at its own first row.  2 n^3 / 3 flops, all on the MFMA GEMM.
// (The evidence gradient needs tr(K^-1 dK/dtheta): SURVEY.md section 8f rank 1.)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void set_identity_kernel(T* __restrict__ A, int64_t lda, int64_t n)
{
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n * n) return;
	const int64_t i = idx / n, j = idx - i * n;
	A[i * lda + j] = (i == j) ? T(1) : T(0);
}

template <typename T>
int potri_lower(int64_t n, const T* L, int64_t ldl, const T* winv, T* Kinv, int64_t ldk, T* work, hipStream_t st)
{
	if (n <= 0) return 0;
	hipLaunchKernelGGL((set_identity_kernel<T>), dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, st, work, n, n);
	int rc = check_launch("potri identity");
	if (rc) return rc;
	rc = trsm_right_lt<T>(n, n, L, ldl, winv, work, n, 0, st, true);
	if (rc) return rc;
	return gemm_nt<T>(n, n, n, work, n, work, n, Kinv, ldk, (T*)nullptr, 0, 0, 2, st);
}

// ------------------------------------------------------------------------------------------
// Vector solves, blocked by 128 with the cached inverse(L_cc).  One launch per block step:
// every workgroup recomputes the 128-vector  t = inverse(L_cc) y_c  (or its transpose form) in
// LDS -- W is 128 KiB and L2-hot -- then streams its share of L once.  Workgroup 0 stores t into
// `out`; updates go to the not-yet-solved part of y only, so there is no hazard on y_c.
// ------------------------------------------------------------------------------------------
constexpr int TV_THREADS = 256;
constexpr int TV_ROWS = 128;     // rows of L per workgroup in the forward sweep (8 passes of 16 rows, all loads in flight at once)
constexpr int TV_COLS = 128;     // columns of L per workgroup in the backward sweep (two threads per column, 64 rows each)

template <typename T>
__global__ __launch_bounds__(TV_THREADS)
void trsv_fwd_step(const T* __restrict__ L, int64_t ldl, const T* __restrict__ W, T* __restrict__ y,
                   T* __restrict__ out, int c, int cb, int n)
{
	__shared__ T yc[IB], tc[IB];
	const int tid = threadIdx.x;
	if (tid < IB) yc[tid] = tid < cb ? y[c + tid] : T(0);
	__syncthreads();
	// t = W y_c : 16 lanes per row, each with 8 consecutive columns (one coalesced 1 KiB read per row), 16 rows in flight,
	// all eight passes' loads issued before the first is consumed (the first version gave a thread half a row to walk
	// through alone: 64 dependent, uncoalesced loads in front of every step of the sweep)
	{
		const int l16 = tid & 15, rsub = tid >> 4;
		T yv[8], wv[8][8];
#pragma unroll
		for (int e = 0; e < 8; ++e) yv[e] = yc[l16 * 8 + e];
#pragma unroll
		for (int ps = 0; ps < 8; ++ps)
#pragma unroll
			for (int e = 0; e < 8; ++e) wv[ps][e] = W[(ps * 16 + rsub) * IB + l16 * 8 + e];
#pragma unroll
		for (int ps = 0; ps < 8; ++ps) {
			T s = T(0);
#pragma unroll
			for (int e = 0; e < 8; ++e) s += wv[ps][e] * yv[e];
			s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
			if (l16 == 0) tc[ps * 16 + rsub] = s;
		}
	}
	__syncthreads();
	if (blockIdx.x == 0) {
		if (tid < cb) out[c + tid] = tc[tid];
		return;
	}
	// y[r] -= L[r, c:c+cb] . t   for this workgroup's rows; 16 lanes per row, 8 elements per lane
	const int row_base = c + cb + (blockIdx.x - 1) * TV_ROWS;
	const int l16 = tid & 15, rsub = tid >> 4;            // 16 rows in flight per pass
	T tv[8];
#pragma unroll
	for (int e = 0; e < 8; ++e) tv[e] = tc[l16 * 8 + e];
	// one memory round trip for the whole workgroup: the eight passes' row segments and the y entries they update are all
	// requested before the first product (the sweep is a chain of n/128 launches; each used to take four round trips)
	constexpr int NPS = TV_ROWS / 16;
	T lv[NPS][8], yold[NPS];
#pragma unroll
	for (int ps = 0; ps < NPS; ++ps) {
		const int r = row_base + ps * 16 + rsub;
		const int rc = r < n ? r : n - 1;
		const T* lr = L + (int64_t)rc * ldl + c + l16 * 8;
#pragma unroll
		for (int e = 0; e < 8; ++e) lv[ps][e] = (l16 * 8 + e < cb) ? lr[e] : T(0);
		yold[ps] = y[rc];
	}
#pragma unroll
	for (int ps = 0; ps < NPS; ++ps) {
		const int r = row_base + ps * 16 + rsub;
		T s = T(0);
#pragma unroll
		for (int e = 0; e < 8; ++e) s += lv[ps][e] * tv[e];
		s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
		if (r < n && l16 == 0) y[r] = yold[ps] - s;
	}
}

template <typename T>
__global__ __launch_bounds__(TV_THREADS)
void trsv_bwd_step(const T* __restrict__ L, int64_t ldl, const T* __restrict__ W, T* __restrict__ y,
                   T* __restrict__ out, int c, int cb)
{
	__shared__ T yc[IB], tc[IB];
	const int tid = threadIdx.x;
	if (tid < IB) yc[tid] = tid < cb ? y[c + tid] : T(0);
	__syncthreads();
	// t = W^T y_c : t[k] = sum_i W[i][k] y_c[i]; lanes along k (coalesced), two halves of i
	{
		const int k = tid & 127, h = tid >> 7;
		T s = T(0);
#pragma unroll 16
		for (int i = h * 64; i < h * 64 + 64; ++i) s += W[i * IB + k] * yc[i];
		if (h == 1) tc[k] = s;
		__syncthreads();
		if (h == 0) tc[k] += s;
	}
	__syncthreads();
	if (blockIdx.x == 0) {
		if (tid < cb) out[c + tid] = tc[tid];
		return;
	}
	// y[j] -= sum_i L[c+i][j] t[i]  for this workgroup's columns j < c: two threads per column (64 rows each, lanes along j
	// so every row segment is one coalesced read), all 64 loads of a thread in flight together -- one memory round trip
	const int jl = tid & (TV_COLS - 1), hf = tid >> 7;
	const int j = (blockIdx.x - 1) * TV_COLS + jl;
	const bool live = j < c;
	const int jc = live ? j : 0;
	T lv[64];
#pragma unroll
	for (int i = 0; i < 64; ++i) {
		const int row = hf * 64 + i;
		lv[i] = (live && row < cb) ? L[(int64_t)(c + (row < cb ? row : 0)) * ldl + jc] : T(0);
	}
	const T yold = y[jc];
	T s = T(0);
#pragma unroll
	for (int i = 0; i < 64; ++i) s += lv[i] * tc[hf * 64 + i];
	__shared__ T red[TV_COLS];
	if (hf == 1) red[jl] = s;
	__syncthreads();
	if (hf == 0 && live) y[j] = yold - (s + red[jl]);
}

// ------------------------------------------------------------------------------------------
// The same solves as ONE launch (n a multiple of 128): a dataflow over the 128-blocks.  The step kernels above are a chain
// of n/128 dependent launches, ~12 us each: 1.5 ms per solve at n = 16 384 (3 ms of BASELINE config 2's 61 ms go to the two
// of them), 6 ms at n = 65 536, for 0.27 / 4.3 ms of HBM streaming.  Here workgroup k (k = its START order: a ticket taken
// from an atomic counter, so every lower ticket is resident or finished and waiting on it cannot deadlock) owns block
// i = k (forward) or nblk-1-k (backward); it streams its blocks of L once, two blocks ahead in registers (one wave per SIMD:
// 512 VGPRs), multiplies each with the solved block t_j as soon as that is published, and publishes t_i = inverse(L_ii) y_i.
// Hand-off of the 1 KiB t blocks: write-through (agent-scope relaxed atomic = sc1) stores, every storing wave drains its
// stores, workgroup barrier, ONE lane publishes the monotonic counter with an sc1 store; consumers poll the counter from
// one lane (sc1 load, s_sleep, bounded), barrier, and read t_j with sc1 loads only (MI355X_MICROARCH.md, "Valid forms":
// sc1 stores + drained + flag on one side, sc1 poll + barrier + sc1 loads on the other; one workgroup per CU).
// ------------------------------------------------------------------------------------------
// ticket / count are zeroed before every launch; `error` is STICKY (set when a hand-off wait gives up, read and cleared by
// stpy_async_status): a timed-out solve also poisons its output with NaN, so the failure shows in every result derived from it
struct TrsvSync { unsigned ticket, count, error, pad; };
int g_trsv_flow = 1;           // stpy_tune key 16: 0 = always the chain of step kernels

template <typename T>
__device__ __forceinline__ T load_sc1(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T>
__device__ __forceinline__ void store_sc1(T* p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <typename T, bool BACK>
__global__ __launch_bounds__(256, 1)
void trsv_flow_kernel(const T* __restrict__ L, int64_t ldl, const T* __restrict__ W, const T* __restrict__ y, T* out, int nblk, TrsvSync* sy, int fault_ticket)
{
	__shared__ int s_k, s_ready, s_failed;
	__shared__ T yc[IB];
	__shared__ T red[16][IB + 1];
	const int tid = threadIdx.x;
	if (tid == 0) { s_k = (int)atomicAdd(&sy->ticket, 1u); s_failed = 0; }
	__syncthreads();
	const int k = s_k;
	if (k >= nblk) return;
	const int i = BACK ? nblk - 1 - k : k;
	// thread -> 8 consecutive columns (c8) of the rows ps*16 + rr, ps = 0..7: every row segment of a 128 x 128 block is one
	// coalesced 1 KiB read of 16 lanes
	const int c8 = tid & 15, rr = tid >> 4;
	typedef T blk[8][8];
	auto load_block = [&](T (&v)[8][8], const T* base) {
#pragma unroll
		for (int ps = 0; ps < 8; ++ps) {
			const T* p = base + (int64_t)(ps * 16 + rr) * ldl + c8 * 8;
#pragma unroll
			for (int e = 0; e < 8; ++e) v[ps][e] = p[e];
		}
	};
	T wv[8][8];          // inverse(L_ii), row-major with leading dimension 128
#pragma unroll
	for (int ps = 0; ps < 8; ++ps)
#pragma unroll
		for (int e = 0; e < 8; ++e) wv[ps][e] = W[(int64_t)i * IB * IB + (ps * 16 + rr) * IB + c8 * 8 + e];

	// right-hand side of this block, fetched now (its latency would otherwise sit on the chain's critical path)
	const T y_early = BACK ? (tid < IB ? y[(int64_t)i * IB + tid] : T(0)) : T(0);
	T y_rows[8];
#pragma unroll
	for (int ps = 0; ps < 8; ++ps) y_rows[ps] = BACK ? T(0) : y[(int64_t)i * IB + ps * 16 + rr];
	int ready = 0;
	auto wait_for = [&](int j) {          // returns once block with ticket j has been published (all threads)
		if (tid == 0) {
			// (a workgroup far behind the front sleeps longer between polls: a hundred workgroups hammering one word slow the
			// publisher's own write-through stores down)
			unsigned c = load_sc1(&sy->count);
			for (int spin = 0; (int)c <= j && spin < 1000000; ++spin) {
				const int dist = k - (int)c;
				if (dist > 8) __builtin_amdgcn_s_sleep(127); else if (dist > 2) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(2);
				c = load_sc1(&sy->count);
			}
			// give up rather than hang -- LOUDLY: the sticky error word (stpy_async_status) and, below, NaN in this block of the
			// output, which every later block and every quantity derived from the solve inherits
			if ((int)c <= j) { atomicExch(&sy->error, 1u); s_failed = 1; c = (unsigned)nblk; }
			s_ready = (int)c;
		}
		__syncthreads();
		const int r = s_ready;
		__syncthreads();
		return r;
	};
	T acc[8];
#pragma unroll
	for (int e = 0; e < 8; ++e) acc[e] = T(0);
	// forward: acc[ps] = partial dot of row ps*16+rr over this thread's 8 columns; backward: acc[e] = partial sum of column
	// c8*8+e over this thread's 8 rows
	const int nprev = k;                  // number of already-solved blocks this one depends on (tickets 0..k-1)
	// two register images of a block, used alternately (the loop is unrolled by two so that both are indexed statically:
	// a run-time index sends the arrays to scratch)
	T lv0[8][8], lv1[8][8];
	auto block_of = [&](int q) {          // address of the block that pairs this workgroup's block with ticket q's
		const int j = BACK ? nblk - 1 - q : q;
		return BACK ? L + (int64_t)j * IB * ldl + (int64_t)i * IB : L + (int64_t)i * IB * ldl + (int64_t)j * IB;
	};
	auto consume = [&](const T (&lv)[8][8], int q) {
		if (q >= ready) ready = wait_for(q);
		const int j = BACK ? nblk - 1 - q : q;
		const T* tj = out + (int64_t)j * IB;
		T tv[8];
#pragma unroll
		for (int u = 0; u < 8; ++u) tv[u] = load_sc1(tj + (BACK ? u * 16 + rr : c8 * 8 + u));
		if (!BACK) {
#pragma unroll
			for (int ps = 0; ps < 8; ++ps)
#pragma unroll
				for (int e = 0; e < 8; ++e) acc[ps] += lv[ps][e] * tv[e];
		} else {
#pragma unroll
			for (int ps = 0; ps < 8; ++ps)
#pragma unroll
				for (int e = 0; e < 8; ++e) acc[e] += lv[ps][e] * tv[ps];
		}
	};
	if (nprev > 0) load_block(lv0, block_of(0));
	for (int q = 0; q < nprev; q += 2) {
		if (q + 1 < nprev) load_block(lv1, block_of(q + 1));
		consume(lv0, q);
		if (q + 1 < nprev) {
			if (q + 2 < nprev) load_block(lv0, block_of(q + 2));
			consume(lv1, q + 1);
		}
	}
	// ---- y_i = rhs_i - (accumulated products)
	if (!BACK) {
#pragma unroll
		for (int ps = 0; ps < 8; ++ps) {
			T sum = acc[ps];
			sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
			if (c8 == 0) yc[ps * 16 + rr] = y_rows[ps] - sum;
		}
	} else {
#pragma unroll
		for (int e = 0; e < 8; ++e) red[rr][c8 * 8 + e] = acc[e];
		__syncthreads();
		if (tid < IB) {
			T sum = T(0);
#pragma unroll
			for (int r = 0; r < 16; ++r) sum += red[r][tid];
			yc[tid] = y_early - sum;
		}
	}
	__syncthreads();
	// ---- t_i = inverse(L_ii) y_i (forward) or inverse(L_ii)^T y_i (backward), published write-through
	T* ti = out + (int64_t)i * IB;
	if (s_failed) {          // (uniform: written before a workgroup barrier every thread has passed) a wait timed out -> poison
		if (tid < IB) yc[tid] = (T)__builtin_nan("");
		__syncthreads();
	}
	if (!BACK) {
		T yv[8];
#pragma unroll
		for (int e = 0; e < 8; ++e) yv[e] = yc[c8 * 8 + e];
#pragma unroll
		for (int ps = 0; ps < 8; ++ps) {
			T sum = T(0);
#pragma unroll
			for (int e = 0; e < 8; ++e) sum += wv[ps][e] * yv[e];
			sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4); sum += __shfl_xor(sum, 8);
			if (c8 == 0) store_sc1(ti + ps * 16 + rr, sum);
		}
	} else {
		T part[8];
#pragma unroll
		for (int e = 0; e < 8; ++e) part[e] = T(0);
#pragma unroll
		for (int ps = 0; ps < 8; ++ps) {
			const T yr = yc[ps * 16 + rr];
#pragma unroll
			for (int e = 0; e < 8; ++e) part[e] += wv[ps][e] * yr;
		}
		__syncthreads();          // (red is reused)
#pragma unroll
		for (int e = 0; e < 8; ++e) red[rr][c8 * 8 + e] = part[e];
		__syncthreads();
		if (tid < IB) {
			T sum = T(0);
#pragma unroll
			for (int r = 0; r < 16; ++r) sum += red[r][tid];
			store_sc1(ti + tid, sum);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-through stores ...
	__syncthreads();                                           // ... before ONE lane publishes the counter
#if STPY_LAB
	if (k + 1 == fault_ticket) return;      // test hook (stpy_tune key 22 = ticket + 1; 0 = off): this block is never published -> its successor's wait must time out
#endif
	if (tid == 0) store_sc1(&sy->count, (unsigned)(k + 1));
}

// sticky error word of the vector solves issued on `st`: waits for the stream, returns the word and clears it
int trsv_async_status(hipStream_t st, int* status)
{
	LookAhead* la = nullptr;
	int rc = lookahead_acquire(st, &la);
	if (rc) return rc;
	*status = 0;
	if (!la->trsv_sync) return 0;
	unsigned err = 0;
	if (hipStreamSynchronize(st) != hipSuccess ||
	    hipMemcpy(&err, (const char*)la->trsv_sync + offsetof(TrsvSync, error), sizeof(err), hipMemcpyDeviceToHost) != hipSuccess ||
	    (err != 0 && hipMemset((char*)la->trsv_sync + offsetof(TrsvSync, error), 0, sizeof(err)) != hipSuccess)) {
		set_error("async status: reading the device error word failed");
		return -1005;
	}
	*status = (int)err;
	return 0;
}

template <typename T>
int trsv(int64_t n, const T* L, int64_t ldl, const T* winv, T* y, T* out, int trans, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("trsv: n exceeds int32"); return -2; }
	if (g_trsv_flow && n % IB == 0 && n >= 4 * IB && ldl % (16 / (int)sizeof(T)) == 0) {
		LookAhead* la = nullptr;
		int rc = lookahead_acquire(st, &la);
		if (rc) return rc;
		if (la->trsv_sync) {
			if (hipMemsetAsync(la->trsv_sync, 0, 2 * sizeof(unsigned), st) != hipSuccess) { set_error("trsv: hipMemsetAsync failed"); return -1004; }          // ticket + count; `error` is sticky
			const int nblk = (int)(n / IB);
			// 84 KiB of (unused) dynamic LDS: ONE workgroup per CU, the configuration the sc1 hand-off form is measured for -- and
			// what the kernel wants anyway (one wave per SIMD, three register images of a block)
			constexpr int PAD_LDS = 84 * 1024;
			static std::atomic<bool> attr_set[2];
			const int which = sizeof(T) == 8 ? 0 : 1;
			if (!attr_set[which].load(std::memory_order_acquire)) {
				hipError_t e = hipFuncSetAttribute((const void*)trsv_flow_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, PAD_LDS);
				if (e == hipSuccess) e = hipFuncSetAttribute((const void*)trsv_flow_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PAD_LDS);
				if (e != hipSuccess) { set_error("trsv: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return -1000 - (int)e; }
				attr_set[which].store(true, std::memory_order_release);
			}
			if (!trans) hipLaunchKernelGGL((trsv_flow_kernel<T, false>), dim3((unsigned)nblk), dim3(256), PAD_LDS, st, L, ldl, winv, (const T*)y, out, nblk, (TrsvSync*)la->trsv_sync, g_trsv_fault_ticket);
			else hipLaunchKernelGGL((trsv_flow_kernel<T, true>), dim3((unsigned)nblk), dim3(256), PAD_LDS, st, L, ldl, winv, (const T*)y, out, nblk, (TrsvSync*)la->trsv_sync, g_trsv_fault_ticket);
			return check_launch("trsv (flow)");
		}
	}
	if (!trans) {
		for (int64_t c = 0; c < n; c += IB) {
			const int cb = (int)((n - c < IB) ? (n - c) : IB);
			const int64_t rest = n - c - cb;
			const unsigned grid = 1 + (unsigned)((rest + TV_ROWS - 1) / TV_ROWS);
			hipLaunchKernelGGL((trsv_fwd_step<T>), dim3(grid), dim3(TV_THREADS), 0, st, L, ldl, winv + (c / IB) * IB * IB, y, out, (int)c, cb, (int)n);
		}
	} else {
		const int64_t last = ((n - 1) / IB) * IB;
		for (int64_t c = last; c >= 0; c -= IB) {
			const int cb = (int)((n - c < IB) ? (n - c) : IB);
			const unsigned grid = 1 + (unsigned)((c + TV_COLS - 1) / TV_COLS);
			hipLaunchKernelGGL((trsv_bwd_step<T>), dim3(grid), dim3(TV_THREADS), 0, st, L, ldl, winv + (c / IB) * IB * IB, y, out, (int)c, cb);
		}
	}
	return check_launch("trsv");
}

// ------------------------------------------------------------------------------------------
// Prediction epilogue: X = K* L^-T is streamed ONCE (M * N * w bytes: HBM-bound), mu and the variance term come out of the
// same pass.  A workgroup owns PR_ROWS consecutive test points: every 16-byte chunk of z it loads is used against all of their
// rows (one workgroup per row re-read the whole of z from L2 for every row -- as many bytes again as X itself through the
// L1/L2 path), and each lane keeps PR_ROWS + 1 sixteen-byte loads per step in flight, two steps unrolled.
// ------------------------------------------------------------------------------------------
constexpr int PR_ROWS = 4;
template <typename T, bool VEC>
__global__ __launch_bounds__(256)
void predict_kernel(const T* __restrict__ X, int64_t ldx, int m, int n, const T* __restrict__ z,
                    const T* __restrict__ kdiag, T* __restrict__ mu, T* __restrict__ sigma, int clamp)
{
	constexpr int CH = 16 / (int)sizeof(T);
	typedef T vch __attribute__((ext_vector_type(CH)));
	const int row0 = blockIdx.x * PR_ROWS;
	const T* xr[PR_ROWS];
#pragma unroll
	for (int r = 0; r < PR_ROWS; ++r) xr[r] = X + (int64_t)min(row0 + r, m - 1) * ldx;          // (rows past the end re-read the last one; not stored)
	T s1[PR_ROWS], s2[PR_ROWS];
#pragma unroll
	for (int r = 0; r < PR_ROWS; ++r) s1[r] = s2[r] = T(0);
	if (VEC) {
		const int nch = n / CH;
#pragma unroll 2
		for (int c = threadIdx.x; c < nch; c += 256) {
			const vch zv = *(const vch*)(z + (int64_t)c * CH);
			vch xv[PR_ROWS];
#pragma unroll
			for (int r = 0; r < PR_ROWS; ++r) xv[r] = __builtin_nontemporal_load((const vch*)(xr[r] + (int64_t)c * CH));          // read once: keep z in the caches instead
#pragma unroll
			for (int r = 0; r < PR_ROWS; ++r)
#pragma unroll
				for (int e = 0; e < CH; ++e) { s1[r] += xv[r][e] * zv[e]; s2[r] += xv[r][e] * xv[r][e]; }
		}
	} else {
		for (int k = threadIdx.x; k < n; k += 256) {
			const T zk = z[k];
#pragma unroll
			for (int r = 0; r < PR_ROWS; ++r) { const T v = xr[r][k]; s1[r] += v * zk; s2[r] += v * v; }
		}
	}
	__shared__ T r1[PR_ROWS][4], r2[PR_ROWS][4];
#pragma unroll
	for (int r = 0; r < PR_ROWS; ++r) {
#pragma unroll
		for (int o = 32; o > 0; o >>= 1) { s1[r] += __shfl_xor(s1[r], o); s2[r] += __shfl_xor(s2[r], o); }
		if ((threadIdx.x & 63) == 0) { r1[r][threadIdx.x >> 6] = s1[r]; r2[r][threadIdx.x >> 6] = s2[r]; }
	}
	__syncthreads();
	if (threadIdx.x < PR_ROWS && row0 + (int)threadIdx.x < m) {
		const int r = threadIdx.x;
		const int64_t i = row0 + r;
		const T a = r1[r][0] + r1[r][1] + r1[r][2] + r1[r][3];
		const T b = r2[r][0] + r2[r][1] + r2[r][2] + r2[r][3];
		if (mu) mu[i] = a;
		if (sigma) {
			if (clamp == 2) {           // raw partial sum of squares (multi-GPU: reduced across ranks first)
				sigma[i] = b;
			} else {
				T var = kdiag[i] - b;
				if (clamp && var < T(0)) var = T(0);
				sigma[i] = sqrt(var);
			}
		}
	}
}

template <typename T>
int predict(int64_t m, int64_t n, const T* X, int64_t ldx, const T* z, const T* kdiag, T* mu, T* sigma, int clamp, hipStream_t st)
{
	if (m <= 0) return 0;
	if (n > INT32_MAX || m > INT32_MAX) { set_error("predict: dimension exceeds int32"); return -2; }
	constexpr int CH = 16 / (int)sizeof(T);
	const bool vec = n > 0 && n % CH == 0 && ldx % CH == 0 && ((((uintptr_t)X | (uintptr_t)z) & 15) == 0);
	const dim3 grid((unsigned)((m + PR_ROWS - 1) / PR_ROWS));
	if (vec) hipLaunchKernelGGL((predict_kernel<T, true>), grid, dim3(256), 0, st, X, ldx, (int)m, (int)n, z, kdiag, mu, sigma, clamp);
	else hipLaunchKernelGGL((predict_kernel<T, false>), grid, dim3(256), 0, st, X, ldx, (int)m, (int)n, z, kdiag, mu, sigma, clamp);
	return check_launch("predict");
}

// Second half of the prediction epilogue when X is column-sharded across ranks: the partial sums <X_i, z> and <X_i, X_i>
// have been all-reduced by the caller; every process row held a replica, hence `scale` = 1 / P_r.
template <typename T>
__global__ __launch_bounds__(256)
void predict_finish_kernel(int64_t m, T* __restrict__ mu, const T* __restrict__ sumsq, const T* __restrict__ kdiag, T scale, T* __restrict__ sigma, int clamp)
{
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= m) return;
	if (mu) mu[i] *= scale;
	if (sigma) {
		T var = kdiag[i] - scale * sumsq[i];
		if (clamp && var < T(0)) var = T(0);
		sigma[i] = sqrt(var);
	}
}

template <typename T>
int predict_finish(int64_t m, T* mu, const T* sumsq, const T* kdiag, double scale, T* sigma, int clamp, hipStream_t st)
{
	if (m <= 0) return 0;
	hipLaunchKernelGGL((predict_finish_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st, m, mu, sumsq, kdiag, (T)scale, sigma, clamp);
	return check_launch("predict_finish");
}

// ------------------------------------------------------------------------------------------
// out2[0] = sum log L_ii, out2[1] = z^T z.  Single workgroup, fixed summation order (bitwise
// reproducible run to run).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(1024)
void logdet_quad_kernel(const T* __restrict__ L, int64_t ldl, const T* __restrict__ z, int n, T* __restrict__ out2)
{
	T s1 = T(0), s2 = T(0);
	for (int i = threadIdx.x; i < n; i += 1024) {
		s1 += log(L[(int64_t)i * ldl + i]);
		if (z) { const T v = z[i]; s2 += v * v; }
	}
	__shared__ T r1[16], r2[16];
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
	if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = s1; r2[threadIdx.x >> 6] = s2; }
	__syncthreads();
	if (threadIdx.x == 0) {
		s1 = s2 = T(0);
		for (int w = 0; w < 16; ++w) { s1 += r1[w]; s2 += r2[w]; }
		out2[0] = s1; out2[1] = s2;
	}
}

template <typename T>
int logdet_quad(int64_t n, const T* L, int64_t ldl, const T* z, T* out2, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("logdet_quad: n exceeds int32"); return -2; }
	hipLaunchKernelGGL((logdet_quad_kernel<T>), dim3(1), dim3(1024), 0, st, L, ldl, z, (int)n, out2);
	return check_launch("logdet_quad");
}

// ------------------------------------------------------------------------------------------
// A[i][j] = A[j][i] for j > i, 64x64 tiles through LDS so both sides are coalesced.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void symmetrize_kernel(T* __restrict__ A, int64_t lda, int n)
{
	__shared__ T tile[64][65];
	const int ti = blockIdx.y, tj = blockIdx.x;       // destination tile (row block ti, col block tj), tj >= ti
	if (tj < ti) return;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	for (int r = ty; r < 64; r += 4) {                // source tile (tj, ti): rows tj*64.., cols ti*64..
		const int gr = tj * 64 + r, gc = ti * 64 + tx;
		tile[r][tx] = (gr < n && gc < n) ? A[(int64_t)gr * lda + gc] : T(0);
	}
	__syncthreads();
	for (int r = ty; r < 64; r += 4) {
		const int gr = ti * 64 + r, gc = tj * 64 + tx;
		if (gr < n && gc < n && gc > gr) A[(int64_t)gr * lda + gc] = tile[tx][r];
	}
}

template <typename T>
int symmetrize_lower(int64_t n, T* A, int64_t lda, hipStream_t st)
{
	if (n <= 0) return 0;
	if (n > INT32_MAX) { set_error("symmetrize: n exceeds int32"); return -2; }
	const unsigned t = (unsigned)((n + 63) / 64);
	hipLaunchKernelGGL((symmetrize_kernel<T>), dim3(t, t), dim3(256), 0, st, A, lda, (int)n);
	return check_launch("symmetrize_lower");
}

#define INST(T) \
	template int trsm_right_lt<T>(int64_t, int64_t, const T*, int64_t, const T*, T*, int64_t, int, hipStream_t, bool, T*, int); \
	template int potri_lower<T>(int64_t, const T*, int64_t, const T*, T*, int64_t, T*, hipStream_t); \
	template int trsv<T>(int64_t, const T*, int64_t, const T*, T*, T*, int, hipStream_t); \
	template int predict<T>(int64_t, int64_t, const T*, int64_t, const T*, const T*, T*, T*, int, hipStream_t); \
	template int logdet_quad<T>(int64_t, const T*, int64_t, const T*, T*, hipStream_t); \
	template int predict_finish<T>(int64_t, T*, const T*, const T*, double, T*, int, hipStream_t); \
	template int symmetrize_lower<T>(int64_t, T*, int64_t, hipStream_t);
INST(double)
INST(float)

}  // namespace stpy
