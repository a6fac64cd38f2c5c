// gemm.hip -- C (op)= A * B^T on the 16x16x4 MFMA (fp64 / fp32), the dense contraction under the
// blocked Cholesky (trailing SYRK/GEMM update, panel TRSM-as-GEMM) and the triangular solves.
//
// Shape of the problem on MI355X: v_mfma_f64_16x16x4_f64 retires 2048 flop per 64 SIMD-cycles,
// so a wave that owns a 64x64 accumulator (16 MFMA tiles) needs only 8 operand fragments per
// k-step: LDS traffic is ~1/16 of its bandwidth and the kernel is MFMA-issue bound.  What has to be
// right is (a) operands staged through LDS in K-contiguous rows so both A and B (both stored
// "row x K") use the same coalesced 16-byte loads, (b) two workgroups per CU (2 waves per SIMD) so
// one workgroup's barrier / global-load latency hides under the other's MFMAs, (c) the block ->
// tile map: 8x8 super-tiles, one super-tile per XCD at a time, so the 8+8 operand strips a
// super-tile needs are served from that XCD's 4 MiB L2 instead of HBM.
//
// Staging: the aligned fp64 path moves operand tiles HBM/L2 -> LDS with LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write instructions -- measured: the 8 ds_write_b128
// per wave and K tile of a register-staged loop cost 8 % of the kernel).  An LDS-DMA wave-instruction
// writes 1 KiB linearly (8 tile rows of 128 B), so rows cannot be padded; bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle applied on the SOURCE address and on
// the read (guide rule 21): 16-byte chunk c of tile row r is stored at chunk c ^ f(r),
// f(r) = ((r>>1)&3)<<1 | (r>>3)&1  -- conflict-free for all four 16-lane groups of ds_read_b128.
// Ragged shapes and fp32 keep the register-staged path (padded rows).
//
// Lane/k mapping: MFMA lane l feeds A[row = l&15][k-slot = l>>4].  The sum over k is order
// independent, so lane group g = l>>4 takes the four *consecutive* k values 4g..4g+3 of a 16-deep
// K tile (two 16-byte LDS reads for f64) and MFMA number s of the tile uses element s of every
// lane: k-slots {s, 4+s, 8+s, 12+s}.  A and B use the same map, which is all that is required.
#include "common.h"
#include <atomic>
#include <type_traits>

namespace stpy {

constexpr int BM = 128, BN = 128, NTHREADS = 256;
// K tile: one 128-byte row per operand row (16 doubles / 32 floats), so both types share the LDS byte layout,
// the LDS-DMA pieces and the swizzle
template <typename T> struct KTile { static constexpr int BK = 128 / (int)sizeof(T); };
// g_gemm_stagger = 40000:     first-round offset (cycles) between the two workgroups of a CU in launches of >= 4096 tiles (stpy_tune key 0; 0 = off, 1 = half a tile)
// g_gemm_tri_diag_last = 0:   stpy_tune key 20: lower-triangular launches enumerate their diagonal super-tiles last (0 = row-major triangle, the default:
                               // tools/potrf_sweep.py "20=0|1" shows no difference at any size -- 1370.07 vs 1370.03 ms at N = 65 536)
// g_gemm_dtv = 1:             A operand direct to VGPR for aligned fp64 products with at least this many tiles (stpy_tune key 6; 0 = never)
// g_gemm_dtv_min_k = 64
int g_gemm_bf3 = 64;             // fp32: aligned plain products with at least this many tiles run on the bf16 matrix cores (stpy_tune key 26; 0 = never)
int g_gemm_k128 = 768;           // K = 128 products with at most this many 64 x 64 tiles take the one-volley kernel (stpy_tune key 8; 0 = never)
// g_gemm_exp = 0:             timing experiments only, lab build (results are wrong when != 0)
#if STPY_LAB
#define STPY_EXP(p) ((p).exp)
#else
#define STPY_EXP(p) 0
#endif
constexpr int ST = 8;   // super-tile edge in tiles (64 tiles = the 64 workgroups one XCD holds at 2 per CU)

constexpr int BC_MAX_ROWS = 64;
template <typename T>
struct GemmArgs {
	const T* A; const T* B; T* C; T* C2;
	int64_t lda, ldb, ldc, ldc2;
	int m, n, k;
	int tiles_m, tiles_n;
	int st_m, st_n;          // super-tile shape in tiles (st_m * st_n == 64)
	int nst_m, nst_n;        // super-tile grid
	int nsuper;              // number of super-tiles enumerated
	int mode;                // 0: C = AB^T   1: C -= AB^T   2: C = rff(AB^T) (fused random-Fourier-feature epilogue)
	int epi_half;            // mode 2: columns < epi_half take cos, the rest sin (all cos when epi_bias != null)
	T epi_scale;             // mode 2: output scale sqrt(2/m) sqrt(kappa)
	const T* epi_bias;       // mode 2: optional phase per column
	const T* epi_fscale;     // mode 2: optional amplitude per feature (quadrature weights), multiplies epi_scale
	int epi_by_row;          // mode 2: features run along the ROWS of C (transposed embedding Phi^T)
	// mode 3: Gram epilogue  C[j][i] (op)= kappa * phi(nb[j] + na[i] - 2 acc) (+ offset) + diag_add [i == j]
	const T* g_na; const T* g_nb;
	T g_kappa, g_offset, g_diag;
	int g_kind, g_combine;
	// mode 4: evidence-gradient weight, in place over a symmetric K^-1:
	//   C[j][i] = (g_w * C[j][i] - g_alpha[j] * g_alpha[i]) * F_kind(scaled squared distance)
	const T* g_alpha; T g_w;
	int kskip;               // 1: operands are upper triangular (B B^T of an inverse factor): K range of tile row ti starts at its first row
	int tri;                 // 1: lower-triangular tile set (square C), super-tiles enumerated over the lower triangle
	int stagger;             // >0: first-round workgroups in the odd wave slot of their SIMD start this many cycles late
	// block-cyclic "staircase" (multi-GPU local trailing update): C is a window of a rank's local
	// matrix; tile (ti,tj) lies in distribution block (ti/bc_nbt + bc_i0, tj/bc_nbt + bc_j0) of
	// the local matrix = global block (I, J) = (.. * bc_pr + bc_myr, .. * bc_pc + bc_myc); tiles
	// with I < J (strictly above the global diagonal) are skipped.  bc_nbt == 0: off.
	int bc_nbt, bc_pr, bc_pc, bc_myr, bc_myc, bc_i0, bc_j0;
	// compact enumeration of the staircase: bc_pref[si] = number of non-empty super-tiles in super-tile rows < si (rows are
	// non-empty from the left); bc_compact == 0: the full rectangle is enumerated and empty super-tiles exit at once
	int bc_compact; int bc_pref[BC_MAX_ROWS + 1];
	int exp;                 // timing experiments (0 in production)
	// split-K (few output tiles, long K): the super-tile range is enumerated ksplit times; pass s
	// contracts K range [s*kchunk, (s+1)*kchunk) into the partial result at C + s*split_stride
	int ksplit, kchunk; int64_t split_stride;
#ifdef STPY_STAMPS
	unsigned long long* dbg;      // diagnostic builds (-DSTPY_STAMPS) only: in-kernel time stamps, never read by any kernel
#endif
};
#ifdef STPY_STAMPS
unsigned long long* g_gemm_dbg = nullptr;
extern "C" __attribute__((visibility("default"))) void stpy_debug_set_stamp_buffer(void* p) { g_gemm_dbg = (unsigned long long*)p; }
#endif

// Random-Fourier-feature epilogue: scale * cos(q + b) or scale * sin(q).  fp64: libm-accurate.
// fp32: the phase is reduced to revolutions in fp32 (q/2pi minus its nearest integer, exact for
// |q/2pi| < 2^22) and fed to the hardware v_sin_f32 / v_cos_f32 (input in revolutions); absolute
// error ~1e-6 on a value that is then scaled by sqrt(2/m).
template <typename T> __device__ __forceinline__ T rff_value(T q, bool use_cos, T scale);
template <> __device__ __forceinline__ double rff_value<double>(double q, bool, double)
{
	return q;       // never used: fp64 takes the unfused path (see the epilogue note)
}
template <> __device__ __forceinline__ float rff_value<float>(float q, bool use_cos, float scale)
{
	float t = q * 0.15915494309189535f;
	t -= rintf(t);
	return scale * (use_cos ? __builtin_amdgcn_cosf(t) : __builtin_amdgcn_sinf(t));
}

// exp(x) for the Gram epilogue, x <= ~0.  Written out (no libm call): 2^k * p(r), r = x - k ln2 in
// two steps, degree-13 Taylor on |r| <= 0.347 (remainder 4e-18), v_ldexp for the scaling; ~20
// instructions, no branches, <= 2 ulp -- small enough that the epilogue loops below stay unrolled.
__device__ __forceinline__ double gram_exp(double x)
{
	x = fmax(x, -745.0);
	const double k = rint(x * 1.4426950408889634074);
	double r = fma(-k, 6.93147180369123816490e-01, x);
	r = fma(-k, 1.90821492927058770002e-10, r);
	double p = 1.6059043836821613e-10;                 // 1/13!
	p = fma(p, r, 2.08767569878681e-09);               // 1/12!
	p = fma(p, r, 2.505210838544172e-08);              // 1/11!
	p = fma(p, r, 2.755731922398589e-07);              // 1/10!
	p = fma(p, r, 2.7557319223985893e-06);             // 1/9!
	p = fma(p, r, 2.48015873015873e-05);               // 1/8!
	p = fma(p, r, 1.984126984126984e-04);              // 1/7!
	p = fma(p, r, 1.388888888888889e-03);              // 1/6!
	p = fma(p, r, 8.333333333333333e-03);              // 1/5!
	p = fma(p, r, 4.1666666666666664e-02);             // 1/4!
	p = fma(p, r, 1.6666666666666666e-01);             // 1/3!
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	return ldexp(p, (int)k);
}
__device__ __forceinline__ float gram_exp(float x) { return __expf(x); }
// Table-assisted form for the fused Gram epilogue, where the 20 instructions above made the fill VALU-bound (32 fp64 instructions
// per element = a 4.9 TB/s ceiling, profiles/r02_a_gram_rff_pmc.json):  x = (k / 256) ln2 + r with |r| <= ln2 / 512, so
// e^x = 2^(k >> 8) * tab[k & 255] * (1 + r + r^2/2 + r^3/6 + r^4/24)  (truncation 4e-17) -- 11 fp64 instructions, two integer ones
// and an LDS read.  tab[i] = 2^(i/256): 2 KiB of LDS filled by the workgroup with the function above.
constexpr int GRAM_TAB = 256;
__device__ __forceinline__ double gram_exp_tab(double x, const double* tab)
{
	x = fmax(x, -745.0);
	// k = rint(x * 256 / ln2) by the magic-number addition (round-to-nearest puts the integer into the low mantissa bits, two's
	// complement in the low word): one FMA + one subtraction instead of multiply, round and convert
	const double kd = fma(x, 369.3299304675746, 0x1.8p52);
	const int ki = __double2loint(kd);
	const double k = kd - 0x1.8p52;
	double r = fma(-k, 0x1.62e42ffp-9, x);               // ln2 / 256 = hi + lo, hi with 29 significant bits: k * hi is exact
	r = fma(-k, -1.6409824502660487e-13, r);
	double q = fma(r, 4.1666666666666664e-02, 1.6666666666666666e-01);
	q = fma(q, r, 0.5);
	q = fma(q, r, 1.0);
	q *= r;                                               // e^r - 1
	const double t = tab[ki & (GRAM_TAB - 1)];
	return ldexp(fma(t, q, t), ki >> 8);
}
__device__ __forceinline__ float gram_exp_tab(float x, const float*) { return __expf(x); }

// x: -0.5 |a - b|^2 straight from the accumulator (the contraction STARTS from -(|a|^2 + |b|^2) / 2, see the accumulator set-up of
// the mode-3 kernel); tab: gram_exp_tab's table
template <typename T, int KIND> __device__ __forceinline__ T gram_value(T x, const T* tab)
{
	if (KIND == STPY_K_LINEAR) return x;                             // (started from zero: the plain dot product)
	if (KIND == STPY_K_SE) return gram_exp_tab(x, tab);              // no clamp, as kernels.py:395
	const T rr = sqrt(fmax(T(-2) * x, T(0)));
	if (KIND == STPY_K_MATERN32) { const T r = rr * T(1.7320508075688772935); return (T(1) + r) * gram_exp_tab(-r, tab); }
	const T r = rr * T(2.2360679774997896964);                       // MATERN52
	return (T(1) + r + r * r * T(0.33333333333333333333)) * gram_exp_tab(-r, tab);
}

// d k / d(lengthscale_m) = F * u_m^2 / lengthscale_m with u_m the scaled coordinate difference; F per family:
template <typename T, int KIND> __device__ __forceinline__ T gram_dfactor(T acc, T na, T nb)
{
	const T sq = na + nb - T(2) * acc;
	if (KIND == STPY_K_SE) return gram_exp(T(-0.5) * sq);
	const T rr = sqrt(fmax(sq, T(0)));
	if (KIND == STPY_K_MATERN12) return rr > T(0) ? gram_exp(-rr) / rr : T(0);
	if (KIND == STPY_K_MATERN32) return T(3) * gram_exp(-rr * T(1.7320508075688772935));
	const T r = rr * T(2.2360679774997896964);                       // MATERN52
	return T(1.6666666666666666667) * (T(1) + r) * gram_exp(-r);
}

// ------------------------------------------------------------------------------------------
// Dedicated fp64 Gram fill (aligned shapes, overwrite, SE / Matern 3/2 / 5/2 / linear): the HBM-write-bound kernel of the path.
// Through the GEMM kernel above (EPI = 3) a workgroup owns 128 x 128 outputs, 206 VGPRs and 73 KiB of LDS: two per CU, and the
// 128 KiB store phase of one overlaps only with the arithmetic of the other (7.1 us per tile pair against 5.3 us of HBM time).
// Here a workgroup owns 128 x 64 outputs (wave w: rows 32 w .. + 31, all 64 columns: 64 accumulator VGPRs), the scaled points come
// straight from L2 into registers in MFMA fragment order (they are a few MB in all: no LDS staging, no barrier in the contraction),
// and the only LDS is the exp table + a wave-private patch that turns a 16-row slab into 16-byte stores of 512 contiguous bytes per
// row -- 22 KiB and <= 128 VGPRs: four workgroups per CU, so that one is always storing.  (Measured, N = 65 536 SE: full fill 7.32 ->
// 6.87 ms with three workgroups per CU and a 16-row patch; the lower-only fill of fit_gp unchanged at 3.7 ms -- a workgroup's chain
// of point loads -> MFMAs -> exp -> stores is latency-bound, so residency is what counts: hence the 8-row patch and four per CU.)
// ------------------------------------------------------------------------------------------
struct GramFillArgs {
	const double* as; const double* bs; const double* na; const double* nb; double* out;
	int64_t ldo; int dpad, nct, lower; double kappa, offset, diag;
};
// (the Matern forms -- a square root and more live values per element -- spill 27 registers under the 128 cap: three per CU for them)
template <int KIND>
__global__ __launch_bounds__(256, (KIND == STPY_K_SE || KIND == STPY_K_LINEAR) ? 4 : 3)
void gram_fill_f64_kernel(GramFillArgs p)
{
	typedef double T;
	typedef Mfma<double> MM;
	typedef MM::v4 v4;
	typedef double d2 __attribute__((ext_vector_type(2)));
	constexpr int PLD = 80;
	__shared__ __attribute__((aligned(16))) double smem[GRAM_TAB + 4 * 8 * PLD];
	double* const tab = smem;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int r16 = lane & 15, g = lane >> 4;
	tab[tid] = gram_exp(T(tid) * T(0.693147180559945309417 / GRAM_TAB));          // 256 threads, 256 entries
	// block -> (row tile of 128, column tile of 64); lower: row tile ti holds the 2 (ti + 1) column tiles on or left of its diagonal block
	int ti, cj;
	const int b = blockIdx.x;
	if (p.lower) {
		ti = (int)((sqrt(4.0 * (double)b + 1.0) - 1.0) * 0.5);
		while ((ti + 1) * (ti + 2) <= b) ++ti;
		while (ti * (ti + 1) > b) --ti;
		cj = b - ti * (ti + 1);
	} else {
		ti = b / p.nct;
		cj = b - ti * p.nct;
	}
	ti = __builtin_amdgcn_readfirstlane(ti);
	cj = __builtin_amdgcn_readfirstlane(cj);
	const int row0 = ti * 128 + wave * 32, col0 = cj * 64;

	// ---- accumulators start from -(|b_j|^2 + |a_i|^2) / 2 (zero for the linear kernel): they end as -|a_i - b_j|^2 / 2
	v4 acc[2][4];
	double ha[4];
#pragma unroll
	for (int tn = 0; tn < 4; ++tn) ha[tn] = KIND == STPY_K_LINEAR ? 0.0 : -0.5 * p.na[col0 + tn * 16 + r16];
#pragma unroll
	for (int tm = 0; tm < 2; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const double hb = KIND == STPY_K_LINEAR ? 0.0 : -0.5 * p.nb[row0 + tm * 16 + MM::crow(lane, i)];
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = ha[tn] + hb;
		}
	// ---- contraction over the (padded) coordinates: lane (r16, g) holds k = 4 g .. 4 g + 3 of its rows; MFMA s uses element s
	const double* const bsp = p.bs + (int64_t)(row0 + r16) * p.dpad + 4 * g;
	const double* const asp = p.as + (int64_t)(col0 + r16) * p.dpad + 4 * g;
	for (int k0 = 0; k0 < p.dpad; k0 += 16) {
		d2 fb[2][2], fa[4][2];
#pragma unroll
		for (int tm = 0; tm < 2; ++tm) { fb[tm][0] = *(const d2*)(bsp + (int64_t)tm * 16 * p.dpad + k0); fb[tm][1] = *(const d2*)(bsp + (int64_t)tm * 16 * p.dpad + k0 + 2); }
#pragma unroll
		for (int tn = 0; tn < 4; ++tn) { fa[tn][0] = *(const d2*)(asp + (int64_t)tn * 16 * p.dpad + k0); fa[tn][1] = *(const d2*)(asp + (int64_t)tn * 16 * p.dpad + k0 + 2); }
#pragma unroll
		for (int s = 0; s < 4; ++s)
#pragma unroll
			for (int tm = 0; tm < 2; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					acc[tm][tn] = MM::mma(fb[tm][s >> 1][s & 1], fa[tn][s >> 1][s & 1], acc[tm][tn]);
	}
	__syncthreads();          // the exp table is complete
	// ---- epilogue, one 16-row slab at a time: kernel function, diagonal term, then the slab's stores (they drain under the next slab's arithmetic)
	double* const patch = smem + GRAM_TAB + wave * (8 * PLD);
	const int prow = lane >> 5, pcol = (lane & 31) * 2;
	const bool on_diag = p.diag != 0.0 && col0 >= ti * 128 && col0 < ti * 128 + 128;
#pragma unroll
	for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
		for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
			for (int i = 0; i < 4; ++i) acc[tm][tn][i] = p.kappa * gram_value<double, KIND>(acc[tm][tn][i], tab) + p.offset;
			__builtin_amdgcn_sched_barrier(0);          // four independent exp chains at a time (see the GEMM epilogue)
		}
		if (on_diag) {
#pragma unroll
			for (int tn = 0; tn < 4; ++tn)
#pragma unroll
				for (int i = 0; i < 4; ++i)
					if (row0 + tm * 16 + MM::crow(lane, i) == col0 + tn * 16 + r16) acc[tm][tn][i] += p.diag;
		}
		// the slab leaves in two halves of 8 rows (registers i = 0, 1: rows g, g + 4; i = 2, 3: rows g + 8, g + 12) through an
		// 8-row patch: 16-byte stores, 512 contiguous bytes per row, two rows per instruction
		double* const orow = p.out + (int64_t)(row0 + tm * 16) * p.ldo + col0 + pcol;
#pragma unroll
		for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
			for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) patch[(g + 4 * i2) * PLD + tn * 16 + r16] = acc[tm][tn][2 * hf + i2];
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int rr = q * 2 + prow;
				const d2 v = *(const d2*)&patch[rr * PLD + pcol];
				__builtin_nontemporal_store(v, (d2*)(orow + (int64_t)(8 * hf + rr) * p.ldo));
			}
		}
		__builtin_amdgcn_sched_barrier(0);
	}
}

int g_potrf_presplit = 1;               // stpy_tune route key 32: the fp32 factorisation splits each panel once into bf16 planes and updates from those (gemm_bf3p.hip; 0 = every tile splits on the fly)
int g_gemm_sliver_tiles = 3200;         // stpy_tune route key 30: plain / lower-only fp64 products of at most this many 128 x 128 tiles take the 32 x 128 sliver kernel (0 = never)
int g_gram_fill = 1;          // stpy_tune route key 28: 1 = the dedicated fp64 fill kernel for aligned overwriting fills, 0 = always the GEMM epilogue

// returns 1 when the dedicated kernel took the fill, 0 when the shape / options are not its, < 0 on a launch error
int gram_fill_f64(int kind, const double* as, const double* bs, const double* na, const double* nb, int dpad, int64_t n, int64_t q,
                  double kappa, double offset, double diag_add, int lower_only, int combine, double* out, int64_t ldo, hipStream_t st)
{
	if (!g_gram_fill || combine != STPY_OUT_SET || n % 128 != 0 || q % 128 != 0 || dpad % 16 != 0 || ldo % 2 != 0 || ((((uintptr_t)out) | ((uintptr_t)as) | ((uintptr_t)bs)) & 15) != 0 ||
	    n >= ((int64_t)1 << 30) || q >= ((int64_t)1 << 30) || (lower_only && n != q)) return 0;
	if (kind != STPY_K_SE && kind != STPY_K_MATERN32 && kind != STPY_K_MATERN52 && kind != STPY_K_LINEAR) return 0;
	GramFillArgs p{as, bs, na, nb, out, ldo, dpad, (int)(n / 64), lower_only ? 1 : 0, kappa, kind == STPY_K_LINEAR ? offset : 0.0, diag_add};
	const int64_t qt = q / 128;
	const int64_t blocks = lower_only ? qt * (qt + 1) : qt * (n / 64);
	if (blocks > INT32_MAX) return 0;
	const dim3 grid((unsigned)blocks), block(256);
	switch (kind) {
	case STPY_K_SE: hipLaunchKernelGGL(gram_fill_f64_kernel<STPY_K_SE>, grid, block, 0, st, p); break;
	case STPY_K_MATERN32: hipLaunchKernelGGL(gram_fill_f64_kernel<STPY_K_MATERN32>, grid, block, 0, st, p); break;
	case STPY_K_MATERN52: hipLaunchKernelGGL(gram_fill_f64_kernel<STPY_K_MATERN52>, grid, block, 0, st, p); break;
	default: hipLaunchKernelGGL(gram_fill_f64_kernel<STPY_K_LINEAR>, grid, block, 0, st, p); break;
	}
	const int rc = check_launch("gram fill");
	return rc ? rc : 1;
}

// EPI selects the fused store epilogue at COMPILE time (0 none, 2 RFF trig, 3 Gram kernel function,
// 4 evidence-gradient weight): the heavy epilogues must not share an instantiation with the plain
// contraction -- their code raises register pressure enough to push the accumulators of the whole
// kernel into scratch.
// ACC: 0 overwrite (C = ...), 1 subtract (C -= A B^T), 2 add (C += A B^T: the slab-wise accumulation of Phi^T Phi)
template <typename T, bool GUARD, int ACC, int EPI>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_nt_kernel(GemmArgs<T> p)
{
	constexpr bool SUB = ACC == 1, LOADC = ACC != 0;
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	typedef typename MM::v2 v2;
	constexpr int CH = 16 / sizeof(T);        // elements per 16-byte chunk
	constexpr int BK = KTile<T>::BK;
	constexpr int CPR = BK / CH;              // chunks per tile row
	constexpr int RPP = NTHREADS / CPR;       // rows staged per pass
	constexpr int NP = BM / RPP;              // passes per operand
	constexpr int PAD = 16 / sizeof(T);       // one 16-byte chunk of padding per LDS row
	constexpr int LLD = BK + PAD;
	typedef T vch __attribute__((ext_vector_type(CH)));
	constexpr bool DMA = !GUARD;                        // LDS-DMA staging (unpadded, swizzled rows)
	constexpr int RLD = DMA ? BK : LLD;                 // LDS row stride actually used

	__shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * LLD];
	T* As = smem;                         // [2][BM][RLD]
	T* Bs = smem + 2 * BM * RLD;          // [2][BN][RLD]

	// ---- block -> tile.  Blocks b, b+8, b+16.. share an XCD (round-robin dispatch; speed only):
	// ---- super-tile S = (b % 8) + 8 * (b / 512), tile within it = (b / 8) % 64.
	const int b = blockIdx.x;
	int S = (b & 7) + 8 * (b >> 9);
	const int w = (b >> 3) & 63;
	int split = 0;
	if (p.ksplit > 1) {
		split = __builtin_amdgcn_readfirstlane(S / p.nsuper);
		S -= split * p.nsuper;
		if (split >= p.ksplit) return;
	} else if (S >= p.nsuper) return;
	int si, sj;
	if (p.tri == 2) {
		// lower triangle, the DIAGONAL super-tiles (36 of 64 tiles) last, so that they form dispatch rounds of equal weight (experiment,
		// see g_gemm_tri_diag_last: unlike the empty super-tiles of the block-cyclic staircase they cost nothing measurable in place)
		const int noff = p.nst_m * (p.nst_m - 1) / 2;
		if (S >= noff) { si = sj = S - noff; }
		else {
			si = (int)((sqrt(8.0 * (double)S + 1.0) + 1.0) * 0.5);          // strict lower triangle: S = si (si - 1) / 2 + sj, sj < si
			while (si * (si + 1) / 2 <= S) ++si;
			while (si * (si - 1) / 2 > S) --si;
			sj = S - si * (si - 1) / 2;
		}
	} else if (p.tri) {
		si = (int)((sqrt(8.0 * (double)S + 1.0) - 1.0) * 0.5);
		while ((si + 1) * (si + 2) / 2 <= S) ++si;
		while (si * (si + 1) / 2 > S) --si;
		sj = S - si * (si + 1) / 2;
	} else if (p.bc_compact) {
		// block-cyclic staircase, compact: only the super-tiles that hold work are enumerated (host-built prefix table in the
		// kernel arguments).  Enumerating the rectangle and letting the empty ones exit costs 13-32 % (tools/bc_vs_tri.py): workgroups
		// are dispatched in order and each goes to XCD b % 8, so an XCD that drew empty super-tiles idles behind the others' full ones.
		int lo = 0, hi = p.nst_m;                 // largest si with bc_pref[si] <= S
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.bc_pref[mid] <= S) lo = mid; else hi = mid; }
		si = lo;
		sj = S - p.bc_pref[si];
	} else {
		si = S / p.nst_n;
		sj = S - si * p.nst_n;
		// (block-cyclic staircase on more super-tile rows than the prefix table holds: rectangle, rows rotated by their index so that
		// every XCD gets a mix of long and short columns)
		if (p.bc_nbt > 0) { sj += si % p.nst_n; if (sj >= p.nst_n) sj -= p.nst_n; }
	}
	// (the tile index comes out of VALU arithmetic; readfirstlane tells the compiler it is uniform,
	// so tile bases live in SGPRs and per-lane addresses stay 32-bit offsets)
	const int ti = __builtin_amdgcn_readfirstlane(si * p.st_m + w / p.st_n);
	const int tj = __builtin_amdgcn_readfirstlane(sj * p.st_n + w % p.st_n);
	if (ti >= p.tiles_m || tj >= p.tiles_n) return;
	if (p.tri && tj > ti) return;
	if (p.bc_nbt > 0) {
		const int I = (ti / p.bc_nbt + p.bc_i0) * p.bc_pr + p.bc_myr;
		const int J = (tj / p.bc_nbt + p.bc_j0) * p.bc_pc + p.bc_myc;
		if (I < J) return;
		if (I == J && (tj % p.bc_nbt) > (ti % p.bc_nbt)) return;          // diagonal distribution block: its lower tiles only (4.7 % of the flops at 64 blocks)
	}

	const int row0 = ti * BM, col0 = tj * BN;
	const int kbeg = p.kskip ? row0 : split * p.kchunk;        // (multiples of the tile size, so K tiles stay aligned)
	const int kend = p.ksplit > 1 ? min(p.k, kbeg + p.kchunk) : p.k;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;
	const int r16 = lane & 15, g = lane >> 4;

	// ---- global -> register staging map: thread owns chunk `lch` of rows lrow + pass*RPP
	const int lrow = tid / CPR, lch = tid % CPR;
	vch ra[NP], rb[NP];

	auto gload = [&](int k0) {
#pragma unroll
		for (int q = 0; q < NP; ++q) {
			const int r = lrow + q * RPP;
			if (!GUARD) {
				ra[q] = *(const vch*)(p.A + (int64_t)(row0 + r) * p.lda + k0 + lch * CH);
				rb[q] = *(const vch*)(p.B + (int64_t)(col0 + r) * p.ldb + k0 + lch * CH);
			} else {
				const int ar = min(row0 + r, p.m - 1), br = min(col0 + r, p.n - 1);
#pragma unroll
				for (int e = 0; e < CH; ++e) {
					const int kk = k0 + lch * CH + e;
					const bool ok = kk < kend;
					const int kc = ok ? kk : 0;
					T va = p.A[(int64_t)ar * p.lda + kc];
					T vb = p.B[(int64_t)br * p.ldb + kc];
					ra[q][e] = ok ? va : T(0);
					rb[q][e] = ok ? vb : T(0);
				}
			}
		}
	};
	// SUB (C -= A B^T): accumulate (-A) B^T on top of C
	auto lstore = [&](int buf) {
#pragma unroll
		for (int q = 0; q < NP; ++q) {
			const int r = lrow + q * RPP;
			*(vch*)(As + (buf * BM + r) * LLD + lch * CH) = (SUB && !DMA) ? -ra[q] : ra[q];
			*(vch*)(Bs + (buf * BN + r) * LLD + lch * CH) = rb[q];
		}
	};

	// ---- LDS-DMA staging: wave w moves tile rows [32w, 32w+32) of A and of B, 8 rows (1 KiB) per
	// ---- instruction; lane -> (row = lane>>3, physical chunk = lane&7), source chunk = phys ^ f(row)
	const T* dma_a[4];
	const T* dma_b[4];
	if (DMA) {
		const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int r = wv * 32 + i * 8 + (lane >> 3);
			const int f = (((r >> 1) & 3) << 1) | ((r >> 3) & 1);
			const int c = (lane & 7) ^ f;
			dma_a[i] = p.A + (int64_t)(row0 + r) * p.lda + c * CH;
			dma_b[i] = p.B + (int64_t)(col0 + r) * p.ldb + c * CH;
		}
	}
	// One LDS-DMA wave-instruction, issued through inline asm ON PURPOSE: with the builtin, hipcc
	// models the DMA as an LDS store and (in some instantiations of this kernel) orders the next
	// ds_read behind it with s_waitcnt vmcnt(0) -- the prefetch is then drained before the MFMAs
	// start and the whole pipeline serialises (measured 5x slower).  The asm form is invisible to
	// that bookkeeping; the matching wait is the explicit vmcnt(0) in front of the barrier below.
	// M0 (LDS destination base) is saved and restored inside the statement (guide section 5.7).
	auto dma_one = [&](const T* gsrc, T* ldst) {
		const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)ldst;
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(gsrc), "s"(laddr) : "memory");
	};
	auto dma_issue = [&](int buf, int k0) {
		const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			dma_one(dma_a[i] + k0, As + (buf * BM + wv * 32 + i * 8) * BK);
			dma_one(dma_b[i] + k0, Bs + (buf * BN + wv * 32 + i * 8) * BK);
		}
	};
	auto dma_wait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

	// ---- desynchronise the two workgroups that share a CU (see the note at the launch site)
	if (p.stagger > 0 && b < 512) {
		const unsigned hwid = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID[3:0] = wave slot in its SIMD
		if (hwid & 1u) {
			const unsigned long long t0 = __builtin_amdgcn_s_memtime();
			while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
		}
	}

	const int KT = (kend - kbeg + BK - 1) / BK;
	if (DMA) dma_issue(0, kbeg);
	else gload(kbeg);

	// ---- accumulators: zero, or the C tile itself when subtracting (its load overlaps the first
	// ---- operand tile's; the epilogue is then store-only)
	// Addressing: uniform tile base (SGPRs) + one 32-bit element offset per accumulator row; the
	// four column tiles of a row are immediate offsets, so 16 VGPRs address all 64 elements.
	v4 acc[4][4];
	T* const ctile = p.C + (int64_t)split * p.split_stride + (int64_t)row0 * p.ldc + col0;
	const unsigned ldc32 = (unsigned)p.ldc;
	// Gram mode: the contraction starts from -(|a_i|^2 + |b_j|^2) / 2, so the accumulator ends as -|a_i - b_j|^2 / 2 and the
	// epilogue spends no instruction on the norm expansion (zero for the linear kernel: the plain dot product)
	T gram_ha[4] = {T(0), T(0), T(0), T(0)};
	if constexpr (EPI == 3) {
		if (p.g_kind != STPY_K_LINEAR) {
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) {
				const int col = col0 + wn * 64 + tn * 16 + r16;
				gram_ha[tn] = T(-0.5) * p.g_na[GUARD ? min(col, p.n - 1) : col];
			}
		}
	}
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
			const int lr_c = GUARD ? min(lr, p.m - 1 - row0) : lr;
			const T* const crow = ctile + ((unsigned)lr_c * ldc32 + (unsigned)(wn * 64 + r16));
			T gram_hb = T(0);
			if constexpr (EPI == 3) { if (p.g_kind != STPY_K_LINEAR) gram_hb = T(-0.5) * p.g_nb[row0 + lr_c]; }
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) {
				T v = (EPI == 3) ? gram_ha[tn] + gram_hb : T(0);
				if (LOADC) {      // unconditional loads (clamped address when ragged): no per-element branches
					if (!GUARD) v = crow[tn * 16];
					else v = ctile[(unsigned)lr_c * ldc32 + (unsigned)min(wn * 64 + r16 + tn * 16, p.n - 1 - col0)];
				}
				acc[tm][tn][i] = (SUB && DMA) ? -v : v;       // DMA path: accumulate +A B^T on -C, negate back at the store
			}
		}

	if (!DMA) lstore(0);
	else dma_wait();
	__syncthreads();
	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);     // read-side swizzle of this lane's rows (row & 15 == r16)
	int buf = 0;
	for (int kt = 0; kt < KT; ++kt) {
		if (kt + 1 < KT && !(STPY_EXP(p) & 1)) {
			if (DMA) dma_issue(buf ^ 1, kbeg + (kt + 1) * BK);
			else gload(kbeg + (kt + 1) * BK);
		}
		const T* as = As + (buf * BM + wm * 64 + r16) * RLD;
		const T* bs = Bs + (buf * BN + wn * 64 + r16) * RLD;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			// keep the two halves' fragments from being live together (32 instead of 64 VGPRs)
			if (h == 1) __builtin_amdgcn_sched_barrier(0);
			// lane group g owns 16-byte chunks 2g and 2g+1 of its rows: CH consecutive k each, one MFMA per k
			vch fa[4], fb[4];
			const int hoff = ((2 * g + h) ^ (DMA ? fsw : 0)) * CH;
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				fa[t] = *(const vch*)(as + t * 16 * RLD + hoff);
				fb[t] = *(const vch*)(bs + t * 16 * RLD + hoff);
			}
#pragma unroll
			for (int s = 0; s < CH; ++s)
#pragma unroll
				for (int tm = 0; tm < 4; ++tm)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						acc[tm][tn] = MM::mma(fa[tm][s], fb[tn][s], acc[tm][tn]);
		}
		if (!DMA && kt + 1 < KT && !(STPY_EXP(p) & 2)) lstore(buf ^ 1);
		if (DMA) dma_wait();       // this wave's pieces of the next tile have landed; the barrier publishes them
		if (!(STPY_EXP(p) & 4)) __syncthreads();
		if (!(STPY_EXP(p) & 8)) buf ^= 1;
	}

	// ---- fused random-Fourier-feature epilogue: ONE uniform branch around the whole transform (a
	// ---- per-element runtime test would make hipcc branch and wait around every element)
	// fp32 only: the fp64 libm sin/cos bodies are so large that hipcc stops unrolling the loops below,
	// indexes `acc` at run time and moves ALL accumulators to scratch -- for every use of the
	// kernel, 5x slower (guide rule 20).  fp64 embeds take the unfused route in rff.hip instead.
	if constexpr (EPI == 2) {
		if (!p.epi_by_row) {
			T bias[4], fsc[4];
			bool use_cos[4];
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) {
				const int col = col0 + wn * 64 + r16 + tn * 16;
				bias[tn] = p.epi_bias ? p.epi_bias[GUARD ? min(col, p.n - 1) : col] : T(0);
				fsc[tn] = p.epi_scale * (p.epi_fscale ? p.epi_fscale[GUARD ? min(col, p.n - 1) : col] : T(1));
				use_cos[tn] = p.epi_bias != nullptr || col < p.epi_half;
			}
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
#pragma unroll
					for (int i = 0; i < 4; ++i)
						acc[tm][tn][i] = rff_value<T>(acc[tm][tn][i] + bias[tn], use_cos[tn], fsc[tn]);
		} else {
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int row = row0 + wm * 64 + tm * 16 + MM::crow(lane, i);
					const T b = p.epi_bias ? p.epi_bias[GUARD ? min(row, p.m - 1) : row] : T(0);
					const T fs = p.epi_scale * (p.epi_fscale ? p.epi_fscale[GUARD ? min(row, p.m - 1) : row] : T(1));
					const bool uc = p.epi_bias != nullptr || row < p.epi_half;
#pragma unroll
					for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = rff_value<T>(acc[tm][tn][i] + b, uc, fs);
				}
		}
	}

	// ---- fused Gram epilogue (mode 3): norms of the (pre-scaled) points come from the workspace.
	// The kernel family is dispatched ONCE around straight-line loops (a per-element switch made
	// the epilogue 130 KB of branchy code that no longer fit the instruction cache); the kernel
	// algebra (+, *) and the diagonal term are separate, rarely taken passes.
	if constexpr (EPI == 3) {
		// the exp table lives in LDS (the staging buffers are free after the K loop)
		T* const tab = smem + 2 * BN;
		__syncthreads();          // every wave is past its last read of the staging buffers
		if constexpr (sizeof(T) == 8) tab[tid] = gram_exp(T(tid) * T(0.693147180559945309417 / GRAM_TAB));          // 256 threads, 256 entries
		__syncthreads();
		// One 16-row slab of the wave's tile at a time: kernel function, algebra, diagonal term, then its stores -- so the stores of
		// slab tm drain under the arithmetic of slab tm + 1.  (All the arithmetic first and 64 stores per lane at the end: 8.5 ms at
		// N = 65 536 for 5.1 ms of arithmetic and 5.4 ms of HBM write; a wave stalled at the store queue issues nothing else.)
		// Aligned tiles turn the slab through a wave-private LDS patch (behind the norms and the exp table) and store 16 bytes per
		// lane, 512 (fp64) / 256 (fp32) contiguous bytes per row; ragged ones keep the element stores.
		constexpr int PLD = 80, CHW = 16 / (int)sizeof(T), LPR = 64 / CHW, RPI = 64 / LPR;          // patch row stride 80: conflict-free writes
		typedef T vst __attribute__((ext_vector_type(CHW)));
		const bool wide = !GUARD && (((uintptr_t)p.C) & 15) == 0 && p.ldc % CHW == 0;
		T* const patch = smem + 2 * BN + GRAM_TAB + wave * (16 * PLD);
		const int prow = lane / LPR, pcol = (lane % LPR) * CHW;
		const bool algebra = p.g_combine != STPY_OUT_SET, add = p.g_combine == STPY_OUT_ADD;
		const bool on_diag = p.g_diag != T(0) && row0 == col0;          // tiles are 128-aligned: only diagonal tiles hold i == j
		const bool no_store = (STPY_EXP(p) & 16) != 0;                        // timing ablation (stpy_tune key 1 = 16): the fill without its stores
		auto finish_slab = [&](auto TM) __attribute__((always_inline)) {
			constexpr int tm = decltype(TM)::value;          // (a compile-time index: a run-time one would put the accumulators into scratch)
			if (algebra) {         // kernel algebra: out (+|*)= k, the old slab read 16 values at a time
				T old[4][4];
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
					const int lrc = GUARD ? min(lr, p.m - 1 - row0) : lr;
#pragma unroll
					for (int tn = 0; tn < 4; ++tn) {
						const int lc = wn * 64 + r16 + tn * 16;
						old[tn][i] = ctile[(unsigned)lrc * ldc32 + (unsigned)(GUARD ? min(lc, p.n - 1 - col0) : lc)];
					}
				}
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						acc[tm][tn][i] = add ? old[tn][i] + acc[tm][tn][i] : old[tn][i] * acc[tm][tn][i];
			}
			if (on_diag) {
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
#pragma unroll
					for (int i = 0; i < 4; ++i)
						if (wm * 64 + tm * 16 + MM::crow(lane, i) == wn * 64 + r16 + tn * 16) acc[tm][tn][i] += p.g_diag;
			}
			if (no_store) {
				if (acc[tm][0][0] + acc[tm][1][1] + acc[tm][2][2] + acc[tm][3][3] == T(12345.678)) ctile[0] = acc[tm][0][0];
			} else if (wide) {
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn) patch[MM::crow(lane, i) * PLD + tn * 16 + r16] = acc[tm][tn][i];
#pragma unroll
				for (int q = 0; q < 16 / RPI; ++q) {
					const int rr = q * RPI + prow;
					const vst v = *(const vst*)&patch[rr * PLD + pcol];
					__builtin_nontemporal_store(v, (vst*)(ctile + ((unsigned)(wm * 64 + tm * 16 + rr) * ldc32 + (unsigned)(wn * 64 + pcol))));
				}
			} else {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
					T* const crow = ctile + ((unsigned)lr * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
					for (int tn = 0; tn < 4; ++tn) {
						if (GUARD && (row0 + lr >= p.m || col0 + wn * 64 + r16 + tn * 16 >= p.n)) continue;
						// streaming output (far larger than the caches): bypass the L2 allocation so that the operand tiles the
						// next workgroups re-read stay resident
						__builtin_nontemporal_store(acc[tm][tn][i], &crow[tn * 16]);
					}
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		};
		auto slab = [&](auto KC, auto TM) __attribute__((always_inline)) {
			constexpr int KIND = decltype(KC)::value, tm = decltype(TM)::value;
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					acc[tm][tn][i] = p.g_kappa * gram_value<T, KIND>(acc[tm][tn][i], tab) + p.g_offset;
					// four independent exp chains at a time cover the FMA latency and the LDS round trip of the table
					// read; letting the scheduler interleave all 64 costs >250 spilled VGPRs
					if (i == 3) __builtin_amdgcn_sched_barrier(0);
				}
			}
			finish_slab(TM);
		};
		auto apply = [&](auto KC) __attribute__((always_inline)) {
			slab(KC, std::integral_constant<int, 0>{});
			slab(KC, std::integral_constant<int, 1>{});
			slab(KC, std::integral_constant<int, 2>{});
			slab(KC, std::integral_constant<int, 3>{});
		};
		switch (p.g_kind) {
		case STPY_K_SE: apply(std::integral_constant<int, STPY_K_SE>{}); break;
		case STPY_K_MATERN32: apply(std::integral_constant<int, STPY_K_MATERN32>{}); break;
		case STPY_K_MATERN52: apply(std::integral_constant<int, STPY_K_MATERN52>{}); break;
		default: apply(std::integral_constant<int, STPY_K_LINEAR>{}); break;
		}
		return;          // (no second copy in this mode)
	}

	// ---- evidence-gradient weight (mode 4): H = (w K^-1 - alpha alpha^T) o F, in place over K^-1 (or K^-1 read from C2)
	if constexpr (EPI == 4) {
		T* const ns = smem;
		const T* const stile = p.C2 ? p.C2 + (int64_t)row0 * p.ldc2 + col0 : ctile;
		const unsigned lds32 = p.C2 ? (unsigned)p.ldc2 : ldc32;
		if (tid < BN) { const int col = col0 + tid; ns[tid] = p.g_na[GUARD ? min(col, p.n - 1) : col]; ns[2 * BN + tid] = p.g_alpha[GUARD ? min(col, p.n - 1) : col]; }
		else { const int row = row0 + tid - BN; ns[tid] = p.g_nb[GUARD ? min(row, p.m - 1) : row]; ns[2 * BN + tid] = p.g_alpha[GUARD ? min(row, p.m - 1) : row]; }
		__syncthreads();
		// pass 1: acc <- kappa * F (pure arithmetic, four exp chains at a time)
		auto apply = [&](auto KC) {
			constexpr int KIND = decltype(KC)::value;
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						acc[tm][tn][i] = p.g_kappa * gram_dfactor<T, KIND>(acc[tm][tn][i], ns[wn * 64 + r16 + tn * 16], ns[BN + lr]);
					__builtin_amdgcn_sched_barrier(0);
				}
		};
		switch (p.g_kind) {
		case STPY_K_SE: apply(std::integral_constant<int, STPY_K_SE>{}); break;
		case STPY_K_MATERN12: apply(std::integral_constant<int, STPY_K_MATERN12>{}); break;
		case STPY_K_MATERN32: apply(std::integral_constant<int, STPY_K_MATERN32>{}); break;
		default: apply(std::integral_constant<int, STPY_K_MATERN52>{}); break;
		}
		// pass 2: acc <- (w * Kinv - alpha_j alpha_i) * acc, the Kinv tile read 16 values at a time
#pragma unroll
		for (int tm = 0; tm < 4; ++tm) {
			T old[4][4];
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
				const int lrc = GUARD ? min(lr, p.m - 1 - row0) : lr;
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) {
					const int lc = wn * 64 + r16 + tn * 16;
					old[tn][i] = stile[(unsigned)lrc * lds32 + (unsigned)(GUARD ? min(lc, p.n - 1 - col0) : lc)];
				}
			}
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					acc[tm][tn][i] *= p.g_w * old[tn][i] - ns[3 * BN + lr] * ns[2 * BN + wn * 64 + r16 + tn * 16];
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	}

	// ---- epilogue: reg i of tile (tm,tn) is C[row0 + wm*64 + tm*16 + crow(lane,i)][col0 + wn*64 + tn*16 + r16]
	const unsigned ldc2_32 = (unsigned)p.ldc2;
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
			T* const crow = ctile + ((unsigned)lr * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) {
				if (GUARD && (row0 + lr >= p.m || col0 + wn * 64 + r16 + tn * 16 >= p.n)) continue;
				if (SUB && DMA) acc[tm][tn][i] = -acc[tm][tn][i];
				// streaming outputs (embedding, Gram matrix: far larger than the caches) bypass the L2 allocation
				// so that the operand tiles the next workgroups re-read stay resident
				if constexpr (EPI == 2) __builtin_nontemporal_store(acc[tm][tn][i], &crow[tn * 16]);
				else crow[tn * 16] = acc[tm][tn][i];
			}
		}
	if (EPI != 4 && p.C2) {         // second copy (panel workspace of potrf): one uniform branch around all its stores
		T* const c2tile = p.C2 + (int64_t)row0 * p.ldc2 + col0;
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int lr = wm * 64 + tm * 16 + MM::crow(lane, i);
				T* const c2row = c2tile + ((unsigned)lr * ldc2_32 + (unsigned)(wn * 64 + r16));
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) {
					if (GUARD && (row0 + lr >= p.m || col0 + wn * 64 + r16 + tn * 16 >= p.n)) continue;
					c2row[tn * 16] = acc[tm][tn][i];
				}
			}
	}
}

// ------------------------------------------------------------------------------------------
// Variant with the A operand DIRECT TO VGPR (aligned shapes, plain products, both types): only B goes through
// LDS (LDS-DMA, two swizzled stages as above); lane (r16, g) of wave (wm, wn) owns, for each of its four
// 16-row tiles, the 32 bytes (chunks 2g, 2g+1) of a 128-byte K-tile row -- exactly the two fragments its MFMAs
// consume -- and refills each fragment IN PLACE for the next K tile right after the last MFMA that reads
// it has been issued.  Half the LDS traffic and half the DMA pieces per flop; the structure the vendor
// library's kernel for these shapes uses (its name says MT128x128x16 ... DTVA1).
// The A loads are inline asm and every wait is placed by hand: hipcc's vmcnt bookkeeping cannot see the
// LDS-DMA pieces, so its own waits either drain them or sit in the wrong place.  In steady state the
// issue order per tile is 4 DMA pieces, 4 first-half refills, 4 second-half refills; when block (h, tm)
// starts, exactly 11 younger operations than the load of fa[tm][h] may still be in flight, and at the end
// of a tile the DMA pieces are older than the 8 refills.  A wait is tied to the fragment it guards through
// a "+v" operand, which keeps the MFMAs that read it behind the wait.
// Measured (tools/dtv_probe.hip, the stand-alone prototype): K-loop slope 73.3 TFLOP/s against 69.1.
// ------------------------------------------------------------------------------------------
template <typename T, int ACC>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_nt_dtv_kernel(GemmArgs<T> p)
{
	constexpr bool SUB = ACC == 1, LOADC = ACC != 0;          // ACC as in gemm_nt_kernel: 0 overwrite, 1 subtract, 2 add
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	constexpr int CH = 16 / sizeof(T);                 // elements per 16-byte chunk (2 doubles / 4 floats): one MFMA k-step each
	typedef T d2 __attribute__((ext_vector_type(CH)));     // one chunk = one fragment
	constexpr int BK = KTile<T>::BK;
	// NB the look-ahead stream's diagonal-block kernel runs beside ONE workgroup of this kernel: two of its waves per
	// SIMD must fit into the registers one workgroup frees (512 - 232 = 280), hence its 128-VGPR cap in potrf.hip --
	// with 138 it waited 2.5 ms per launch for a CU to drain completely and the factorisation got slower
	__shared__ __attribute__((aligned(16))) T smem[2 * BN * BK];      // 32 KiB: B only

	const int b = blockIdx.x;
	int S = (b & 7) + 8 * (b >> 9);
	const int w = (b >> 3) & 63;
	int split = 0;
	if (p.ksplit > 1) {
		split = __builtin_amdgcn_readfirstlane(S / p.nsuper);
		S -= split * p.nsuper;
		if (split >= p.ksplit) return;
	} else if (S >= p.nsuper) return;
	int si, sj;
	if (p.tri == 2) {
		// lower triangle, the DIAGONAL super-tiles (36 of 64 tiles) last, so that they form dispatch rounds of equal weight (experiment,
		// see g_gemm_tri_diag_last: unlike the empty super-tiles of the block-cyclic staircase they cost nothing measurable in place)
		const int noff = p.nst_m * (p.nst_m - 1) / 2;
		if (S >= noff) { si = sj = S - noff; }
		else {
			si = (int)((sqrt(8.0 * (double)S + 1.0) + 1.0) * 0.5);          // strict lower triangle: S = si (si - 1) / 2 + sj, sj < si
			while (si * (si + 1) / 2 <= S) ++si;
			while (si * (si - 1) / 2 > S) --si;
			sj = S - si * (si - 1) / 2;
		}
	} else if (p.tri) {
		si = (int)((sqrt(8.0 * (double)S + 1.0) - 1.0) * 0.5);
		while ((si + 1) * (si + 2) / 2 <= S) ++si;
		while (si * (si + 1) / 2 > S) --si;
		sj = S - si * (si + 1) / 2;
	} else if (p.bc_compact) {
		// block-cyclic staircase, compact: only the super-tiles that hold work are enumerated (host-built prefix table in the
		// kernel arguments).  Enumerating the rectangle and letting the empty ones exit costs 13-32 % (tools/bc_vs_tri.py): workgroups
		// are dispatched in order and each goes to XCD b % 8, so an XCD that drew empty super-tiles idles behind the others' full ones.
		int lo = 0, hi = p.nst_m;                 // largest si with bc_pref[si] <= S
		while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.bc_pref[mid] <= S) lo = mid; else hi = mid; }
		si = lo;
		sj = S - p.bc_pref[si];
	} else {
		si = S / p.nst_n;
		sj = S - si * p.nst_n;
		// (block-cyclic staircase on more super-tile rows than the prefix table holds: rectangle, rows rotated by their index so that
		// every XCD gets a mix of long and short columns)
		if (p.bc_nbt > 0) { sj += si % p.nst_n; if (sj >= p.nst_n) sj -= p.nst_n; }
	}
	const int ti = __builtin_amdgcn_readfirstlane(si * p.st_m + w / p.st_n);
	const int tj = __builtin_amdgcn_readfirstlane(sj * p.st_n + w % p.st_n);
	if (ti >= p.tiles_m || tj >= p.tiles_n) return;
	if (p.tri && tj > ti) return;
	if (p.bc_nbt > 0) {
		const int I = (ti / p.bc_nbt + p.bc_i0) * p.bc_pr + p.bc_myr;
		const int J = (tj / p.bc_nbt + p.bc_j0) * p.bc_pc + p.bc_myc;
		if (I < J) return;
		if (I == J && (tj % p.bc_nbt) > (ti % p.bc_nbt)) return;          // diagonal distribution block: its lower tiles only (4.7 % of the flops at 64 blocks)
	}
	const int row0 = ti * BM, col0 = tj * BN;
	const int kbeg = p.kskip ? row0 : split * p.kchunk;
	const int kend = p.ksplit > 1 ? min(p.k, kbeg + p.kchunk) : p.k;
	const int KT = (kend - kbeg) / BK;
	// desynchronise the two workgroups of a CU once: started together they share the MFMA pipe, finish together and then
	// both sit in their C-tile store / next C-tile load at the same time with the pipe idle; an offset of at least that
	// memory phase persists from round to round (each slot is refilled when its workgroup ends)
	if (p.stagger > 0 && b < 512) {
		const unsigned hwid = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4);   // HW_REG_HW_ID[3:0] = wave slot in its SIMD
		if (hwid & 1u) {
			const unsigned long long t0 = __builtin_amdgcn_s_memtime();
			while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
		}
	}
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;

	// ---- B through LDS-DMA: wave w moves rows [32w, 32w+32), 8 rows (1 KiB) per piece
	// Every global address below is a wave-uniform 64-bit base in SGPRs (advanced on the scalar unit) plus a 32-bit
	// lane offset that never changes: no vector address arithmetic in the K loop (VALU issue is not hidden under the
	// MFMAs of the same SIMD).  Piece i of a wave holds rows r = 32 wave + 8 i + (lane >> 3); the swizzle
	// f(r) has ((r >> 1) & 3) from the lane and (r >> 3) & 1 = i & 1, so two lane offsets serve the four pieces.
	const T* const bbase = p.B + (int64_t)(col0 + wave * 32) * p.ldb + kbeg;           // uniform
	unsigned blane[2];
#pragma unroll
	for (int par = 0; par < 2; ++par) {
		const int rl = lane >> 3;
		const int f = (((rl >> 1) & 3) << 1) | par;
		blane[par] = ((unsigned)rl * (unsigned)p.ldb + (unsigned)(((lane & 7) ^ f) * CH)) * (unsigned)sizeof(T);
	}
	auto dma_one = [&](const T* gbase, unsigned voff, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(laddr) : "memory");
	};
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)smem;
	auto dma_tile = [&](int buf, int k0) {
		const unsigned base = lds0 + (unsigned)(buf * BN + wave * 32) * 128u;       // 128-byte rows
#pragma unroll
		for (int i = 0; i < 4; ++i) dma_one(bbase + (int64_t)(i * 8) * p.ldb + k0, blane[i & 1], base + i * 8 * 128u);
	};

	// ---- A straight into registers: uniform row-tile base + one 32-bit lane offset
	const T* const abase = p.A + (int64_t)(row0 + wm * 64) * p.lda + kbeg;
	const unsigned alane = ((unsigned)r16 * (unsigned)p.lda + (unsigned)g * 2 * CH) * (unsigned)sizeof(T);  // bytes: chunk 2g of the row
	d2 fa[4][2];
	auto lda_frag = [&](int tm, int h, int k0) {
		const T* const ub = abase + (int64_t)tm * 16 * p.lda + k0 + h * CH;        // uniform
		asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fa[tm][h]) : "v"(alane), "s"(ub) : "memory");
	};
	auto wait_frag = [&](int tm, int h) { asm volatile("s_waitcnt vmcnt(11)" : "+v"(fa[tm][h]) :: "memory"); };

	// ---- prologue: tile 0's B pieces and A fragments first, the C tile behind them (its latency overlaps theirs), then ONE
	// ---- wait that hipcc can see (a builtin, not asm): otherwise it parks its vmcnt waits for the C loads at their first
	// ---- use -- inside the loop, where they would drain the hand-counted queue every iteration
	dma_tile(0, 0);
#pragma unroll
	for (int tm = 0; tm < 4; ++tm) lda_frag(tm, 0, 0);
#pragma unroll
	for (int tm = 0; tm < 4; ++tm) lda_frag(tm, 1, 0);
	// accumulators: the C tile itself when subtracting (loaded straight into the accumulator registers; the B
	// fragments are negated after their LDS read, so the product comes out as C - A B^T and the epilogue is store-only)
	v4 acc[4][4];
	T* const ctile = p.C + (int64_t)split * p.split_stride + (int64_t)row0 * p.ldc + col0;
	const unsigned ldc32 = (unsigned)p.ldc;
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const T* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + MM::crow(lane, i)) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = LOADC ? crow[tn * 16] : T(0);
		}
	__builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
	if constexpr (SUB && !MM::HAS_NEG) {         // f32 has no negating MFMA: accumulate A B^T - C and flip the sign once at the store
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = -acc[tm][tn];
	}
	__syncthreads();

	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);
	const int boff = (wn * 64 + r16) * BK;
	int buf = 0;
	// one K tile; LAST: nothing is prefetched any more, so the waits count down what is still in flight
	auto tile = [&](auto last_tag, int knext) {
		constexpr bool LAST = decltype(last_tag)::value;
		if (!LAST) dma_tile(buf ^ 1, knext);
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			if (h == 1) __builtin_amdgcn_sched_barrier(0);
			d2 fb[4];
			const T* bs = smem + buf * BN * BK + boff + ((2 * g + h) ^ fsw) * CH;
#pragma unroll
			for (int t = 0; t < 4; ++t) fb[t] = *(const d2*)(bs + t * 16 * BK);
#pragma unroll
			for (int tm = 0; tm < 4; ++tm) {
				if (!LAST) wait_frag(tm, h);
				else {      // younger loads than fa[tm][h]: the rest of its half, plus the whole second half while in the first
					if (h == 0) { if (tm == 0) asm volatile("s_waitcnt vmcnt(7)" : "+v"(fa[tm][h]) :: "memory"); else if (tm == 1) asm volatile("s_waitcnt vmcnt(6)" : "+v"(fa[tm][h]) :: "memory");
					              else if (tm == 2) asm volatile("s_waitcnt vmcnt(5)" : "+v"(fa[tm][h]) :: "memory"); else asm volatile("s_waitcnt vmcnt(4)" : "+v"(fa[tm][h]) :: "memory"); }
					else { if (tm == 0) asm volatile("s_waitcnt vmcnt(3)" : "+v"(fa[tm][h]) :: "memory"); else if (tm == 1) asm volatile("s_waitcnt vmcnt(2)" : "+v"(fa[tm][h]) :: "memory");
					       else if (tm == 2) asm volatile("s_waitcnt vmcnt(1)" : "+v"(fa[tm][h]) :: "memory"); else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fa[tm][h]) :: "memory"); }
				}
#pragma unroll
				for (int s = 0; s < CH; ++s)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						acc[tm][tn] = (SUB && MM::HAS_NEG) ? MM::mms(fa[tm][h][s], fb[tn][s], acc[tm][tn]) : MM::mma(fa[tm][h][s], fb[tn][s], acc[tm][tn]);
				if (!LAST) lda_frag(tm, h, knext);            // refill in place: every MFMA that reads fa[tm][h] has been issued
			}
		}
		if (!LAST) {
			asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the four DMA pieces are older than the eight A refills
			__syncthreads();
			buf ^= 1;
		}
	};
	for (int kt = 0; kt + 1 < KT; ++kt) tile(std::false_type{}, (kt + 1) * BK);
	tile(std::true_type{}, 0);
	// C may alias A (the block solve multiplies a row block by an inverse diagonal block in place): every wave of the
	// workgroup has its last A fragments in registers (vmcnt(0) above) before anyone stores
	__syncthreads();
	if constexpr (SUB && !MM::HAS_NEG) {
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = -acc[tm][tn];
	}

#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			T* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + MM::crow(lane, i)) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) crow[tn * 16] = acc[tm][tn][i];
		}
	if (p.C2) {
		T* const c2tile = p.C2 + (int64_t)row0 * p.ldc2 + col0;
		const unsigned ldc2_32 = (unsigned)p.ldc2;
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				T* const c2row = c2tile + ((unsigned)(wm * 64 + tm * 16 + MM::crow(lane, i)) * ldc2_32 + (unsigned)(wn * 64 + r16));
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) c2row[tn * 16] = acc[tm][tn][i];
			}
	}
}

// ------------------------------------------------------------------------------------------
// fp32 products on the bf16 MATRIX cores (aligned plain / lower-triangular products, fp32 in, fp32 out).
// v_mfma_f32_16x16x4_f32 runs on the SIMD's own fp32 lanes at the packed-fp32 vector rate (157 TFLOP/s, shared with every VALU
// instruction next to it); the bf16 matrix pipe is a separate unit at 16x that rate.  Every fp32 operand value is split EXACTLY into
// three bf16 parts (x = x1 + x2 + x3: 8 + 8 + 8 significant bits, by truncation -- no rounding anywhere), and the six products of
// weight >= 2^-16 are accumulated in fp32 by v_mfma_f32_16x16x32_bf16, smallest first:
//     a.b = a1 b3 + a1 b2 + a2 b2 + a1 b1 + a2 b1 + a3 b1      (dropped: a2 b3 + a3 b2 + a3 b3 <= 2^-23 |a||b|, below one fp32 rounding
//                                                               of the product -- the recipe of rff_stream_bf16x3_kernel, rff.hip)
// 96 bf16 MFMAs (1536 cycles) per 64 x 64 x 32 wave step replace 256 fp32 MFMAs (8192 cycles): a 390 TFLOP/s ceiling.
// The split is done ON THE FLY, once per workgroup and K step: thread t takes 16 consecutive k of row t/2 of the A tile and of the
// B tile from global memory into registers (the loads of step k+1 are in flight under the MFMAs of step k), splits them (4 VALU
// operations per value + 3 byte-permutes per pair) and writes the three bf16 planes of both tiles to LDS (48 KiB per workgroup:
// two or three workgroups per CU, whose split / MFMA phases interleave on the two pipes).  Plane rows are 64 bytes (32 bf16); the
// 16-byte chunk c of row r lives at c ^ ((r >> 1) & 3), which makes the ds_read_b128 fragment reads conflict-free.
// No operand is pre-split in memory, so every aligned fp32 product of the path takes this kernel unchanged: the trailing updates of
// potrf, the block solve, the feature-space SYRK of KernelizedFeatures.
// ------------------------------------------------------------------------------------------
constexpr int B3_BK = 32;                                  // fp32 values per K step = one bf16 MFMA k-extent
constexpr int B3_PLANE = (BM + BN) * B3_BK * 2;           // bytes of one bf16 plane of both tiles: 16 KiB
// Two structurally different forms were built and measured against this one in one process (tools/f32_gemm_bench.py, trailing update
// n = 32 768, K = 1024: this kernel 191-195 TFLOP/s, the fp32-MFMA kernel 140): (a) ONE workgroup per CU, planes double-buffered,
// the split of step k+1 interleaved instruction by instruction with the MFMAs of step k in the same wave -- 120 TFLOP/s: an in-order
// wave that stops at an LDS store or a wait stops its MFMAs too; (b) a 512-thread workgroup with four MFMA-only waves and four
// load / split / store waves meeting at one barrier per step -- 176-184 TFLOP/s: no better than two independent workgroups whose
// phases drift apart by themselves.  Neither is kept.
// (Three workgroups per CU would need <= 168 VGPRs; the kernel wants 186 -- 64 accumulators, 96 fragments, 32 raw values in
// flight -- and spills 18-34 of them when capped, with or without the fragment reads issued group by group.  Two per CU.)
template <int ACC>
__global__ __launch_bounds__(NTHREADS, 2)
void gemm_nt_bf3_kernel(GemmArgs<float> p)
{
	constexpr bool SUB = ACC == 1, LOADC = ACC != 0;
	typedef float v4f __attribute__((ext_vector_type(4)));
	typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
	typedef unsigned u4v __attribute__((ext_vector_type(4)));
	__shared__ __attribute__((aligned(16))) unsigned char smem[3 * B3_PLANE];          // [part][A rows 0..127 | B rows 128..255][64 B]

	// ---- block -> tile: the XCD-local super-tile map of the kernels above (plain rectangle or lower triangle)
	const int b = blockIdx.x;
	const int S = (b & 7) + 8 * (b >> 9);
	const int w = (b >> 3) & 63;
	if (S >= p.nsuper) return;
	int si, sj;
	if (p.tri) {
		si = (int)((sqrt(8.0 * (double)S + 1.0) - 1.0) * 0.5);
		while ((si + 1) * (si + 2) / 2 <= S) ++si;
		while (si * (si + 1) / 2 > S) --si;
		sj = S - si * (si + 1) / 2;
	} else {
		si = S / p.nst_n;
		sj = S - si * p.nst_n;
	}
	const int ti = __builtin_amdgcn_readfirstlane(si * p.st_m + w / p.st_n);
	const int tj = __builtin_amdgcn_readfirstlane(sj * p.st_n + w % p.st_n);
	if (ti >= p.tiles_m || tj >= p.tiles_n) return;
	if (p.tri && tj > ti) return;
	const int row0 = ti * BM, col0 = tj * BN;
	const int kbeg = p.kskip ? row0 : 0;
	const int KT = (p.k - kbeg) / B3_BK;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, kq = lane >> 4;

	// ---- global -> registers: thread t owns k = 16 (t & 1) .. + 15 of tile row t >> 1, of A and of B
	const int srow = tid >> 1, shalf = tid & 1;
	const float* const ga = p.A + (int64_t)(row0 + srow) * p.lda + kbeg + 16 * shalf;
	const float* const gb = p.B + (int64_t)(col0 + srow) * p.ldb + kbeg + 16 * shalf;
	v4f ra[4], rb[4];
	auto gload = [&](int k0) {
#pragma unroll
		for (int q = 0; q < 4; ++q) { ra[q] = *(const v4f*)(ga + k0 + 4 * q); rb[q] = *(const v4f*)(gb + k0 + 4 * q); }
	};
	// ---- split + store: 16 values -> 3 parts x 16 bf16 = 3 x two 16-byte chunks (logical chunks 2 shalf, 2 shalf + 1 of the row)
	const int swz = (srow >> 1) & 3;
	auto split_store = [&](const v4f (&r)[4], int rowbase) {
		unsigned part[3][8];
#pragma unroll
		for (int q = 0; q < 8; ++q) {            // value pair (2q, 2q + 1)
			unsigned h[2][3];
#pragma unroll
			for (int e = 0; e < 2; ++e) {
				const float x = r[(2 * q + e) >> 2][(2 * q + e) & 3];
				const unsigned u1 = __float_as_uint(x) & 0xffff0000u;
				const float r1 = x - __uint_as_float(u1);                  // exact: the low 16 significant bits
				const unsigned u2 = __float_as_uint(r1) & 0xffff0000u;
				const float r2 = r1 - __uint_as_float(u2);                 // exact: at most 8 significant bits left
				h[e][0] = u1; h[e][1] = u2; h[e][2] = __float_as_uint(r2);
			}
#pragma unroll
			for (int pt = 0; pt < 3; ++pt) part[pt][q] = __builtin_amdgcn_perm(h[1][pt], h[0][pt], 0x07060302u);          // high halves: [odd value | even value]
		}
#pragma unroll
		for (int pt = 0; pt < 3; ++pt) {
			unsigned char* const rowp = smem + pt * B3_PLANE + (rowbase + srow) * 64;
			*(u4v*)(rowp + (((2 * shalf) ^ swz) << 4)) = u4v{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
			*(u4v*)(rowp + (((2 * shalf + 1) ^ swz) << 4)) = u4v{part[pt][4], part[pt][5], part[pt][6], part[pt][7]};
		}
	};

	gload(0);
	// ---- accumulators: zero, or the C tile (negated when subtracting: the products are accumulated on -C and the sign flipped at the store)
	v4f acc[4][4];
	float* const ctile = p.C + (int64_t)row0 * p.ldc + col0;
	const unsigned ldc32 = (unsigned)p.ldc;
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const float* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + 4 * kq + i) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) { const float v = LOADC ? crow[tn * 16] : 0.f; acc[tm][tn][i] = SUB ? -v : v; }
		}
	split_store(ra, 0);
	split_store(rb, BM);
	__syncthreads();

	// fragment reads: lane (r16, kq) takes chunk kq of row r16 of a 16-row tile: physical chunk kq ^ ((r16 >> 1) & 3)
	const unsigned frag = (unsigned)r16 * 64u + (unsigned)((kq ^ ((r16 >> 1) & 3)) << 4);
	const unsigned a_off = (unsigned)(wm * 64) * 64u + frag, b_off = (unsigned)(BM + wn * 64) * 64u + frag;
	constexpr int AP[6] = {0, 0, 1, 0, 1, 2}, BP[6] = {2, 1, 1, 0, 0, 0};
	// Two-level accumulation.  The bf16 MFMA does not round its accumulate to nearest: measured against fp64, a sum carried in the
	// MFMA accumulator loses about half an ulp OF THE ACCUMULATOR per instruction, always toward zero -- 768 instructions of a
	// K = 4096 product on an accumulator that starts at C cost 4.5x the error of the fp32-MFMA kernel, and over a whole fp32
	// factorisation (tens of thousands of instructions onto each element) the bias reached 1e-4 relative and made a well
	// conditioned feature-space matrix fail its Cholesky.  So the MFMAs accumulate B3_CHUNK K steps at a time into a FRESH
	// accumulator (`part`, started from the zero constant: its magnitude, hence its ulp, is that of a 128-deep partial sum, not of C),
	// and the partial sums are added to the real accumulator by the vector ALU, which rounds to nearest (64 additions per 384 MFMAs).
	constexpr int B3_CHUNK = 4;
	v4f part[4][4];
	auto kstep = [&](auto first_tag, int kt) {
		constexpr bool FIRST = decltype(first_tag)::value;
		if (kt + 1 < KT) gload((kt + 1) * B3_BK);
		bf8 fa[3][4], fb[3][4];
#pragma unroll
		for (int pt = 0; pt < 3; ++pt)
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				fa[pt][t] = __builtin_bit_cast(bf8, *(const u4v*)(smem + pt * B3_PLANE + a_off + t * 16 * 64));
				fb[pt][t] = __builtin_bit_cast(bf8, *(const u4v*)(smem + pt * B3_PLANE + b_off + t * 16 * 64));
			}
#pragma unroll
		for (int g = 0; g < 6; ++g)
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					part[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[AP[g]][tm], fb[BP[g]][tn], (FIRST && g == 0) ? v4f{0.f, 0.f, 0.f, 0.f} : part[tm][tn], 0, 0, 0);
		__syncthreads();                       // every wave has read this step's planes
		if (kt + 1 < KT) {
			split_store(ra, 0);
			split_store(rb, BM);
			__syncthreads();
		}
	};
	for (int kc = 0; kc < KT; kc += B3_CHUNK) {
		kstep(std::true_type{}, kc);
		for (int j = 1; j < B3_CHUNK && kc + j < KT; ++j) kstep(std::false_type{}, kc + j);
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn] += part[tm][tn];
	}
	// (C may alias A -- the block solve multiplies in place: all of A's contribution is in the accumulators, and the barrier above
	// has every wave past its last operand load)
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			float* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + 4 * kq + i) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) crow[tn * 16] = SUB ? -acc[tm][tn][i] : acc[tm][tn][i];
		}
}

// ------------------------------------------------------------------------------------------
// K = 128 products on small grids (the panel chain of the factorisation and of the block solve: a row block
// times an inverse diagonal block, C may alias A).  Through the kernels above such a launch is a chain of eight
// K tiles on a few dozen workgroups, each tile waiting a full memory latency with nothing to hide it: 45-90 us
// for ~3 us of MFMA work.  Here a workgroup owns 64 rows x 128 columns: it loads ALL of its A rows and half of its B rows
// (64 x 128 each, 128 KiB of LDS) in one volley of LDS-DMA rows, waits once, multiplies from LDS, and repeats with the
// second half of B.  One workgroup per row block covers the full width, so an in-place product (n = 128) is safe.  Rows are 1 KiB; the XOR swizzle of the
// 128-byte layout is applied to the low three bits of the chunk index, which keeps ds_read_b128 conflict-free.
// ------------------------------------------------------------------------------------------
template <bool SUB>
__global__ __launch_bounds__(NTHREADS, 1)
void gemm_nt_k128_kernel(GemmArgs<double> p)
{
	typedef double T;
	typedef Mfma<double> MM;
	typedef MM::v4 v4;
	typedef double d2 __attribute__((ext_vector_type(2)));
	constexpr int TM = 64, TH = 64, K = 128;           // a workgroup owns 64 rows x 128 columns, as two 64-column halves
	__shared__ __attribute__((aligned(16))) double smem[(TM + TH) * K];       // 128 KiB: the A rows + one half of the B rows
	const int ti = blockIdx.x, tj = blockIdx.y;
	const int row0 = ti * TM, col0 = tj * 2 * TH;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;

	// ---- one LDS-DMA instruction per 1 KiB row: wave w moves rows [16w, 16w+16) of A / of the current half of B
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)smem;
	auto dma_row = [&](const T* g, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(g), "s"(laddr) : "memory");
	};
	auto load_rows = [&](const T* base, int64_t ld, int first_row, int row_limit, int lds_row0) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int r = wave * 16 + i;
			const int f = (((r >> 1) & 3) << 1) | ((r >> 3) & 1);
			const int c = (lane & ~7) | ((lane & 7) ^ f);            // source chunk that lands at physical chunk `lane`
			dma_row(base + (int64_t)min(first_row + r, row_limit) * ld + c * 2, lds0 + (unsigned)(lds_row0 + r) * 1024u);
		}
	};
	load_rows(p.A, p.lda, row0, p.m - 1, 0);
	load_rows(p.B, p.ldb, col0, p.n - 1, TM);

	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);
	const T* as = smem + (wm * 32 + r16) * K;
	const T* bs = smem + (TM + wn * 32 + r16) * K;
	v4 acc[2][2][2];                                   // [half][tm][tn]
#pragma unroll
	for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
		for (int tm = 0; tm < 2; ++tm)
#pragma unroll
			for (int tn = 0; tn < 2; ++tn) acc[hf][tm][tn] = v4{0, 0, 0, 0};
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
#pragma unroll
		for (int kt = 0; kt < K / 16; ++kt)
#pragma unroll
			for (int h = 0; h < 2; ++h) {
				const int off = (kt * 8 + ((2 * g + h) ^ fsw)) * 2;
				d2 fa[2], fb[2];
#pragma unroll
				for (int t = 0; t < 2; ++t) { fa[t] = *(const d2*)(as + t * 16 * K + off); fb[t] = *(const d2*)(bs + t * 16 * K + off); }
#pragma unroll
				for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
					for (int tm = 0; tm < 2; ++tm)
#pragma unroll
						for (int tn = 0; tn < 2; ++tn) acc[hf][tm][tn] = MM::mma(fa[tm][s2], fb[tn][s2], acc[hf][tm][tn]);
			}
		if (hf == 0) {
			__syncthreads();                               // every wave has finished reading the first half of B
			load_rows(p.B, p.ldb, col0 + TH, p.n - 1, TM);
		}
	}
	// every A row of this workgroup is in LDS (and consumed) before anything is stored: C may alias A

	T* const ctile = p.C + (int64_t)row0 * p.ldc + col0;
	T* const c2tile = p.C2 ? p.C2 + (int64_t)row0 * p.ldc2 + col0 : nullptr;
#pragma unroll
	for (int hf = 0; hf < 2; ++hf)
#pragma unroll
		for (int tm = 0; tm < 2; ++tm)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int lr = wm * 32 + tm * 16 + MM::crow(lane, i);
				if (row0 + lr >= p.m) continue;
#pragma unroll
				for (int tn = 0; tn < 2; ++tn) {
					const int lc = hf * TH + wn * 32 + tn * 16 + r16;
					T* c = ctile + (int64_t)lr * p.ldc + lc;
					const T v = SUB ? *c - acc[hf][tm][tn][i] : acc[hf][tm][tn][i];
					*c = v;
					if (c2tile) c2tile[(int64_t)lr * p.ldc2 + lc] = v;
				}
			}
}

// ------------------------------------------------------------------------------------------
// "Sliver" variant of the same panel-chain products (n = 128 columns, K a multiple of 16, fp64), for launches that are
// enqueued WHILE a trailing update floods the chip (look-ahead panels, STPY_FLAG_BESIDE_UPDATE).  Two update workgroups
// leave 512 - 2*224 = 64 VGPRs per SIMD lane and 160 - 2*32 = 96 KiB of LDS on every CU.  A workgroup that needs more
// waits for an update workgroup to exit -- i.e. for the next "round" boundary of the update, 70-270 us per launch in the
// kernel traces, three launches per 128-column block -- while one that fits into the leftovers is placed at once
// (tools/sliver_probe.hip: 79 us beside the update = 79 us alone for a 64-VGPR / 80 KiB workgroup; 101-128 us instead of
// 63 for the 128-VGPR shape).  So: 256 threads capped at 64 VGPRs, 80 KiB of LDS, one workgroup per CU beside the update.
//   tile 32 rows x 128 columns (the full width: an in-place product C = A W^T is safe), wave tile 16 x 64;
//   both operands through LDS-DMA (costs no VGPRs) in a ring of four 16-deep K tiles, one barrier per K tile;
//   every K tile issues its prefetch unconditionally (clamped to the last tile), so the vmcnt bookkeeping is constant.
// ------------------------------------------------------------------------------------------
constexpr int SL_TM = 32, SL_TN = 128, SL_BK = 16, SL_NS = 4;          // (SL_BK: fp64 values per K tile; a row of a K tile is 128 bytes for both types: 32 fp32 values)
constexpr int SL_LDS_BYTES = SL_NS * (SL_TM + SL_TN) * SL_BK * 8;          // 80 KiB
// (dynamic LDS on purpose: with a static 80 KiB array hipcc sees that eight waves per SIMD are out of reach anyway and
// drops the register cap that amdgpu_waves_per_eu(8, 8) = 64 VGPRs is here to enforce)
// (round 4, late: templated on the type -- fp32 takes the few-tile role too; its fp32 MFMA has no negating form, so a subtracting launch
// accumulates on -C and flips the sign at the store)
template <typename T, bool SUB>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void gemm_nt_sliver_kernel(GemmArgs<T> p)
{
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	constexpr int CHE = 16 / (int)sizeof(T);                     // elements per 16-byte chunk: 2 / 4
	constexpr int BKT = 128 / (int)sizeof(T);                    // elements per K tile: 16 / 32
	typedef T chunk_t __attribute__((ext_vector_type(CHE)));
	constexpr bool F64 = sizeof(T) == 8;
	constexpr int SROWS = SL_TM + SL_TN;                         // rows of one stage: 32 of A, 128 of B, 128 bytes each
	extern __shared__ __attribute__((aligned(16))) unsigned char sliver_smem_raw[];
	T* const smem = reinterpret_cast<T*>(sliver_smem_raw);
#ifdef STPY_STAMPS
	unsigned long long stamp[5];
	stamp[0] = __builtin_amdgcn_s_memrealtime();
#endif
	// The sliver shares every SIMD with two (older) update waves that issue MFMAs back to back; instruction issue is arbitrated
	// by priority, then age, so at equal priority it only gets the leftover slots (kernel trace: 105-117 us beside the update
	// for a launch that takes 15 us alone).  It is the latency-critical party: top priority for its whole life.
	__builtin_amdgcn_s_setprio(3);
	const int row0 = blockIdx.x * SL_TM;
	// round 4: blockIdx.y = 128-column block of C (and 128-row block of B): the same 32 x 128 sliver serves products of FEW 128 x 128
	// tiles (small trailing updates, the end of every factorisation): four times the workgroups of the tile kernels, each with a
	// quarter of a tile's K loop to run -- a 128 x 128 x K tile is 2 * 128^2 * K flop on ONE CU (27 us at K = 256), whatever the
	// chip has idle.  Lower-only launches skip the column blocks right of the sliver's own 128-row block.
	const int cb = blockIdx.y;
	if (p.tri && cb * SL_TN > row0) return;
	p.B += (int64_t)cb * SL_TN * p.ldb;
	p.C += cb * SL_TN;
	if (p.C2) p.C2 += cb * SL_TN;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;
	const int KT = p.k / BKT;

	// ---- LDS-DMA: a stage is 20 pieces of 8 rows (1 KiB); wave w moves piece w (A rows 8w..) and pieces w + 4i (B rows
	// ---- 8w + 32(i-1)..), i = 1..4.  Piece parity = w & 1 for all five, so one swizzle term serves them; addresses are a
	// ---- uniform 64-bit base (SGPRs) plus one 32-bit lane offset per operand.
	const int rl = lane >> 3;
	const int fsrc = (((rl >> 1) & 3) << 1) | (wave & 1);
	const unsigned alane = ((unsigned)rl * (unsigned)p.lda + (unsigned)(((lane & 7) ^ fsrc) * CHE)) * (unsigned)sizeof(T);
	const unsigned blane = ((unsigned)rl * (unsigned)p.ldb + (unsigned)(((lane & 7) ^ fsrc) * CHE)) * (unsigned)sizeof(T);
	const T* const abase = p.A + (int64_t)(row0 + 8 * wave) * p.lda;
	const T* const bbase = p.B + (int64_t)(8 * wave) * p.ldb;
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) T*)smem;
	auto dma_one = [&](const T* gbase, unsigned voff, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(voff), "s"(gbase), "s"(laddr) : "memory");
	};
	auto dma_stage = [&](int kt) {                               // K tile kt (clamped) into ring slot kt % NS
		const int k0 = min(kt, KT - 1) * BKT;
		const unsigned base = lds0 + (unsigned)((kt & (SL_NS - 1)) * SROWS) * 128u;
		dma_one(abase + k0, alane, base + (unsigned)(8 * wave) * 128u);
#pragma unroll
		for (int i = 1; i <= 4; ++i) dma_one(bbase + (int64_t)(32 * (i - 1)) * p.ldb + k0, blane, base + (unsigned)(SL_TM + 8 * wave + 32 * (i - 1)) * 128u);
	};

	// ---- prologue: three K tiles in flight, the C tile behind them, ONE wait the compiler can see (see gemm_nt_dtv_kernel)
	dma_stage(0);
	dma_stage(1);
	dma_stage(2);
	// C addressing: register i of a lane is row (lane >> 4) + 4 i of the wave's 16 rows: a uniform base per i (SGPRs) plus ONE
	// 32-bit lane offset; the four column tiles are immediate offsets
	v4 acc[4];
	T* const ctile = p.C + (int64_t)(row0 + wm * 16) * p.ldc + wn * 64;          // uniform
	// (byte offsets in 32 bits, so that every access is  global_load/store v, v_off, s[base:base+1] offset:imm  and no 64-bit
	// per-lane address has to stay alive across the K loop -- with 64 VGPRs those went to scratch)
	// (fp32: register i is row 4 (lane >> 4) + i)
	constexpr int RSTEP = F64 ? 4 : 1, GMUL = F64 ? 1 : 4, CTB = 16 * (int)sizeof(T);          // row step per register, row step per lane group, bytes per column tile
	const unsigned clane = ((unsigned)(g * GMUL) * (unsigned)p.ldc + (unsigned)r16) * (unsigned)sizeof(T);
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const char* const ci = (const char*)(ctile + (int64_t)(RSTEP * i) * p.ldc);  // uniform
#pragma unroll
		for (int t = 0; t < 4; ++t) { const T v = SUB ? *(const T*)(ci + clane + t * CTB) : T(0); acc[t][i] = (SUB && !F64) ? -v : v; }
	}
	__builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
	__syncthreads();
#ifdef STPY_STAMPS
	stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif

	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);
	const int aoff = (wm * 16 + r16) * BKT, boff = (SL_TM + wn * 64 + r16) * BKT;
	for (int kt = 0; kt < KT; ++kt) {
		if (kt > 0) {
			// tiles kt+1 and kt+2 (ten pieces of this wave) may still be in flight; tile kt has landed
			asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
			__syncthreads();                     // ... for every wave, and everybody has finished reading tile kt-1
		}
		dma_stage(kt + 3);                       // into the slot tile kt-1 occupied
		const T* st = smem + (kt & (SL_NS - 1)) * SROWS * BKT;
#pragma unroll
		for (int h = 0; h < 2; ++h) {
			const int off = ((2 * g + h) ^ fsw) * CHE;
			const chunk_t fa = *(const chunk_t*)(st + aoff + off);
#pragma unroll
			for (int t = 0; t < 4; ++t) {
				const chunk_t fb = *(const chunk_t*)(st + boff + t * 16 * BKT + off);
#pragma unroll
				for (int s2 = 0; s2 < CHE; ++s2) acc[t] = (SUB && F64) ? MM::mms(fa[s2], fb[s2], acc[t]) : MM::mma(fa[s2], fb[s2], acc[t]);
			}
		}
	}
#ifdef STPY_STAMPS
	stamp[2] = __builtin_amdgcn_s_memrealtime();
#endif
	// the clamped prefetches of the last tiles are still landing: drained before the LDS is given back, and every wave has
	// consumed its A rows before anybody stores (C may alias A)
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
#ifdef STPY_STAMPS
	stamp[3] = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		char* const ci = (char*)(ctile + (int64_t)(RSTEP * i) * p.ldc);
#pragma unroll
		for (int t = 0; t < 4; ++t) *(T*)(ci + clane + t * CTB) = (SUB && !F64) ? -acc[t][i] : acc[t][i];
	}
	if (p.C2) {
		T* const c2tile = p.C2 + (int64_t)(row0 + wm * 16) * p.ldc2 + wn * 64;
		const unsigned c2lane = ((unsigned)(g * GMUL) * (unsigned)p.ldc2 + (unsigned)r16) * (unsigned)sizeof(T);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			char* const ci = (char*)(c2tile + (int64_t)(RSTEP * i) * p.ldc2);
#pragma unroll
			for (int t = 0; t < 4; ++t) *(T*)(ci + c2lane + t * CTB) = (SUB && !F64) ? -acc[t][i] : acc[t][i];
		}
	}
#ifdef STPY_STAMPS
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	stamp[4] = __builtin_amdgcn_s_memrealtime();
	if (p.dbg && tid == 0 && blockIdx.x < 512) {
		unsigned long long* d = p.dbg + (size_t)blockIdx.x * 8;
		for (int q = 0; q < 5; ++q) d[q] = stamp[q];
		d[5] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 8 << 6 | 4) | (__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 8);   // HW_ID cu / XCC_ID
	}
#endif
}

// ---- C (=, -=) sum over the split-K partial products (fixed order: the result does not depend on scheduling)
template <typename T>
__global__ __launch_bounds__(256)
void splitk_reduce_kernel(const T* __restrict__ work, int splits, int64_t m, int64_t n, T* __restrict__ C, int64_t ldc, int mode)
{
	const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (e >= m * n) return;
	T sum = T(0);
	for (int s = 0; s < splits; ++s) sum += work[(int64_t)s * m * n + e];
	T* c = C + (e / n) * ldc + (e % n);
	*c = mode == 1 ? *c - sum : (mode == 5 ? *c + sum : sum);
}

// ---- skinny product: m <= SKINNY_MAX rows of A against all n rows of B (a handful of right-hand
// ---- sides through the distributed solve, z = L^-1 y).  HBM-bound on B: one workgroup per row of
// ---- B, its four waves stream interleaved 16-byte chunks of the row, A stays in L2.  The sum order is
// ---- fixed by the thread layout.
constexpr int SKINNY_MAX = 8;
template <typename T, int MR, bool VEC>
__global__ __launch_bounds__(256)
void gemm_skinny_kernel(int m, int64_t k, const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb,
                        T* __restrict__ C, int64_t ldc, int mode)
{
	constexpr int CH = 16 / sizeof(T);
	typedef T vch __attribute__((ext_vector_type(CH)));
	const int j = blockIdx.x, tid = threadIdx.x;
	const T* brow = B + (int64_t)j * ldb;
	T acc[MR];
#pragma unroll
	for (int i = 0; i < MR; ++i) acc[i] = T(0);
	if (VEC) {
		for (int64_t kk = (int64_t)tid * CH; kk < k; kk += 256 * CH) {
			const vch b = *(const vch*)(brow + kk);
#pragma unroll
			for (int i = 0; i < MR; ++i) {
				const vch a = *(const vch*)(A + (int64_t)min(i, m - 1) * lda + kk);
#pragma unroll
				for (int e = 0; e < CH; ++e) acc[i] = fma(a[e], b[e], acc[i]);
			}
		}
	} else {
		for (int64_t kk = tid; kk < k; kk += 256) {
			const T b = brow[kk];
#pragma unroll
			for (int i = 0; i < MR; ++i) acc[i] = fma(A[(int64_t)min(i, m - 1) * lda + kk], b, acc[i]);
		}
	}
	__shared__ T red[4][MR];
#pragma unroll
	for (int i = 0; i < MR; ++i) {
		T v = acc[i];
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
		if ((tid & 63) == 0) red[tid >> 6][i] = v;
	}
	__syncthreads();
	if (tid < m) {
		const T sum = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
		T* c = C + (int64_t)tid * ldc + j;
		*c = mode == 1 ? *c - sum : (mode == 5 ? *c + sum : sum);
	}
}

template <typename T>
static int gemm_skinny(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B, int64_t ldb, T* C, int64_t ldc, int mode, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("gemm_nt: dimension exceeds int32"); return -2; }
	constexpr int CH = 16 / sizeof(T);
	const bool vec = (k % CH == 0) && (lda % CH == 0) && (ldb % CH == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
	const dim3 grid((unsigned)n), block(256);
#define STPY_SK(MR) do { if (vec) hipLaunchKernelGGL((gemm_skinny_kernel<T, MR, true>), grid, block, 0, st, (int)m, k, A, lda, B, ldb, C, ldc, mode); \
                         else hipLaunchKernelGGL((gemm_skinny_kernel<T, MR, false>), grid, block, 0, st, (int)m, k, A, lda, B, ldb, C, ldc, mode); } while (0)
	if (m == 1) STPY_SK(1);
	else if (m <= 2) STPY_SK(2);
	else if (m <= 4) STPY_SK(4);
	else STPY_SK(8);
#undef STPY_SK
	return check_launch("gemm_nt (skinny)");
}

template <typename T>
int gemm_nt(int64_t m, int64_t n, int64_t k, const T* A, int64_t lda, const T* B, int64_t ldb,
            T* C, int64_t ldc, T* C2, int64_t ldc2, int mode, int lower_only, hipStream_t st, const BlockCyclic* bc, const RffEpilogue<T>* rff,
            const GramEpilogue<T>* gr, int ksplit, T* split_work, int gflags)
{
	if (m <= 0 || n <= 0) return 0;
	if (k <= 0) {
		if (mode == 0) { set_error("gemm_nt: k == 0 with overwrite mode is not supported"); return -4; }
		return 0;
	}
	// mode 5: C += A B^T (public mode 2 of stpy_gemm_nt) -- the tile kernels with the C tile loaded into the accumulators
	const bool plain = (mode == 0 || mode == 1 || mode == 5) && !lower_only && !bc && !C2;
	// (the block solve multiplies a row block by an inverse diagonal block IN PLACE, C == A: fine for the
	// tile kernel, whose single column of tiles reads its rows of A before storing, not for a kernel
	// that finishes one column of C at a time)
	if (plain && m <= SKINNY_MAX && (const T*)C != A && (const T*)C != B) return gemm_skinny<T>(m, n, k, A, lda, B, ldb, C, ldc, mode, st);
	int64_t kchunk = 0;
	if (ksplit > 1) {
		if (!plain || !split_work) { set_error("gemm_nt: split-K needs a plain product (modes 0/1) and a workspace"); return -12; }
		kchunk = ((k + ksplit - 1) / ksplit + KTile<T>::BK - 1) / KTile<T>::BK * KTile<T>::BK;
		ksplit = (int)((k + kchunk - 1) / kchunk);          // every pass starts inside [0, k)
	}
	if (m > INT32_MAX || n > INT32_MAX || k > INT32_MAX) { set_error("gemm_nt: dimension exceeds int32"); return -2; }
	// the tile kernels address C inside a 128-row tile as (unsigned)row * (unsigned)ldc: rows of 2^25 elements or more would wrap
	if (ldc >= ((int64_t)1 << 25) || ldc2 >= ((int64_t)1 << 25)) { set_error("gemm_nt: leading dimension of C (%lld) must be below 2^25 elements", (long long)(ldc > ldc2 ? ldc : ldc2)); return -10; }
	GemmArgs<T> p;
	p.A = A; p.B = B; p.C = C; p.C2 = C2;
	p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldc2 = ldc2;
	p.m = (int)m; p.n = (int)n; p.k = (int)k;
	p.tiles_m = (int)((m + BM - 1) / BM);
	p.tiles_n = (int)((n + BN - 1) / BN);
	p.mode = mode;
	p.epi_half = 0; p.epi_scale = T(1); p.epi_bias = nullptr; p.epi_fscale = nullptr; p.epi_by_row = 0;
	if (mode == 2) {
		if (!rff || sizeof(T) != 4) { set_error("gemm_nt: mode 2 (fused RFF epilogue) is fp32 only and needs its parameters"); return -12; }
		p.epi_half = rff->half; p.epi_scale = rff->scale; p.epi_bias = rff->bias; p.epi_fscale = rff->fscale; p.epi_by_row = rff->by_row;
	}
	p.g_na = p.g_nb = nullptr; p.g_kappa = T(1); p.g_offset = p.g_diag = T(0); p.g_kind = 0; p.g_combine = 0;
	p.g_alpha = nullptr; p.g_w = T(1); p.kskip = 0;
	if (mode == 3 || mode == 4) {
		if (!gr) { set_error("gemm_nt: modes 3/4 need the Gram epilogue parameters"); return -12; }
		p.g_na = gr->na; p.g_nb = gr->nb; p.g_kappa = gr->kappa; p.g_offset = gr->offset; p.g_diag = gr->diag_add;
		p.g_kind = gr->kind; p.g_combine = gr->combine; p.g_alpha = gr->alpha; p.g_w = gr->weight;
		if (mode == 4 && !gr->alpha) { set_error("gemm_nt: mode 4 needs alpha"); return -12; }
	}
	if (lower_only == 2) {          // lower tiles + upper-triangular operands (K range starts at the tile's first row)
		if (m != n || m % BM != 0 && false) { set_error("gemm_nt: kskip needs a square problem"); return -12; }
		p.kskip = 1;
	}
	p.ksplit = 1; p.kchunk = 0; p.split_stride = 0;
#ifdef STPY_STAMPS
	p.dbg = g_gemm_dbg;
#endif
	if (ksplit > 1) {       // partial products go to the packed workspace [ksplit][m][n]; summed below
		p.ksplit = ksplit;
		p.kchunk = (int)kchunk;
		p.split_stride = m * n;
		p.C = split_work; p.ldc = n;
	}
	p.exp = g_gemm_exp;
	p.tri = (lower_only && m == n && !bc) ? (g_gemm_tri_diag_last ? 2 : 1) : 0;
	p.bc_nbt = 0; p.bc_pr = p.bc_pc = 1; p.bc_myr = p.bc_myc = p.bc_i0 = p.bc_j0 = 0;
	if (bc) {
		if (bc->nb_dist <= 0 || bc->nb_dist % BM != 0) { set_error("gemm_nt: block-cyclic block must be a positive multiple of %d", BM); return -13; }
		p.bc_nbt = bc->nb_dist / BM; p.bc_pr = bc->pr; p.bc_pc = bc->pc; p.bc_myr = bc->myr; p.bc_myc = bc->myc; p.bc_i0 = bc->i0; p.bc_j0 = bc->j0;
	}
	if (p.tri) {
		p.st_m = ST; p.st_n = ST;
	} else {
		int sn = 1;
		while (sn < ST && sn < p.tiles_n) sn *= 2;
		p.st_n = sn; p.st_m = (ST * ST) / sn;
	}
	p.nst_m = (p.tiles_m + p.st_m - 1) / p.st_m;
	p.nst_n = (p.tiles_n + p.st_n - 1) / p.st_n;
	p.nsuper = p.tri ? p.nst_m * (p.nst_m + 1) / 2 : p.nst_m * p.nst_n;
	p.bc_compact = 0;
	if (p.bc_nbt > 0 && p.nst_m <= BC_MAX_ROWS && ksplit <= 1) {
		// a tile is needed iff its global block (I, J) has I > J, or I == J and it lies on / below that block's own diagonal; along a
		// row the needed tiles are a prefix, so a super-tile row's count follows from its LAST tile row and each column's FIRST tile
		auto needed = [&](int ti, int tj) {
			const int I = (ti / p.bc_nbt + p.bc_i0) * p.bc_pr + p.bc_myr, J = (tj / p.bc_nbt + p.bc_j0) * p.bc_pc + p.bc_myc;
			return I > J || (I == J && (tj % p.bc_nbt) <= (ti % p.bc_nbt));
		};
		int acc = 0;
		for (int si = 0; si < p.nst_m; ++si) {
			p.bc_pref[si] = acc;
			int ti = si * p.st_m + p.st_m - 1;
			if (ti >= p.tiles_m) ti = p.tiles_m - 1;
			int cnt = 0;
			while (cnt < p.nst_n && needed(ti, cnt * p.st_n)) ++cnt;
			acc += cnt;
		}
		p.bc_pref[p.nst_m] = acc;
		p.nsuper = acc;
		p.bc_compact = 1;
		if (acc == 0) return 0;          // nothing below the staircase in this window
	}
	const int64_t nblocks = (int64_t)(((int64_t)p.nsuper * p.ksplit + 7) / 8) * 512;
	if (nblocks > INT32_MAX) { set_error("gemm_nt: grid too large"); return -2; }
	// Workgroups of one launch all take the same time, so the two that share a CU would reach
	// their memory-bound prologue/epilogue together, round after round, and the MFMA pipes would
	// idle for both.  Starting the odd wave slot half a tile late in the FIRST round only keeps the
	// pair out of phase for the whole launch (later workgroups start when a predecessor ends).
	// Only worth its cost (half a tile, once) when the launch runs for several rounds.
	{
		const int64_t real_tiles = p.tri ? (int64_t)p.tiles_m * (p.tiles_m + 1) / 2 : (int64_t)p.tiles_m * p.tiles_n;
		const int64_t kt = (k + KTile<T>::BK - 1) / KTile<T>::BK;
		p.stagger = (g_gemm_stagger && real_tiles >= 8 * 512) ? (g_gemm_stagger > 1 ? g_gemm_stagger : (int)(kt * 64 * 64)) : 0;     // 1: kt * 64 MFMAs * 64 cycles = half of a two-wave tile; > 1: that many cycles
	}
	constexpr int CH = 16 / sizeof(T);
	const bool aligned = (m % BM == 0) && (n % BN == 0) && (k % KTile<T>::BK == 0) && (lda % CH == 0) && (ldb % CH == 0) &&
	                     (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0);
	const dim3 grid((unsigned)nblocks), block(NTHREADS);
#define STPY_LAUNCH(G, S, E) hipLaunchKernelGGL((gemm_nt_kernel<T, G, S, E>), grid, block, 0, st, p)
	{
		// panel-chain products enqueued beside a trailing update (see gemm_nt_sliver_kernel; fp64 only: beside the fp32 update's 144 KiB workgroups
		// nothing else fits on a CU)
		// ... and (round 4, route key 30) plain / lower-only products of few 128 x 128 tiles with K >= 64, both types: see the kernel
		constexpr int CHE = 16 / (int)sizeof(T), BKT = 128 / (int)sizeof(T);
		const int64_t sl_tiles = p.tri ? (int64_t)p.tiles_m * (p.tiles_m + 1) / 2 : (int64_t)p.tiles_m * p.tiles_n;
		const bool sl_beside = sizeof(T) == 8 && (gflags & GEMM_BESIDE) && n == SL_TN && !lower_only;
		// (crossover, tools/sliver_vs_tile.py + tools/potrf_sweep.py "30=...": lower-triangular updates up to ~3200 tiles, i.e. 10 000 rows, at every
		// K = 128 .. 2048 -- 4.8x at 36 tiles, 1.7x at 136, 1.3-1.4x at 528, par at 2080, 0.93x at 4656; rectangles with a long K, the block solve's
		// products, turn earlier: half the threshold)
		// (fp32 rectangles: an eighth -- above ~400 tiles the bf16-split tile kernel, whose pipe is 2.6x the fp32 MFMA's, is ahead again: block solve
		// n = 65 536 fp32, M = 256 / 1024 / 4096 rows: 19.6 / 35.8 / 99.4 ms without slivers, 16.2 / 34.7 / 102.8 with the fp64 rule)
		const int64_t sl_rect = sizeof(T) == 4 ? g_gemm_sliver_tiles / 8 : g_gemm_sliver_tiles / 2;
		const bool sl_few = g_gemm_sliver_tiles > 0 && sl_tiles <= (p.tri ? g_gemm_sliver_tiles : sl_rect) && (n % SL_TN == 0) && k >= 64 && (!lower_only || p.tri == 1) && lower_only != 2 && n / SL_TN <= 65535;
		if ((sl_beside || sl_few) && (m % SL_TM == 0) && (k % BKT == 0) && k >= BKT && (mode == 0 || mode == 1) && !bc &&
		    p.ksplit == 1 && !g_gemm_exp && (lda % CHE == 0) && (ldb % CHE == 0) && lda < (1 << 24) && ldb < (1 << 24) && ldc < ((int64_t)1 << 25) && ldc2 < ((int64_t)1 << 25) &&
		    (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0)) {
			static std::atomic<bool> attr_set[2];
			constexpr int which = sizeof(T) == 8 ? 0 : 1;
			if (!attr_set[which].load(std::memory_order_acquire)) {
				hipError_t e = hipFuncSetAttribute((const void*)gemm_nt_sliver_kernel<T, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES);
				if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_nt_sliver_kernel<T, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SL_LDS_BYTES);
				if (e != hipSuccess) { set_error("gemm_nt (sliver): hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return -1000 - (int)e; }
				attr_set[which].store(true, std::memory_order_release);
			}
			const dim3 gs((unsigned)(m / SL_TM), (unsigned)(n / SL_TN));
			if (mode == 1) hipLaunchKernelGGL((gemm_nt_sliver_kernel<T, true>), gs, block, SL_LDS_BYTES, st, p);
			else hipLaunchKernelGGL((gemm_nt_sliver_kernel<T, false>), gs, block, SL_LDS_BYTES, st, p);
			return check_launch("gemm_nt (sliver)");
		}
	}
	if constexpr (sizeof(T) == 8) {
		// the panel chain's K = 128 products (see gemm_nt_k128_kernel)
		const int64_t t64 = ((m + 63) / 64) * (n / 64);
		if (g_gemm_k128 && !(gflags & GEMM_BESIDE) && k == 128 && (n % 128 == 0) && (mode == 0 || mode == 1) && !lower_only && !bc && p.ksplit == 1 && !g_gemm_exp &&
		    t64 <= g_gemm_k128 && (lda % 2 == 0) && (ldb % 2 == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)B & 15) == 0)) {
			const dim3 g64((unsigned)((m + 63) / 64), (unsigned)(n / 128));
			if (mode == 1) hipLaunchKernelGGL((gemm_nt_k128_kernel<true>), g64, block, 0, st, p);
			else hipLaunchKernelGGL((gemm_nt_k128_kernel<false>), g64, block, 0, st, p);
			return check_launch("gemm_nt");
		}
	}
	if constexpr (sizeof(T) == 4) {
		// fp32 products on the bf16 matrix cores (exact three-way split, see gemm_nt_bf3_kernel): aligned shapes from 64 tiles on
		// (below that the launch is latency-bound either way and the fp32-MFMA kernels keep their tuned small-grid forms)
		const int64_t ntiles = p.tri ? (int64_t)p.tiles_m * (p.tiles_m + 1) / 2 : (int64_t)p.tiles_m * p.tiles_n;
		if (g_gemm_bf3 > 0 && ntiles >= g_gemm_bf3 && aligned && (mode == 0 || mode == 1 || mode == 5) && !bc && !C2 && p.ksplit == 1 && !g_gemm_exp &&
		    k >= 2 * B3_BK && p.tri != 2 && ldc < ((int64_t)1 << 31)) {
			if (mode == 1) hipLaunchKernelGGL((gemm_nt_bf3_kernel<1>), grid, block, 0, st, p);
			else if (mode == 5) hipLaunchKernelGGL((gemm_nt_bf3_kernel<2>), grid, block, 0, st, p);
			else hipLaunchKernelGGL((gemm_nt_bf3_kernel<0>), grid, block, 0, st, p);
			return check_launch("gemm_nt (bf16x3)");
		}
	}
	{
		const int64_t dtv_tiles = p.tri ? (int64_t)p.tiles_m * (p.tiles_m + 1) / 2 : (int64_t)p.tiles_m * p.tiles_n * p.ksplit;
		if (g_gemm_dtv > 0 && dtv_tiles >= g_gemm_dtv && aligned && (mode == 0 || mode == 1 || mode == 5) && !g_gemm_exp && lda < (1 << 24) && ldb < (1 << 24) && k >= g_gemm_dtv_min_k * (int)(8 / sizeof(T))) {
			if (mode == 1 && p.ksplit == 1) hipLaunchKernelGGL((gemm_nt_dtv_kernel<T, 1>), grid, block, 0, st, p);
			else if (mode == 5 && p.ksplit == 1) hipLaunchKernelGGL((gemm_nt_dtv_kernel<T, 2>), grid, block, 0, st, p);
			else hipLaunchKernelGGL((gemm_nt_dtv_kernel<T, 0>), grid, block, 0, st, p);
			if (p.ksplit > 1) {
				const int64_t total = m * n;
				hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
				                   (const T*)split_work, p.ksplit, m, n, C, ldc, mode);
			}
			return check_launch("gemm_nt");
		}
	}
	if (p.ksplit > 1) {
		if (aligned) STPY_LAUNCH(false, 0, 0); else STPY_LAUNCH(true, 0, 0);
		const int64_t total = m * n;
		hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st,
		                   (const T*)split_work, p.ksplit, m, n, C, ldc, mode);
	}
	else if (mode == 1) { if (aligned) STPY_LAUNCH(false, 1, 0); else STPY_LAUNCH(true, 1, 0); }
	else if (mode == 5) { if (aligned) STPY_LAUNCH(false, 2, 0); else STPY_LAUNCH(true, 2, 0); }
	else if (mode == 0) { if (aligned) STPY_LAUNCH(false, 0, 0); else STPY_LAUNCH(true, 0, 0); }
	else if (mode == 3) { if (aligned) STPY_LAUNCH(false, 0, 3); else STPY_LAUNCH(true, 0, 3); }
	else if (mode == 4) { if (aligned) STPY_LAUNCH(false, 0, 4); else STPY_LAUNCH(true, 0, 4); }
	else if (mode == 2) {
		if constexpr (sizeof(T) == 4) { if (aligned) STPY_LAUNCH(false, 0, 2); else STPY_LAUNCH(true, 0, 2); }
	}
	else { set_error("gemm_nt: unknown mode %d", mode); return -11; }
#undef STPY_LAUNCH
	return check_launch("gemm_nt");
}

template int gemm_nt<double>(int64_t, int64_t, int64_t, const double*, int64_t, const double*, int64_t, double*, int64_t, double*, int64_t, int, int, hipStream_t, const BlockCyclic*, const RffEpilogue<double>*, const GramEpilogue<double>*, int, double*, int);
template int gemm_nt<float>(int64_t, int64_t, int64_t, const float*, int64_t, const float*, int64_t, float*, int64_t, float*, int64_t, int, int, hipStream_t, const BlockCyclic*, const RffEpilogue<float>*, const GramEpilogue<float>*, int, float*, int);

int gemm_splitk_plan(int64_t m, int64_t n, int64_t k)
{
	// enough workgroups for two per CU on 256 CUs, at least 64 K-tiles (1024 deep) per pass
	const int64_t tiles = ((m + BM - 1) / BM) * ((n + BN - 1) / BN);
	if (m <= SKINNY_MAX || tiles >= 384) return 1;
	int64_t s = (512 + tiles - 1) / tiles;
	const int64_t smax = k / 1024;
	if (s > smax) s = smax;
	if (s > 16) s = 16;
	return s < 2 ? 1 : (int)s;
}

}  // namespace stpy
