// potrf.hip -- blocked right-looking Cholesky (lower), two levels:
//   outer panels of nb columns (default 512): the trailing update A22 -= P P^T has K = nb, which
//     puts the C read+write traffic at nb/8 flop per byte -- far on the MFMA side of the roofline;
//   inner 128-column blocks, left-looking inside the panel: update block column from the panel
//     so far (MFMA GEMM), factor + invert the 128x128 diagonal block in ONE workgroup (LDS
//     resident, wave shuffles), then the sub-diagonal part by a GEMM with inverse(L_cc)^T.
// Every O(n^3) and O(n^2 nb) term runs on the MFMA GEMM of gemm.hip; only the 128x128 diagonal
// blocks (n/128 of them, ~0.7 Mflop each) run on the vector ALU.
//
// The factored panel is kept twice: in place in A (the result) and in a contiguous n x nb
// workspace P (leading dimension nb) that the left-looking updates and the trailing update read:
// rows of P are nb*8 bytes apart instead of lda*8, and the GEMM's two operands become the same
// buffer.  inverse(L_cc) of every diagonal block is kept in `winv` for the triangular solves.
#include <atomic>
#include <type_traits>
#include <map>
#include <vector>
#include <mutex>

#include "common.h"

namespace stpy {

// default outer panel width: 1024 halves the read+write passes over the trailing matrix compared
// with 512 (measured 2 % faster end to end at N = 65 536); the solves keep 512 (see solve.hip)
// g_potf2_scalar = 0 (lab knob, key 2): 1 = the column-by-column VALU kernel (kept for A/B runs in the lab build)
// stpy_tune key 18: the rows below a panel's diagonal block as one strip launch (solve.hip) instead of two products per 128 columns.
// 0 (default): off.  1: every panel, on the look-ahead stream -- the strip kernel's 120-202 VGPRs wait for update workgroups to retire.
// 2: only the first panel (nothing else on the chip: no measurable difference).  3: the look-ahead stream factors only the panel's
// diagonal block and the strip runs on the update's stream after the trailing update (with key 12 the whole chain then sits on the
// reserved CUs).  All slower than 0 (tools/potrf_sweep.py "18=0|1|3;12=0|16384", ms at N = 16 384 / 32 768 / 65 536: 34.5 / 190.7 / 1376
// against 35.5 / 193 / 1381 (1), 35.8 / 197.5 / 1403 (3), 37.8 / 203 / 1407 (3 + reserved)): in the kernel trace the diagonal-block
// kernel alone is 21 of the 35 ms at N = 16 384 (165 us average beside the update, 58 us alone) whatever surrounds it, and keeping
// the update off eight CUs costs more than that kernel gains.
// g_potrf_strip = 0 (lab knob, key 18)
// g_potrf_serial_below = 3200:         stpy_tune key 21 (see potrf()): no look-ahead below this many remaining rows.  Round 3 had it off (the fully
                                       // serial order cost 42.4 ms at N = 16 384 against 35.2 overlapped).  Round 4: with the small trailing updates on the sliver
                                       // kernel (8-16 us instead of 36-40 per launch) the two event hand-overs of a look-ahead step (16 us each) cost more than the
                                       // overlap gains once fewer than ~3000 rows are left: potrf 0.50 -> 0.44 / 1.07 -> 0.93 / 2.34 -> 2.18 / 6.31 -> 6.12 /
                                       // 28.3 -> 28.2 ms at N = 1024 / 2048 / 4096 / 8192 / 16 384 (flat between 2100 and 4200, worse from 6400 on)
// stpy_tune key 11 (0 = off, the default): blocks factored beside a trailing update take the 64-VGPR / four-wave form below.
// Measured (tools/potrf_sweep.py, gpurun_out/potrf_sweep3.log): it is placed at once, as intended, but then RUNS 8x slower
// beside the real update than alone (730-770 us against 94 us in the kernel trace) and loses to the eight-wave kernel that
// waits for a slot (N = 16 384 potrf 41.0 ms against 36.7 ms).  Kept switchable for the next experiment, not used.
// g_potf2_sliver = 0 (lab knob, key 11)
constexpr int PT_THREADS = 512;
constexpr int SLD = 132;     // LDS row stride in elements: 132 = 4 (mod 32) keeps the (row, k mod 4) lane map conflict-free

#if STPY_LAB
// One workgroup: Cholesky of a (<=128)x(<=128) SPD block + inverse of its factor.
//   A    : block in global memory (lower triangle read; L written back to the lower triangle)
//   W    : 128x128 row-major, receives inverse(L) (zeros above the diagonal, identity padding)
//   P2   : optional second copy of L (rows x 128 cols window of the panel workspace, zeros above the diagonal)
// Thread map: 4 lanes per row (q = lane & 3 splits the k range), 128 rows -> 512 threads.
template <typename T>
__global__ __launch_bounds__(PT_THREADS)
void potf2_trtri_kernel(T* __restrict__ A, int64_t lda, int nbk, T* __restrict__ W,
                        T* __restrict__ P2, int64_t ldp2, int32_t* info, int block_row0)
{
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* S = reinterpret_cast<T*>(smem_raw);          // [128][SLD]
	T* dinv = S + IB * SLD;                         // [128]  1 / l_jj
	T* ldiag = dinv + IB;                           // [128]  l_jj
	const int tid = threadIdx.x;
	const int row = tid >> 2, q = tid & 3;

	// ---- load: lower triangle of the block, identity for the padding rows/cols
	for (int idx = tid; idx < IB * IB; idx += PT_THREADS) {
		const int i = idx >> 7, j = idx & 127;
		T v = T(0);
		if (i < nbk && j <= i) v = A[(int64_t)i * lda + j];
		else if (i >= nbk && i == j) v = T(1);
		S[i * SLD + j] = v;
	}
	__syncthreads();

	// ---- potf2, left-looking by column: t_i = a_ij - sum_{k<j} l_ik l_jk for all rows i >= j.
	// ---- S[j][j] keeps the pivot d_j (never overwritten: everybody reads it between the two
	// ---- barriers); l_jj and 1/l_jj live in ldiag / dinv.
	for (int j = 0; j < IB; ++j) {
		T part = T(0);
		if (row >= j) {
			const T* si = S + row * SLD;
			const T* sj = S + j * SLD;
			// four independent partial sums: the LDS reads of a 16-deep step are all in flight together
			T p1 = T(0), p2 = T(0), p3 = T(0);
			int k = q;
			for (; k + 12 < j; k += 16) {
				part += si[k] * sj[k];
				p1 += si[k + 4] * sj[k + 4];
				p2 += si[k + 8] * sj[k + 8];
				p3 += si[k + 12] * sj[k + 12];
			}
			for (; k < j; k += 4) part += si[k] * sj[k];
			part += (p1 + p2) + p3;
		}
		part += __shfl_xor(part, 1);
		part += __shfl_xor(part, 2);
		if (row >= j && q == 0) S[row * SLD + j] -= part;
		__syncthreads();
		T d = S[j * SLD + j];
		if (!(d > T(0)) || !(d < T(1e300))) {
			if (tid == 0) atomicCAS(info, 0, block_row0 + j + 1);
			d = T(1);
		}
		const T l = sqrt(d);
		const T rl = T(1) / l;
		if (q == 0) {
			if (row > j) S[row * SLD + j] *= rl;
			else if (row == j) { ldiag[j] = l; dinv[j] = rl; }
		}
		__syncthreads();
	}

	// ---- write L back (and the panel copy with explicit zeros above the diagonal)
	for (int idx = tid; idx < IB * IB; idx += PT_THREADS) {
		const int i = idx >> 7, j = idx & 127;
		if (i < nbk && j < nbk) {
			const T v = (j < i) ? S[i * SLD + j] : (j == i ? ldiag[i] : T(0));
			if (j <= i) A[(int64_t)i * lda + j] = v;
			if (P2) P2[(int64_t)i * ldp2 + j] = v;
		}
	}
	__syncthreads();

	// ---- trtri: column c of W = inverse(L) by forward substitution, W[k][c] kept at S[c][k]
	// ---- (k >= c: the strict upper triangle + diagonal of S, which L no longer needs).
	{
		const int c = row;
		if (q == 0) S[c * SLD + c] = dinv[c];
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		// the four lanes of a column and all 16 columns of a wave run the i loop together; a
		// column simply idles until i passes its own index
		const int cmin = (tid >> 6) * 16;          // first column handled by this wave
		for (int i = cmin + 1; i < IB; ++i) {
			T part = T(0);
			if (i > c) {
				const T* li = S + i * SLD;
				const T* wc = S + c * SLD;
				T p1 = T(0), p2 = T(0), p3 = T(0);
				int k = c + q;
				for (; k + 12 < i; k += 16) {
					part += li[k] * wc[k];
					p1 += li[k + 4] * wc[k + 4];
					p2 += li[k + 8] * wc[k + 8];
					p3 += li[k + 12] * wc[k + 12];
				}
				for (; k < i; k += 4) part += li[k] * wc[k];
				part += (p1 + p2) + p3;
			}
			part += __shfl_xor(part, 1);
			part += __shfl_xor(part, 2);
			if (i > c && q == 0) S[c * SLD + i] = -part * dinv[i];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
		}
	}
	__syncthreads();
	for (int idx = tid; idx < IB * IB; idx += PT_THREADS) {
		const int i = idx >> 7, c = idx & 127;
		W[idx] = (c <= i) ? S[c * SLD + i] : T(0);
	}
}

#endif  // STPY_LAB (scalar diagonal-block kernel)

// ------------------------------------------------------------------------------------------
// MFMA-blocked version of the same job (the default): the 128x128 block is processed in 16-wide
// sub-blocks so that only the 16x16 diagonal sub-blocks see a serial pivot chain (one wave, rows in
// registers, v_readlane broadcasts -- no LDS, no barriers inside it); the 16-wide panel below it,
// the trailing update and the whole triangular inverse are 16x16x16 products on the MFMA.
// 3 workgroup barriers per 16 columns instead of 2 per column.
//
// LDS budget matters more than anything else here: the kernel runs on the look-ahead stream while
// the trailing-update GEMM keeps two 72 KiB workgroups on every CU.  A workgroup that needs more
// LDS than ONE freed GEMM slot (+ the 16 KiB the GEMM pair leaves over) is never placed until the
// GEMM grid drains -- measured: the 150 KiB version waited 2.8 ms on average for an 80 us job.
// So the block is held as a PACKED lower triangle (row i at i(i+1)/2: 64.5 KiB fp64) plus the eight
// 16x16 inverse diagonal blocks (17 KiB): 83 KiB.  The off-diagonal blocks of inverse(L) go straight
// to the output array in global memory and are read back from there (L2-hot, written by the same
// wave) when a later block of the same block column needs them.
//   S   packed lower triangle of the block, L as it is produced
//   WD  [8][16][17]  inverse of every 16x16 diagonal sub-block of L
// W = inverse(L): W_ii = WD_i;  W_ij = -WD_i * sum_{k=j}^{i-1} L_ik W_kj  for i > j; wave j owns
// block column j, no workgroup barrier is needed while the eight columns advance.
// ------------------------------------------------------------------------------------------
constexpr int SB = 16, NSB = IB / SB, WLD = 17;
#ifdef STPY_STAMPS
__device__ unsigned long long* stpy_dbg_potf2 = nullptr;      // diagnostic builds only: [count, pad, 8 stamps x 1000]
extern "C" __attribute__((visibility("default"))) void stpy_debug_set_potf2_buffer(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(stpy_dbg_potf2), &p, sizeof(p)); }
#endif
constexpr int TRI = IB * (IB + 1) / 2;

// (24-bit multiply: v_mul_u32_u24 issues at full rate, the 32-bit v_mul_lo_u32 at a quarter -- the index arithmetic of the sub-block
// products was half of their time; i < 2^12 here)
__device__ __forceinline__ int tri(int i, int k) { return (int)(__umul24((unsigned)i, (unsigned)(i + 1)) >> 1) + k; }      // k <= i

// broadcast of one lane's value to the whole wave through SGPRs (v_readlane), lane index uniform
__device__ __forceinline__ double bcast(double v, int src)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bcast(float v, int src)
{
	return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// (VGPR cap: the kernel runs beside ONE trailing-update workgroup; two of its waves per SIMD must fit into the
// registers one such workgroup frees -- 512 - 232 = 280 next to the direct-to-VGPR GEMM, i.e. at most 136 each;
// the second argument of __launch_bounds__ is how that cap is expressed: four waves per SIMD = 128 VGPRs)
template <typename T>
__global__ __launch_bounds__(PT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void potf2_trtri_mfma_kernel(T* __restrict__ A, int64_t lda, int nbk, T* W,
                             T* __restrict__ P2, int64_t ldp2, int32_t* info, int block_row0)
{
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* S = reinterpret_cast<T*>(smem_raw);          // [TRI] packed lower triangle
	T* WD = S + TRI;                                // [8][16][WLD]
	T* SC = WD + NSB * SB * WLD;                    // [8 waves][16][WLD] scratch (fp32 operand re-layout only)
#ifdef STPY_STAMPS
	const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r16 = lane & 15, g = lane >> 4;

	// all 32 loads of a thread are issued before the first LDS store (one memory round trip instead of 32 in a chain:
	// this phase took 9.8 us of the kernel's 87)
	{
		constexpr int NIT = IB * IB / PT_THREADS;
		T v[NIT];
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int idx = tid + it * PT_THREADS, i = idx >> 7, j = idx & 127;
			v[it] = (j <= i && i < nbk) ? A[(int64_t)i * lda + j] : ((i == j) ? T(1) : T(0));
		}
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int idx = tid + it * PT_THREADS, i = idx >> 7, j = idx & 127;
			if (j <= i) S[tri(i, j)] = v[it];
		}
	}
	__syncthreads();
#ifdef STPY_STAMPS
	const unsigned long long stamp_t1 = __builtin_amdgcn_s_memrealtime();
#endif

	// ---- 16x16 diagonal sub-block kb (ONE wave): Cholesky factor and its inverse together.
	// Lane (q, i) = (lane >> 4, lane & 15) holds row i, columns 4q..4q+3, of the block (a[]) and of the inverse being
	// built (w[], starts as the identity).  Pivot j: the scaled column j lives in lane group q = j/4; every lane fetches
	// its row's multiplier and the four column-j entries of ITS columns with lane permutes, then does 4 + 4 FMAs:
	// the right-looking update a_ik -= l_ij l_kj and the row operation W_i -= l_ij W_j that turns I into inverse(L).
	// (The first version kept a whole row per lane on 16 lanes: 15 broadcasts + 15 FMAs per pivot in one dependent chain,
	// then a second pass for the inverse -- 6.6 us per sub-block, three quarters of the kernel.)
	auto diag_block = [&](int kb) {
		const int o = kb * SB;
		if constexpr (sizeof(T) == 8) {
			// fp64 (round 4): block elimination with 4 x 4 pivot blocks on the MFMA.  R = [A | I] (16 x 32) is held as two C/D
			// fragments (reg r of lane (g, i) = row g + 4r, column i).  Step k:  D = R[4k.., 4k..] (4 x 4) is broadcast through SGPRs and
			// EVERY lane factors it and inverts the factor in straight-line code (four reciprocal-square-root chains: the only serial
			// part left); R_k <- inv(L_kk) R_k and R <- R - M R_k are four v_mfma_f64_16x16x4 -- a row block of a C/D fragment IS the B
			// operand of a rank-4 product, and the normalised block's A part, read as an A operand, IS M = L[:, 4k..4k+3] (masked
			// below the block).  After four steps the A half holds L^T and the I half inverse(L).  Was: 16 pivots of nine lane permutes
			// + 8 FMAs each in one dependent chain, 4.9 us per sub-block (8 x 4.9 of the kernel's 57 us).
			const int i = r16;
			v4 RA, RW;
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int row = g + 4 * r;
				RA[r] = (i <= row) ? S[tri(o + row, o + i)] : S[tri(o + i, o + row)];
				RW[r] = (row == i) ? T(1) : T(0);
			}
			int first_bad = 0;
			const v4 zero4 = v4{0, 0, 0, 0};
#pragma unroll
			for (int k = 0; k < 4; ++k) {
				const int c0 = 4 * k;
				// D[a][b] sits in reg k of lane (a, 4k + b)
				T d00 = bcast(RA[k], 0 * 16 + c0 + 0);
				const T d10 = bcast(RA[k], 1 * 16 + c0 + 0), d20 = bcast(RA[k], 2 * 16 + c0 + 0), d30 = bcast(RA[k], 3 * 16 + c0 + 0);
				T d11 = bcast(RA[k], 1 * 16 + c0 + 1);
				const T d21 = bcast(RA[k], 2 * 16 + c0 + 1), d31 = bcast(RA[k], 3 * 16 + c0 + 1);
				T d22 = bcast(RA[k], 2 * 16 + c0 + 2);
				const T d32 = bcast(RA[k], 3 * 16 + c0 + 2);
				T d33 = bcast(RA[k], 3 * 16 + c0 + 3);
				// 1/sqrt(d) from the hardware estimate + two Newton steps; l = d * rl; a failing pivot is replaced by 1 and reported once
#define STPY_PIVOT(d, rl, jj) \
				{ const bool bad_ = !(d > T(0)) || !(d < T(1e300)); first_bad = (bad_ && first_bad == 0) ? c0 + jj + 1 : first_bad; d = bad_ ? T(1) : d; } \
				T rl = (T)__builtin_amdgcn_rsq(d); { const T hd_ = T(0.5) * d; rl = rl * (T(1.5) - hd_ * rl * rl); rl = rl * (T(1.5) - hd_ * rl * rl); }
				STPY_PIVOT(d00, r0, 0)
				const T l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
				d11 -= l10 * l10;
				STPY_PIVOT(d11, r1, 1)
				const T l21 = (d21 - l20 * l10) * r1, l31 = (d31 - l30 * l10) * r1;
				d22 -= l20 * l20 + l21 * l21;
				STPY_PIVOT(d22, r2, 2)
				const T l32 = (d32 - l30 * l20 - l31 * l21) * r2;
				d33 -= l30 * l30 + l31 * l31 + l32 * l32;
				STPY_PIVOT(d33, r3, 3)
#undef STPY_PIVOT
				// X = inverse(L_kk), lower 4 x 4 (1 / l_jj = the reciprocal square roots)
				const T x10 = -r1 * (l10 * r0), x21 = -r2 * (l21 * r1), x32 = -r3 * (l32 * r2);
				const T x20 = -r2 * (l20 * r0 + l21 * x10), x31 = -r3 * (l31 * r1 + l32 * x21);
				const T x30 = -r3 * (l30 * r0 + l31 * x10 + l32 * x20);
				// A operand of X padded to 16 x 4: lane (g, i) holds X[i][g] (i < 4, g <= i), zero elsewhere
				T xs = T(0);
				xs = (i == 0 && g == 0) ? r0 : xs;
				xs = (i == 1 && g == 0) ? x10 : xs;
				xs = (i == 1 && g == 1) ? r1 : xs;
				xs = (i == 2 && g == 0) ? x20 : xs;
				xs = (i == 2 && g == 1) ? x21 : xs;
				xs = (i == 2 && g == 2) ? r2 : xs;
				xs = (i == 3 && g == 0) ? x30 : xs;
				xs = (i == 3 && g == 1) ? x31 : xs;
				xs = (i == 3 && g == 2) ? x32 : xs;
				xs = (i == 3 && g == 3) ? r3 : xs;
				// normalised row block k: rows 0..3 of X_pad * R_k, i.e. reg 0 of the product, lane (g, i) = row g
				v4 dn0 = MM::mma(xs, RA[k], zero4);
				v4 dn1 = MM::mma(xs, RW[k], zero4);
				// Both 8-register results are kept whole and simultaneously live: only element 0 of each is used, and left to itself
				// hipcc packs the two destination tuples so that they OVERLAP (v[28:35] / v[30:37]) -- two in-flight fp64 MFMAs with
				// partially overlapping destinations and a constant C are not ordered by the hardware the way the compiler assumes
				// (measured: rows 4k..4k+3 of a sub-block wrong in ~3 of 128 diagonal blocks, only beside other MFMA traffic).
				asm volatile("" : "+v"(dn0), "+v"(dn1));
				const T n0 = dn0[0], n1 = dn1[0];
				if (k < 3) {
					// M = L[:, 4k..4k+3] below the block: lane (g, i) needs M[i][g] = (L^T)[4k + g][i] = n0 of this very lane
					const T mop = (i >= c0 + 4) ? n0 : T(0);
					RA = MM::mms(mop, n0, RA);
					RW = MM::mms(mop, n1, RW);
				}
				RA[k] = n0;
				RW[k] = n1;
			}
			if (first_bad != 0 && lane == 0) atomicCAS(info, 0, block_row0 + o + first_bad);
			// A half = L^T: lane (g, i) reg r = L[i][g + 4r];  I half = inverse(L): reg r = W[g + 4r][i]
#pragma unroll
			for (int r = 0; r < 4; ++r) {
				const int c = g + 4 * r;
				if (c <= i) S[tri(o + i, o + c)] = RA[r];
				WD[(kb * SB + c) * WLD + i] = (i <= c) ? RW[r] : T(0);
			}
		} else {
			const int q = lane >> 4, i = lane & 15;
			T a[4], w[4];
	#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				a[c] = (col <= i) ? S[tri(o + i, o + col)] : T(0);
				w[c] = (col == i) ? T(1) : T(0);
			}
			int first_bad = 0;
	#pragma unroll
			for (int j = 0; j < SB; ++j) {
				const int qj = j >> 2, cj = j & 3;
				T d = bcast(a[cj], 16 * qj + j);
				// (no branch here: one basic block over all 16 pivots lets the scheduler start the next pivot's reciprocal
				// square root under the tail of this pivot's update; the failing pivot is reported once, after the loop)
				const bool bad = !(d > T(0)) || !(d < T(1e300));
				first_bad = (bad && first_bad == 0) ? j + 1 : first_bad;
				d = bad ? T(1) : d;
				// 1/sqrt(d) from the hardware estimate + two Newton steps, l = d * rl (no sqrt + division on the pivot chain)
				T rl = (T)__builtin_amdgcn_rsq(d);
				rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
				rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
				const T l = d * rl;
				// scaled column j (meaningful on the lanes of group qj, rows >= j)
				const T colv = (i == j) ? l : a[cj] * rl;
				a[cj] = (q == qj && i >= j) ? colv : a[cj];
				// all nine lane permutes first, then straight-line selects: with `if`s hipcc builds an exec-mask region per
				// column and waits for each permute on its own
				const T mi = __shfl(colv, 16 * qj + i, 64);                  // l_ij of this lane's row
				T lk[4], wj[4];
	#pragma unroll
				for (int c = 0; c < 4; ++c) {
					lk[c] = __shfl(colv, 16 * qj + 4 * q + c, 64);           // l_kj of this lane's column k = 4q + c
					wj[c] = __shfl(w[c], 16 * q + j, 64);                    // pivot row of the inverse
				}
	#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int k = 4 * q + c;
					const T na = a[c] - mi * lk[c];
					a[c] = (k > j && i >= k) ? na : a[c];
					const T ws = wj[c] * rl;
					const T nw = w[c] - mi * ws;
					w[c] = (i == j) ? ws : ((i > j) ? nw : w[c]);
				}
			}
			if (first_bad != 0 && lane == 0) atomicCAS(info, 0, block_row0 + o + first_bad);
	#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				if (col <= i) S[tri(o + i, o + col)] = a[c];
				WD[(kb * SB + i) * WLD + col] = (col <= i) ? w[c] : T(0);       // W[row i][col]
			}
		}
	};
	// ---- one trailing sub-block (bi >= bj > kb): A[bi][bj] -= X_bi X_bj^T  (diagonal sub-blocks: lower part only)
	auto trail_pair = [&](int kb, int bi, int bj) {
		const int o = kb * SB;
		const bool diag = bi == bj;
		v4 acc;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int rr = MM::crow(lane, q);
			acc[q] = (!diag || r16 <= rr) ? S[tri(bi * SB + rr, bj * SB + r16)] : T(0);
		}
#pragma unroll
		for (int s4 = 0; s4 < 4; ++s4)
			acc = MM::mma(-S[tri(bi * SB + r16, o + 4 * s4 + g)], S[tri(bj * SB + r16, o + 4 * s4 + g)], acc);
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int rr = MM::crow(lane, q);
			if (!diag || r16 <= rr) S[tri(bi * SB + rr, bj * SB + r16)] = acc[q];
		}
	};

	// One sub-block of look-ahead: after panel kb, the sub-block column kb+1 is updated first; wave 0 then factors the
	// next diagonal sub-block (the serial part: 6.6 us of every step) while the other seven waves finish the update.
	if (wave == 0) diag_block(0);
	for (int kb = 0; kb < NSB; ++kb) {
		const int o = kb * SB;
		__syncthreads();
		// ---- panel below: X_bi = A_bi * WD^T   (one sub-block per wave; every element is strictly below the diagonal)
		for (int bi = kb + 1 + wave; bi < NSB; bi += 8) {
			v4 acc = v4{0, 0, 0, 0};
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4)
				acc = MM::mma(S[tri(bi * SB + r16, o + 4 * s4 + g)], WD[(kb * SB + r16) * WLD + 4 * s4 + g], acc);
#pragma unroll
			for (int q = 0; q < 4; ++q) S[tri(bi * SB + MM::crow(lane, q), o + r16)] = acc[q];
		}
		__syncthreads();
		if (kb + 1 < NSB) {
			if (kb + 1 + wave < NSB) trail_pair(kb, kb + 1 + wave, kb + 1);      // column kb+1 (wave 0: its diagonal sub-block)
			__syncthreads();
			if (wave == 0) diag_block(kb + 1);
			else {
				const int nrem = NSB - 2 - kb, npair = nrem * (nrem + 1) / 2;
				for (int pidx = wave - 1; pidx < npair; pidx += 7) {
					int a = 0, rem = pidx;
					while (rem > a) { rem -= a + 1; ++a; }
					trail_pair(kb, kb + 2 + a, kb + 2 + rem);
				}
			}
		}
	}
	__syncthreads();
#ifdef STPY_STAMPS
	const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif

	// ---- write L back (and the panel copy with explicit zeros above the diagonal); inverse(L):
	// ---- zeros above the diagonal and the diagonal sub-blocks now, off-diagonal blocks below
	for (int idx = tid; idx < IB * IB; idx += PT_THREADS) {
		const int i = idx >> 7, j = idx & 127;
		const T v = (j <= i) ? S[tri(i, j)] : T(0);
		if (i < nbk && j < nbk) {
			if (j <= i) A[(int64_t)i * lda + j] = v;
			if (P2) P2[(int64_t)i * ldp2 + j] = v;
		}
		const int bi = i >> 4, bj = j >> 4;
		if (bj > bi) W[idx] = T(0);
		else if (bj == bi) W[idx] = WD[(bi * SB + (i & 15)) * WLD + (j & 15)];
	}
	// (no barrier needed: the inverse below reads S and WD only, and writes W blocks nobody else touches)
#ifdef STPY_STAMPS
	const unsigned long long stamp_t3 = __builtin_amdgcn_s_memrealtime();
#endif

	// ---- triangular inverse, block column `wave`; W_ij goes to global memory, later steps of the
	// ---- same wave read W_kj back from there
	if (wave < NSB - 1) {
		const int j = wave;
		if constexpr (sizeof(T) == 8) {
			// fp64: the C/D fragment of a finished block (reg s of lane l = element [(l>>4) + 4s][l&15]) IS the B operand
			// of k-step s, so the blocks W_kj this wave has produced stay in registers and feed the next products
			// directly -- no trip through memory between the steps of the chain.  Loops are indexed relative to j so
			// that every register array index is a compile-time constant.
			v4 wcol[NSB - 1];        // wcol[kk - 1] = W_{j+kk, j}
#pragma unroll
			for (int ii = 1; ii < NSB; ++ii) {
				const int i = j + ii;
				if (i < NSB) {
					v4 t = v4{0, 0, 0, 0};
#pragma unroll
					for (int kk = 0; kk < ii; ++kk) {
						const int k = j + kk;
#pragma unroll
						for (int s4 = 0; s4 < 4; ++s4) {
							const T a = S[tri(i * SB + r16, k * SB + 4 * s4 + g)];                      // L_ik[r16][kk]
							const T b = (kk == 0) ? WD[(j * SB + 4 * s4 + g) * WLD + r16] : wcol[kk - 1][s4];    // W_kj[kk][c]
							t = MM::mma(a, b, t);
						}
					}
					v4 w = v4{0, 0, 0, 0};
#pragma unroll
					for (int s4 = 0; s4 < 4; ++s4) w = MM::mma(-WD[(i * SB + r16) * WLD + 4 * s4 + g], t[s4], w);
					wcol[ii - 1] = w;
#pragma unroll
					for (int q = 0; q < 4; ++q) W[(i * SB + MM::crow(lane, q)) * IB + j * SB + r16] = w[q];
				}
			}
		} else {
		T* sc = SC + wave * SB * WLD;
		for (int i = j + 1; i < NSB; ++i) {
			v4 t = v4{0, 0, 0, 0};
			for (int k = j; k < i; ++k) {
#pragma unroll
				for (int s4 = 0; s4 < 4; ++s4) {
					const T a = S[tri(i * SB + r16, k * SB + 4 * s4 + g)];                            // L_ik[r16][kk]
					const T b = (k == j) ? WD[(j * SB + 4 * s4 + g) * WLD + r16]                      // W_jj[kk][c]
					                     : W[(k * SB + 4 * s4 + g) * IB + j * SB + r16];              // W_kj[kk][c]
					t = MM::mma(a, b, t);
				}
			}
			v4 w = v4{0, 0, 0, 0};
#pragma unroll
			for (int q = 0; q < 4; ++q) sc[MM::crow(lane, q) * WLD + r16] = t[q];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4) w = MM::mma(-WD[(i * SB + r16) * WLD + 4 * s4 + g], sc[(4 * s4 + g) * WLD + r16], w);
#pragma unroll
			for (int q = 0; q < 4; ++q) W[(i * SB + MM::crow(lane, q)) * IB + j * SB + r16] = w[q];
			// the next step of this wave re-reads these values: stores drained, then ordered before the loads
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
		}
		}
	}
#ifdef STPY_STAMPS
	__syncthreads();
	if (tid == 0 && stpy_dbg_potf2) {
		const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
		const unsigned slot = atomicAdd((unsigned*)stpy_dbg_potf2, 1u);
		if (slot < 1000) { unsigned long long* d = stpy_dbg_potf2 + 2 + 8 * slot; d[0] = stamp_t0; d[1] = stamp_t1; d[2] = stamp_t2; d[3] = stamp_t3; d[4] = t1; }
	}
#endif
}

// ------------------------------------------------------------------------------------------
// fp64 diagonal-block kernel, round 4 ("flow" form): the same mathematics as potf2_trtri_mfma_kernel<double> above with the
// latency chain cut down.  What bounds this kernel is ONE dependent chain -- sub-block kb's factor -> the 16 rows below it ->
// their diagonal update -> sub-block kb + 1's factor -> ... -- and everything that is not on it should not be in its way:
//   * the critical wave (7) does that chain itself: it takes the panel rows of sub-block kb + 1, applies them to the diagonal
//     sub-block (kb + 1, kb + 1) and factors it, with two workgroup barriers per step instead of three (3.25 -> ~2.3 us per step);
//   * the 16 x 16 factor + inverse is the 4 x 4-blocked MFMA elimination (diag16_f64 below);
//   * the other seven waves do the remaining panel rows, all other trailing sub-blocks AND the triangular inverse while the
//     critical wave factors: row i of inverse(L) is formed right after sub-block i is final (W_ij = -WD_i sum_k L_ik W_kj, finished
//     blocks go to the output array and are read back through the L2 by whichever wave needs them -- the workgroup barriers of the
//     steps order those accesses); the 6.4 us inverse phase after the factorisation shrinks to the last row's products;
//   * the critical wave loads sub-block (0, 0) straight from global memory into its registers and starts on it while the other
//     waves bring the rest of the block into LDS.
// LDS: the packed lower triangle + the eight inverse diagonal sub-blocks, 83 KiB as before (must fit beside one update workgroup).
// ------------------------------------------------------------------------------------------
struct Diag16 {
	typedef Mfma<double> MM;
	typedef MM::v4 v4;
	// R = [A | I] (16 x 32) as two C/D fragments (reg r of lane (g, i) = row g + 4r, column i) -> A half = L^T, I half = inverse(L).
	// The pivot test is a side computation (one compare per pivot, the flag kept in scalar registers): nothing on the dependent
	// chain waits for it and the pivot is NOT replaced -- a pivot that is not positive turns its reciprocal square root into NaN /
	// inf and with it the rest of the block (also the rows above it: the masked rank-4 updates multiply 0 by NaN), which is why the
	// index of the FIRST failing pivot is recorded here and not recovered from the result.  Returns 0 or that 1-based index.
	static __device__ __forceinline__ int run(v4& RA, v4& RW, int g, int i)
	{
		int first_bad = 0;
		const v4 zero4 = v4{0, 0, 0, 0};
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			const int c0 = 4 * k;
			// D[a][b] sits in reg k of lane (a, 4k + b)
			const double d00 = bcast(RA[k], 0 * 16 + c0 + 0);
			const double d10 = bcast(RA[k], 1 * 16 + c0 + 0), d20 = bcast(RA[k], 2 * 16 + c0 + 0), d30 = bcast(RA[k], 3 * 16 + c0 + 0);
			double d11 = bcast(RA[k], 1 * 16 + c0 + 1);
			const double d21 = bcast(RA[k], 2 * 16 + c0 + 1), d31 = bcast(RA[k], 3 * 16 + c0 + 1);
			double d22 = bcast(RA[k], 2 * 16 + c0 + 2);
			const double d32 = bcast(RA[k], 3 * 16 + c0 + 2);
			double d33 = bcast(RA[k], 3 * 16 + c0 + 3);
			// 1/sqrt(d): hardware estimate + two Newton steps; l = d * rl and 1/l = rl
#define STPY_RSQ(d, rl, jj) first_bad = (!(d > 0.0) && first_bad == 0) ? c0 + jj + 1 : first_bad; \
			double rl = __builtin_amdgcn_rsq(d); { const double hd_ = 0.5 * d; rl = rl * (1.5 - hd_ * rl * rl); rl = rl * (1.5 - hd_ * rl * rl); }
			STPY_RSQ(d00, r0, 0)
			const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
			d11 -= l10 * l10;
			STPY_RSQ(d11, r1, 1)
			const double l21 = (d21 - l20 * l10) * r1, l31 = (d31 - l30 * l10) * r1;
			d22 -= l20 * l20 + l21 * l21;
			STPY_RSQ(d22, r2, 2)
			const double l32 = (d32 - l30 * l20 - l31 * l21) * r2;
			d33 -= l30 * l30 + l31 * l31 + l32 * l32;
			STPY_RSQ(d33, r3, 3)
#undef STPY_RSQ
			// X = inverse(L_kk), lower 4 x 4
			const double x10 = -r1 * (l10 * r0), x21 = -r2 * (l21 * r1), x32 = -r3 * (l32 * r2);
			const double x20 = -r2 * (l20 * r0 + l21 * x10), x31 = -r3 * (l31 * r1 + l32 * x21);
			const double x30 = -r3 * (l30 * r0 + l31 * x10 + l32 * x20);
			// A operand of X padded to 16 x 4: lane (g, i) holds X[i][g] (i < 4, g <= i), zero elsewhere
			double xs = 0.0;
			xs = (i == 0 && g == 0) ? r0 : xs;
			xs = (i == 1 && g == 0) ? x10 : xs;
			xs = (i == 1 && g == 1) ? r1 : xs;
			xs = (i == 2 && g == 0) ? x20 : xs;
			xs = (i == 2 && g == 1) ? x21 : xs;
			xs = (i == 2 && g == 2) ? r2 : xs;
			xs = (i == 3 && g == 0) ? x30 : xs;
			xs = (i == 3 && g == 1) ? x31 : xs;
			xs = (i == 3 && g == 2) ? x32 : xs;
			xs = (i == 3 && g == 3) ? r3 : xs;
			// normalised row block k = rows 0..3 of X_pad * R_k: reg 0 of the product, lane (g, i) = row g
			v4 dn0 = MM::mma(xs, RA[k], zero4);
			v4 dn1 = MM::mma(xs, RW[k], zero4);
			// both results whole and live together: see potf2_trtri_mfma_kernel (overlapping destination tuples of two MFMAs in flight)
			asm volatile("" : "+v"(dn0), "+v"(dn1));
			const double n0 = dn0[0], n1 = dn1[0];
			if (k < 3) {
				// M = L[:, 4k..4k+3] below the block: lane (g, i) needs M[i][g] = (L^T)[4k + g][i] = n0 of this very lane
				const double mop = (i >= c0 + 4) ? n0 : 0.0;
				RA = MM::mms(mop, n0, RA);
				RW = MM::mms(mop, n1, RW);
			}
			RA[k] = n0;
			RW[k] = n1;
		}
		return first_bad;
	}
};

// TS: the type of the matrix in memory.  The arithmetic is fp64 for both -- the kernel is a latency chain, not a throughput kernel, and an fp32
// block factored in fp64 is rounded once, at the store (round 4: the fp32 factorisation's diagonal blocks, 90 us in the fp32 form of the
// kernel above, take this kernel too).
template <typename TS>
__global__ __launch_bounds__(PT_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))
void potf2_trtri_flow_kernel(TS* __restrict__ A, int64_t lda, int nbk, TS* W,
                             TS* __restrict__ P2, int64_t ldp2, int32_t* info, int block_row0)
{
	typedef double T;
	typedef Mfma<double> MM;
	typedef MM::v4 v4;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* S = reinterpret_cast<T*>(smem_raw);          // [TRI] packed lower triangle
	T* WD = S + TRI;                                // [8][16][WLD]
#ifdef STPY_STAMPS
	const unsigned long long stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (uniform: block / task indices derived from it live in SGPRs)
	const int r16 = lane & 15, g = lane >> 4;
	constexpr int CW = 7;                           // the critical wave

	// ---- factor + inverse of the diagonal sub-block kb by the critical wave; input from S, or (kb = 0) straight from global memory
	auto diag_block = [&](int kb, bool from_global) {
		const int o = kb * SB;
		v4 RA, RW;
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int row = g + 4 * r;
			const int hi = row > r16 ? row : r16, lo = row > r16 ? r16 : row;          // the stored (lower) element of the symmetric pair
			if (from_global) RA[r] = (hi < nbk) ? (T)A[(int64_t)(o + hi) * lda + o + lo] : (hi == lo ? T(1) : T(0));
			else RA[r] = S[tri(o + hi, o + lo)];
			RW[r] = (row == r16) ? T(1) : T(0);
		}
		const int fb = Diag16::run(RA, RW, g, r16);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const int c = g + 4 * r;
			if (c <= r16) S[tri(o + r16, o + c)] = RA[r];
			WD[(kb * SB + c) * WLD + r16] = (r16 <= c) ? RW[r] : T(0);
		}
		if (fb != 0 && lane == 0) atomicCAS(info, 0, block_row0 + o + fb);
	};
	// ---- 16 rows of the panel below sub-block kb: X_bi = A_bi * WD_kb^T (in place in S)
	auto panel_rows = [&](int kb, int bi) {
		const int o = kb * SB;
		v4 acc = v4{0, 0, 0, 0};
#pragma unroll
		for (int s4 = 0; s4 < 4; ++s4)
			acc = MM::mma(S[tri(bi * SB + r16, o + 4 * s4 + g)], WD[(kb * SB + r16) * WLD + 4 * s4 + g], acc);
#pragma unroll
		for (int q = 0; q < 4; ++q) S[tri(bi * SB + MM::crow(lane, q), o + r16)] = acc[q];
	};
	// ---- one trailing sub-block (bi >= bj > kb): A[bi][bj] -= X_bi X_bj^T  (diagonal sub-blocks: lower part only; the off-diagonal
	// ---- form carries no lane predicates -- with them hipcc wraps every LDS access in an exec-mask region)
	auto trail_pair = [&](int kb, int bi, int bj) {
		const int o = kb * SB;
		v4 acc;
		if (bi == bj) {
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int rr = MM::crow(lane, q);
				acc[q] = (r16 <= rr) ? S[tri(bi * SB + rr, bj * SB + r16)] : T(0);
			}
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4)
				acc = MM::mms(S[tri(bi * SB + r16, o + 4 * s4 + g)], S[tri(bj * SB + r16, o + 4 * s4 + g)], acc);
#pragma unroll
			for (int q = 0; q < 4; ++q) {
				const int rr = MM::crow(lane, q);
				if (r16 <= rr) S[tri(bi * SB + rr, bj * SB + r16)] = acc[q];
			}
		} else {
#pragma unroll
			for (int q = 0; q < 4; ++q) acc[q] = S[tri(bi * SB + MM::crow(lane, q), bj * SB + r16)];
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4)
				acc = MM::mms(S[tri(bi * SB + r16, o + 4 * s4 + g)], S[tri(bj * SB + r16, o + 4 * s4 + g)], acc);
#pragma unroll
			for (int q = 0; q < 4; ++q) S[tri(bi * SB + MM::crow(lane, q), bj * SB + r16)] = acc[q];
		}
	};
	// ---- row block rb (16 rows) is final once its diagonal sub-block is factored: L to A (lower part) and to the panel copy (zeros
	// ---- above the diagonal), and of inverse(L) the diagonal sub-block and the zeros to its right (the blocks to its left come from
	// ---- inverse_block).  `t` of `nt` threads share the 16 x 128 elements.
	auto writeback_rows = [&](int rb, int t, int nt) {
		for (int e = t; e < SB * IB; e += nt) {
			const int i = rb * SB + (e >> 7), j = e & 127;
			const T v = (j <= i) ? S[tri(i, j)] : T(0);
			if (i < nbk && j < nbk) {
				if (j <= i) A[(int64_t)i * lda + j] = (TS)v;
				if (P2) P2[(int64_t)i * ldp2 + j] = (TS)v;
			}
			const int bj = j >> 4;
			if (bj > rb) W[i * IB + j] = TS(0);
			else if (bj == rb) W[i * IB + j] = (TS)WD[(rb * SB + (i & 15)) * WLD + (j & 15)];
		}
	};

	// ---- load: the critical wave takes sub-block (0, 0) from global memory and factors it at once; the other 448 threads bring the
	// ---- lower triangle (without that sub-block) into LDS, identity on the padding, in batches of loads that are in flight together
	if (wave == CW) {
		diag_block(0, true);
	} else {
		// the packed triangle is walked as 64 row pairs (p, 127 - p) of 129 elements, so that every thread has the same 19 loads
		// to issue, all in flight together (one memory round trip)
		constexpr int NLD = PT_THREADS - 64, NEL = (IB / 2) * (IB + 1), NIT = (NEL + NLD - 1) / NLD;
		T v[NIT];
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int e = tid + it * NLD, p = e / (IB + 1), q = e - p * (IB + 1);
			const int i = (q <= p) ? p : IB - 1 - p, j = (q <= p) ? q : q - p - 1;
			const bool want = e < NEL && i >= SB;                                 // (rows < 16 hold sub-block (0, 0) only)
			v[it] = (want && i < nbk) ? (T)A[(int64_t)i * lda + j] : ((i == j) ? T(1) : T(0));
		}
#pragma unroll
		for (int it = 0; it < NIT; ++it) {
			const int e = tid + it * NLD, p = e / (IB + 1), q = e - p * (IB + 1);
			const int i = (q <= p) ? p : IB - 1 - p, j = (q <= p) ? q : q - p - 1;
			if (e < NEL && i >= SB) S[tri(i, j)] = v[it];
		}
	}
#ifdef STPY_STAMPS
	const unsigned long long stamp_t1 = __builtin_amdgcn_s_memrealtime();
#endif

	// ---- the steps.  Barrier 1: sub-block kb's factor / inverse and every update of step kb - 1 are visible.  Barrier 2: the panel
	// ---- rows of step kb are.  Between them all eight waves take panel rows (the critical wave those of sub-block kb + 1); after
	// ---- barrier 2 the critical wave updates and factors sub-block kb + 1 while the others share the remaining updates
	// ---- (column kb + 1 first) and row kb of the inverse.
#ifdef STPY_STAMPS
	unsigned long long cw_b1 = 0, cw_b2 = 0, cw_de = 0, cw_nb1 = 0, ow_b2 = 0, ow_tr = 0, ow_wb = 0, ow_iv = 0;
#endif
	// The barriers of the steps order LDS only (S, WD): nothing read in this kernel was written to global memory by it, so no wave
	// waits for the L / inverse rows on their way to memory (a microsecond or two until they are acknowledged).  The critical wave
	// and the others run two separate copies of the step loop with the same barrier sequence (the inverse column a wave keeps in
	// registers must not be live across the critical wave's factor code, or both spill).
#define STPY_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
	if (wave == CW) {
		for (int kb = 0; kb < NSB; ++kb) {
			STPY_LDS_BARRIER();
#ifdef STPY_STAMPS
			if (kb == 3) cw_b1 = __builtin_amdgcn_s_memrealtime();
			if (kb == 4) cw_nb1 = __builtin_amdgcn_s_memrealtime();
#endif
			if (kb + 1 < NSB) panel_rows(kb, kb + 1);
			STPY_LDS_BARRIER();
#ifdef STPY_STAMPS
			if (kb == 3) cw_b2 = __builtin_amdgcn_s_memrealtime();
#endif
			if (kb + 1 < NSB) {
				trail_pair(kb, kb + 1, kb + 1);
				diag_block(kb + 1, false);
			}
#ifdef STPY_STAMPS
			if (kb == 3) cw_de = __builtin_amdgcn_s_memrealtime();
#endif
		}
	} else {
		// ---- inverse(L), block column j = this wave's: W_ij = -WD_i * sum_{k=j}^{i-1} L_ik W_kj for i = j+1, j+2, ... -- one block per
		// ---- step, as soon as row block i of L is final.  The finished blocks of the column stay in this wave's registers (the C/D
		// ---- fragment of W_kj is the B operand the next product needs) and go to the output array with stores nobody waits for.
		v4 wcol[NSB - 1];          // wcol[kk - 1] = W_{j+kk, j}
		auto inverse_step = [&](auto nconst) {
			constexpr int n = decltype(nconst)::value;          // i = j + n
			const int j = wave, i = wave + n;
			v4 t = v4{0, 0, 0, 0};
#pragma unroll
			for (int kk = 0; kk < n; ++kk) {
				const int kx = j + kk;
#pragma unroll
				for (int s4 = 0; s4 < 4; ++s4) {
					const T av = S[tri(i * SB + r16, kx * SB + 4 * s4 + g)];                                       // L_ik[r16][4 s4 + g]
					const T bv = (kk == 0) ? WD[(j * SB + 4 * s4 + g) * WLD + r16] : wcol[kk == 0 ? 0 : kk - 1][s4];   // W_kj[4 s4 + g][r16]
					t = MM::mma(av, bv, t);
				}
			}
			v4 w = v4{0, 0, 0, 0};
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4) w = MM::mms(WD[(i * SB + r16) * WLD + 4 * s4 + g], t[s4], w);
			wcol[n - 1] = w;
#pragma unroll
			for (int q = 0; q < 4; ++q) W[(i * SB + MM::crow(lane, q)) * IB + j * SB + r16] = (TS)w[q];
		};
		for (int kb = 0; kb < NSB; ++kb) {
			STPY_LDS_BARRIER();
			if (kb + 2 + wave < NSB) panel_rows(kb, kb + 2 + wave);
			STPY_LDS_BARRIER();
#ifdef STPY_STAMPS
			if (kb == 3) ow_b2 = __builtin_amdgcn_s_memrealtime();
#endif
			// the updates are dealt from wave 6 down: a round that does not go round leaves the extra ones with the waves whose inverse
			// column is short or has not started
			int turn = 6;          // (a running counter: the modulo of a pair index costs a dozen scalar instructions per candidate pair)
			for (int bj = kb + 1; bj < NSB; ++bj)
				for (int bi2 = (bj == kb + 1 ? bj + 1 : bj); bi2 < NSB; ++bi2) {
					if (turn == wave) trail_pair(kb, bi2, bj);
					turn = turn == 0 ? 6 : turn - 1;
				}
#ifdef STPY_STAMPS
			if (kb == 3) ow_tr = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef STPY_STAMPS
			if (kb == 3) ow_wb = __builtin_amdgcn_s_memrealtime();
#endif
			switch (kb - wave) {          // block (kb, wave) of the inverse, if this wave's column has reached row kb
				case 1: inverse_step(std::integral_constant<int, 1>{}); break;
				case 2: inverse_step(std::integral_constant<int, 2>{}); break;
				case 3: inverse_step(std::integral_constant<int, 3>{}); break;
				case 4: inverse_step(std::integral_constant<int, 4>{}); break;
				case 5: inverse_step(std::integral_constant<int, 5>{}); break;
				case 6: inverse_step(std::integral_constant<int, 6>{}); break;
				case 7: inverse_step(std::integral_constant<int, 7>{}); break;
				default: break;
			}
#ifdef STPY_STAMPS
			if (kb == 3) ow_iv = __builtin_amdgcn_s_memrealtime();
#endif
		}
	}
#undef STPY_LDS_BARRIER
	__syncthreads();
#ifdef STPY_STAMPS
	const unsigned long long stamp_t2 = __builtin_amdgcn_s_memrealtime();
#endif

	// ---- write L back (and the panel copy with explicit zeros above the diagonal); of inverse(L) the zeros above the diagonal and the
	// ---- diagonal sub-blocks (the blocks below went out as they were formed).  Measured inside the step loop instead (each row block
	// ---- as soon as it is final, shared by the seven non-critical waves): ~1 us per row block and wave, more than those waves have
	// ---- to spare beside the critical wave's 3 us per step -- 39 us against 33 with the write-back here.
	const bool vec_ok = nbk == IB && ((((uintptr_t)A | (uintptr_t)W | (uintptr_t)(P2 ? P2 : A)) & (2 * sizeof(TS) - 1)) == 0) && (lda & 1) == 0 && (!P2 || (ldp2 & 1) == 0);
	if (vec_ok) {
		// two columns per thread and 16-byte (fp32 matrix: 8-byte) stores: thread t owns the column pair (2 jp, 2 jp + 1) in the rows rs, rs + 8, ... (the strict
		// upper triangle of A is scratch by contract, so the pair that straddles the diagonal may write its zero)
		typedef TS v2 __attribute__((ext_vector_type(2)));
		const int jp = tid & 63, rs = tid >> 6, j = 2 * jp, bj = j >> 4;
		for (int i = rs; i < IB; i += PT_THREADS / 64) {
			const int bi = i >> 4;
			v2 v;
			v.x = (TS)((j <= i) ? S[tri(i, j)] : T(0));
			v.y = (TS)((j + 1 <= i) ? S[tri(i, j + 1)] : T(0));
			if (j <= i) *(v2*)(A + (int64_t)i * lda + j) = v;
			if (P2) *(v2*)(P2 + (int64_t)i * ldp2 + j) = v;
			if (bj > bi) *(v2*)(W + i * IB + j) = v2{0, 0};
			else if (bj == bi) { v2 wv; wv.x = (TS)WD[i * WLD + (j & 15)]; wv.y = (TS)WD[i * WLD + (j & 15) + 1]; *(v2*)(W + i * IB + j) = wv; }
		}
	} else {
		for (int rb = 0; rb < NSB; ++rb) writeback_rows(rb, tid, PT_THREADS);
	}
#ifdef STPY_STAMPS
	const unsigned long long stamp_t3 = __builtin_amdgcn_s_memrealtime();
	__syncthreads();
	if (tid == 0 && stpy_dbg_potf2) {
		const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
		const unsigned slot = atomicAdd((unsigned*)stpy_dbg_potf2, 1u);
		if (slot < 1000) { unsigned long long* d = stpy_dbg_potf2 + 2 + 8 * slot; d[0] = stamp_t0; d[1] = stamp_t1; d[2] = stamp_t2; d[3] = stamp_t3; d[4] = t1; stpy_dbg_potf2[1] = slot; }
	}
	// step 3 seen from the critical wave: barrier 1 -> barrier 2 (panel rows), barrier 2 -> its sub-block factored, -> next barrier 1 (waiting for the others)
	__syncthreads();
	if (tid == CW * 64 && stpy_dbg_potf2) {
		const unsigned slot = (unsigned)stpy_dbg_potf2[1];
		if (slot < 1000 && !stpy_dbg_potf2[1000 * 8 + 2]) { unsigned long long* d = stpy_dbg_potf2 + 2 + 8 * slot; d[5] = cw_b2 - cw_b1; d[6] = cw_de - cw_b2; d[7] = cw_nb1 - cw_de; }
	}
	// ... or (debug word set) seen from wave 0: its trailing updates, its share of the write-back, its inverse block
	if (tid == 0 && stpy_dbg_potf2 && stpy_dbg_potf2[1000 * 8 + 2]) {
		const unsigned slot = (unsigned)stpy_dbg_potf2[1];
		if (slot < 1000) { unsigned long long* d = stpy_dbg_potf2 + 2 + 8 * slot; d[5] = ow_tr - ow_b2; d[6] = ow_wb - ow_tr; d[7] = ow_iv - ow_wb; }
	}
#endif
}

#if STPY_LAB
// ------------------------------------------------------------------------------------------
// "Sliver" form of the same kernel for blocks factored WHILE a trailing update floods the chip (look-ahead panels):
// 256 threads capped at 64 VGPRs and the same 83 KiB of LDS, i.e. it fits into what two update workgroups leave over on
// a CU (512 - 2*224 VGPRs, 160 - 2*32 KiB) and is placed at once, where the eight-wave / 128-VGPR form above has to wait
// for an update workgroup to exit -- 200-270 us per block in the kernel traces (tools/sliver_probe.hip: 79 us beside the
// update = 79 us alone).  Same algorithm; what changes with four waves and a quarter of the registers:
//   * the block is loaded in four batches of 16 values per thread;
//   * sub-block rows / trailing pairs are dealt over 4 (3) waves instead of 8 (7);
//   * the triangular inverse handles two block columns per wave, and finished blocks W_kj are re-read from the output
//     array (L2-hot, written by the same wave) instead of being kept in registers.
// ------------------------------------------------------------------------------------------
constexpr int PS_THREADS = 256, PS_WAVES = 4;
template <typename T>
__global__ __launch_bounds__(PS_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))
void potf2_trtri_sliver_kernel(T* __restrict__ A, int64_t lda, int nbk, T* W,
                               T* __restrict__ P2, int64_t ldp2, int32_t* info, int block_row0)
{
	typedef Mfma<T> MM;
	typedef typename MM::v4 v4;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
	T* S = reinterpret_cast<T*>(smem_raw);          // [TRI] packed lower triangle
	T* WD = S + TRI;                                // [8][16][WLD]
	T* SC = WD + NSB * SB * WLD;                    // [4 waves][16][WLD] scratch: C/D fragment -> B operand re-layout
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int r16 = lane & 15, g = lane >> 4;
	// beside two older, MFMA-saturating update waves per SIMD this kernel's serial pivot chain gets only leftover issue slots
	// at equal priority (kernel trace: 730-770 us beside the update, 94 us alone): top priority for its whole life
	__builtin_amdgcn_s_setprio(3);

	{
		// thread t owns column j = t & 127 and the rows i = (t >> 7) + 2 it: one base pointer and a uniform stride; eight loads in
		// flight per batch (the eight-wave kernel keeps all 32 of a thread in registers, which 64 VGPRs cannot hold)
		const int j = tid & 127, i0 = tid >> 7;
		const T* src = A + (int64_t)i0 * lda + j;
		const int64_t step = 2 * lda;
		constexpr int NIT = IB * IB / PS_THREADS, BATCH = 8;
#pragma unroll 1
		for (int b0 = 0; b0 < NIT; b0 += BATCH) {
			T v[BATCH];
#pragma unroll
			for (int it = 0; it < BATCH; ++it) {
				const int i = i0 + 2 * (b0 + it);
				v[it] = (j <= i && i < nbk) ? src[(int64_t)it * step] : ((i == j) ? T(1) : T(0));
			}
#pragma unroll
			for (int it = 0; it < BATCH; ++it) {
				const int i = i0 + 2 * (b0 + it);
				if (j <= i) S[tri(i, j)] = v[it];
			}
			src += BATCH * step;
		}
	}
	__syncthreads();

	// ---- 16x16 diagonal sub-block (one wave): factor and inverse together, four lanes per row (see the kernel above)
	auto diag_block = [&](int kb) {
		const int o = kb * SB;
		const int q = lane >> 4, i = lane & 15;
		T a[4], w[4];
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const int col = 4 * q + c;
			a[c] = (col <= i) ? S[tri(o + i, o + col)] : T(0);
			w[c] = (col == i) ? T(1) : T(0);
		}
		int first_bad = 0;
		// (the group loop stays rolled: sixteen unrolled pivots hoist so many lane-permute addresses and constants that the
		// kernel cannot stay within 64 VGPRs; inside a group the column index is a compile-time register index)
#pragma unroll 1
		for (int qj = 0; qj < 4; ++qj) {
#pragma unroll
			for (int cj = 0; cj < 4; ++cj) {
				const int j = 4 * qj + cj;
				T d = bcast(a[cj], 16 * qj + j);
				const bool bad = !(d > T(0)) || !(d < T(1e300));
				first_bad = (bad && first_bad == 0) ? j + 1 : first_bad;
				d = bad ? T(1) : d;
				T rl = (T)__builtin_amdgcn_rsq(d);
				rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
				rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
				const T l = d * rl;
				const T colv = (i == j) ? l : a[cj] * rl;
				a[cj] = (q == qj && i >= j) ? colv : a[cj];
				const T mi = __shfl(colv, 16 * qj + i, 64);
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int k = 4 * q + c;
					const T lk = __shfl(colv, 16 * qj + 4 * q + c, 64);
					const T wj = __shfl(w[c], 16 * q + j, 64);
					const T na = a[c] - mi * lk;
					a[c] = (k > j && i >= k) ? na : a[c];
					const T ws = wj * rl;
					const T nw = w[c] - mi * ws;
					w[c] = (i == j) ? ws : ((i > j) ? nw : w[c]);
				}
			}
		}
		if (first_bad != 0 && lane == 0) atomicCAS(info, 0, block_row0 + o + first_bad);
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const int col = 4 * q + c;
			if (col <= i) S[tri(o + i, o + col)] = a[c];
			WD[(kb * SB + i) * WLD + col] = (col <= i) ? w[c] : T(0);
		}
	};
	auto trail_pair = [&](int kb, int bi, int bj) {
		const int o = kb * SB;
		const bool diag = bi == bj;
		v4 acc;
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int rr = MM::crow(lane, q);
			acc[q] = (!diag || r16 <= rr) ? S[tri(bi * SB + rr, bj * SB + r16)] : T(0);
		}
#pragma unroll
		for (int s4 = 0; s4 < 4; ++s4)
			acc = MM::mma(-S[tri(bi * SB + r16, o + 4 * s4 + g)], S[tri(bj * SB + r16, o + 4 * s4 + g)], acc);
#pragma unroll
		for (int q = 0; q < 4; ++q) {
			const int rr = MM::crow(lane, q);
			if (!diag || r16 <= rr) S[tri(bi * SB + rr, bj * SB + r16)] = acc[q];
		}
	};

	// (one call site for the diagonal sub-block: the loop starts one step early, with only wave 0's first sub-block in it)
#pragma unroll 1
	for (int kb = -1; kb < NSB; ++kb) {
		const int o = kb * SB;
		if (kb >= 0) {
			__syncthreads();
			for (int bi = kb + 1 + wave; bi < NSB; bi += PS_WAVES) {           // panel below: X_bi = A_bi * WD^T
				v4 acc = v4{0, 0, 0, 0};
#pragma unroll
				for (int s4 = 0; s4 < 4; ++s4)
					acc = MM::mma(S[tri(bi * SB + r16, o + 4 * s4 + g)], WD[(kb * SB + r16) * WLD + 4 * s4 + g], acc);
#pragma unroll
				for (int q = 0; q < 4; ++q) S[tri(bi * SB + MM::crow(lane, q), o + r16)] = acc[q];
			}
			__syncthreads();
		}
		if (kb + 1 < NSB) {
			if (kb >= 0) {
				for (int bi = kb + 1 + wave; bi < NSB; bi += PS_WAVES) trail_pair(kb, bi, kb + 1);      // column kb+1 first
				__syncthreads();
			}
			if (wave == 0) diag_block(kb + 1);
			else if (kb >= 0) {
				const int nrem = NSB - 2 - kb, npair = nrem * (nrem + 1) / 2;
				for (int pidx = wave - 1; pidx < npair; pidx += PS_WAVES - 1) {
					int a = 0, rem = pidx;
					while (rem > a) { rem -= a + 1; ++a; }
					trail_pair(kb, kb + 2 + a, kb + 2 + rem);
				}
			}
		}
	}
	__syncthreads();

	for (int idx = tid; idx < IB * IB; idx += PS_THREADS) {
		const int i = idx >> 7, j = idx & 127;
		const T v = (j <= i) ? S[tri(i, j)] : T(0);
		if (i < nbk && j < nbk) {
			if (j <= i) A[(int64_t)i * lda + j] = v;
			if (P2) P2[(int64_t)i * ldp2 + j] = v;
		}
		const int bi = i >> 4, bj = j >> 4;
		if (bj > bi) W[idx] = T(0);
		else if (bj == bi) W[idx] = WD[(bi * SB + (i & 15)) * WLD + (j & 15)];
	}

	// ---- triangular inverse: wave w owns block columns w and w + 4; W_ij = -WD_i * sum_{k=j}^{i-1} L_ik W_kj, the finished
	// ---- blocks of the column re-read from W (stores drained and ordered before the loads of the next step)
	T* sc = SC + wave * SB * WLD;
	for (int j = wave; j < NSB - 1; j += PS_WAVES) {
		for (int i = j + 1; i < NSB; ++i) {
			v4 t = v4{0, 0, 0, 0};
			for (int k = j; k < i; ++k) {
#pragma unroll
				for (int s4 = 0; s4 < 4; ++s4) {
					const T a = S[tri(i * SB + r16, k * SB + 4 * s4 + g)];
					const T b = (k == j) ? WD[(j * SB + 4 * s4 + g) * WLD + r16] : W[(k * SB + 4 * s4 + g) * IB + j * SB + r16];
					t = MM::mma(a, b, t);
				}
			}
			v4 w = v4{0, 0, 0, 0};
#pragma unroll
			for (int q = 0; q < 4; ++q) sc[MM::crow(lane, q) * WLD + r16] = t[q];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
#pragma unroll
			for (int s4 = 0; s4 < 4; ++s4) w = MM::mma(-WD[(i * SB + r16) * WLD + 4 * s4 + g], sc[(4 * s4 + g) * WLD + r16], w);
#pragma unroll
			for (int q = 0; q < 4; ++q) W[(i * SB + MM::crow(lane, q)) * IB + j * SB + r16] = w[q];
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
			__builtin_amdgcn_wave_barrier();          // (sc is rewritten by the next step only after every lane has read it)
		}
	}
}

#endif  // STPY_LAB (four-wave diagonal-block kernel)

template <typename T>
int potf2_trtri(T* A, int64_t lda, int nbk, T* W, T* P2, int64_t ldp2, int32_t* info, int block_row0, hipStream_t st, bool beside)
{
	const size_t lds_new = (size_t)(TRI + NSB * SB * WLD + (sizeof(T) == 4 ? 8 * SB * WLD : 0)) * sizeof(T);
	const size_t lds_flow = (size_t)(TRI + NSB * SB * WLD) * sizeof(double);          // (the flow kernel computes in fp64 whatever the matrix type)
	static std::atomic<bool> attr_set[2];          // (idempotent: two threads racing here both set the same attribute values)
	const int which = sizeof(T) == 8 ? 0 : 1;
#if STPY_LAB
	const size_t lds_old = (size_t)(IB * SLD + 2 * IB) * sizeof(T);
	const size_t lds_sliver = (size_t)(TRI + NSB * SB * WLD + PS_WAVES * SB * WLD) * sizeof(T);
#endif
	if (!attr_set[which].load(std::memory_order_acquire)) {
		hipError_t e = hipFuncSetAttribute((const void*)potf2_trtri_mfma_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)potf2_trtri_flow_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_flow);
#if STPY_LAB
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)potf2_trtri_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)potf2_trtri_sliver_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_sliver);
#endif
		if (e != hipSuccess) { set_error("potf2: hipFuncSetAttribute(%zu B LDS) failed: %s", lds_new, hipGetErrorString(e)); return -1000 - (int)e; }
		attr_set[which].store(true, std::memory_order_release);
	}
#if STPY_LAB
	if (g_potf2_scalar)
		hipLaunchKernelGGL((potf2_trtri_kernel<T>), dim3(1), dim3(PT_THREADS), lds_old, st, A, lda, nbk, W, P2, ldp2, info, block_row0);
	else if (beside && g_potf2_sliver)
		hipLaunchKernelGGL((potf2_trtri_sliver_kernel<T>), dim3(1), dim3(PS_THREADS), lds_sliver, st, A, lda, nbk, W, P2, ldp2, info, block_row0);
	else
#endif
	if (g_potf2_flow && (sizeof(T) == 8 || g_potf2_flow != 2)) hipLaunchKernelGGL((potf2_trtri_flow_kernel<T>), dim3(1), dim3(PT_THREADS), lds_flow, st, A, lda, nbk, W, P2, ldp2, info, block_row0);
	else hipLaunchKernelGGL((potf2_trtri_mfma_kernel<T>), dim3(1), dim3(PT_THREADS), lds_new, st, A, lda, nbk, W, P2, ldp2, info, block_row0);
	(void)beside;
	return check_launch("potf2_trtri");
}

// Side stream + events for the look-ahead: one set per (device, caller stream), created on first use under a
// mutex and never destroyed (a process drives a handful of streams).  The side stream has the highest priority so
// the small panel kernels are dispatched ahead of the queued trailing-update workgroups as CU slots free up.
// Two host threads on two streams get two sets; calls on one stream are enqueued in program order and the events
// of a set are only ever recorded / awaited by calls on that stream, so re-use across calls needs no further care.
// g_potrf_diag_first_below = 0:        stpy_tune key 7 (8192 until round 4: with the flow-form diagonal-block kernel and the sliver updates starting the
//                                       update only after the chain's first kernel no longer pays -- 6.13 -> 5.97 ms at N = 8192, 28.17 -> 28.08 at 16 384)
// stpy_tune key 10: a look-ahead panel whose trailing update still has at least this many rows runs BESIDE that update and
// takes the "sliver" GEMM for its panel products (gemm.hip: fits into what two update workgroups leave over on a CU, so it
// is placed at once) instead of the tile kernels, which wait for an update workgroup to exit, or the 128 KiB one-volley
// kernel, which needs a CU without any update workgroup.  tools/potrf_sweep.py, one process: with the threshold at 0 (every
// look-ahead panel) potrf takes 10.2 / 36.7 / 194.4 / 1396 ms at N = 8192 / 16 384 / 32 768 / 65 536 against
// 11.6 / 40.4 / 206.5 / 1411 ms without the beside mode.
// g_potrf_beside_min = 0 (lab knob, key 10)
// stpy_tune keys 12 / 13: trailing matrices with at most / at least this many rows are updated in reserved mode (see potrf());
// 12 = 0 (the default) switches the mode off.  Measured, one process (tools/potrf_sweep.py, gpurun_out/potrf_sweep6.log): with the
// mode on for every trailing matrix <= 24 576 rows potrf takes 10.5 / 38.7 / 208.0 / 1385 ms at N = 8192 / 16 384 / 32 768 /
// 65 536 against 10.1 / 36.5 / 195.9 / 1371 ms without: the diagonal-block kernel does run at its stand-alone speed on its
// reserved CU, but the update loses more (3 % of the CUs, 62 instead of 64 slots per XCD under a 64-tile super-tile map, and two
// event hand-overs per 128-column block) than the chain gains.  Kept switchable; not used.
// g_potrf_reserve_below = 0, g_potrf_reserve_above = 2048 (lab knobs, keys 12 / 13)
// stpy_tune keys 14 / 15.  With the panel products in "beside" mode the chain of a wide panel is cheaper than it was, and wide
// panels win (tools/potrf_sweep.py, gpurun_out/potrf_sweep8.log / 10.log, one process: N = 16 384 36.5 -> 34.4 ms, 32 768 196 -> 190 ms
// against the former 16384 / 32768 thresholds; N = 65 536 unchanged within 0.2 %)
// g_potrf_nb256_upto = 2048, g_potrf_nb512_upto = 16384 (lab knobs, keys 14 / 15)
namespace {
struct LaKey { int device; hipStream_t stream; bool operator<(const LaKey& o) const { return device != o.device ? device < o.device : stream < o.stream; } };
std::mutex g_la_mutex;
std::map<LaKey, LookAhead*> g_la_map;
}

int lookahead_acquire(hipStream_t caller, LookAhead** out)
{
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess) { set_error("potrf: hipGetDevice failed"); return -1001; }
	std::lock_guard<std::mutex> lock(g_la_mutex);
	const LaKey key{dev, caller};
	auto it = g_la_map.find(key);
	if (it != g_la_map.end()) { *out = it->second; return 0; }
	LookAhead* la = new LookAhead();
	int lo = 0, hi = 0;
	(void)hipDeviceGetStreamPriorityRange(&lo, &hi);
	if (hipStreamCreateWithPriority(&la->side, hipStreamNonBlocking, hi) != hipSuccess ||
	    hipEventCreateWithFlags(&la->col_ready, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&la->panel_done, hipEventDisableTiming) != hipSuccess ||
	    hipEventCreateWithFlags(&la->trail_done, hipEventDisableTiming) != hipSuccess) {
		set_error("potrf: could not create the look-ahead stream/events");
		delete la;
		return -1002;
	}
	if (hipMalloc(&la->trsv_sync, 64) != hipSuccess || hipMemset(la->trsv_sync, 0, 64) != hipSuccess) { (void)hipGetLastError(); la->trsv_sync = nullptr; }          // (the vector solves then take the step kernels)
	g_la_map[key] = la;
	*out = la;
	return 0;
}


#if STPY_LAB
// Created on first use only (the mode is off by default, stpy_tune key 12): masked streams alive at process exit have been
// seen to crash the profiler's finalisation.
static std::vector<LookAhead*> g_la_masked;          // objects that own CU-masked streams (guarded by g_la_mutex)
static void destroy_masked_streams()                 // atexit: registered after the HIP runtime's own handlers, so it runs before them
{
	std::lock_guard<std::mutex> lock(g_la_mutex);
	for (LookAhead* la : g_la_masked) {
		if (la->upd) (void)hipStreamDestroy(la->upd);
		if (la->diag) (void)hipStreamDestroy(la->diag);
		la->upd = la->diag = nullptr;
	}
	g_la_masked.clear();
}
static void lookahead_reserved_streams(LookAhead* la)
{
	std::lock_guard<std::mutex> lock(g_la_mutex);
	if (la->upd || la->reserved_tried) return;
	la->reserved_tried = true;
	static bool registered = false;
	if (!registered) { registered = true; (void)atexit(destroy_masked_streams); }
	// "reserved" mode (potrf, mid-size trailing matrices): the trailing update runs on a stream masked OFF one CU per XCD and
	// the diagonal-block kernel on a stream masked ONTO those eight CUs.  Mask bits are dealt round-robin over the eight XCCs
	// (tools/cumask_probe.hip: bits 0-7 = se0.cu0 of xcc0..7; an XCC whose bits are all clear gets ALL its CUs, so no XCC is
	// ever left empty).  Failure to create them only switches the mode off.
	{
		unsigned m_upd[8], m_diag[8];
		for (int w = 0; w < 8; ++w) { m_upd[w] = 0xffffffffu; m_diag[w] = 0u; }
		m_upd[0] = 0xffffff00u;
		m_diag[0] = 0x000000ffu;
		if (hipExtStreamCreateWithCUMask(&la->upd, 8, m_upd) != hipSuccess || hipExtStreamCreateWithCUMask(&la->diag, 8, m_diag) != hipSuccess ||
		    hipEventCreateWithFlags(&la->ev_diag, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&la->ev_gemm, hipEventDisableTiming) != hipSuccess ||
		    hipEventCreateWithFlags(&la->ev_mode, hipEventDisableTiming) != hipSuccess) {
			(void)hipGetLastError();
			la->upd = la->diag = nullptr;
		} else g_la_masked.push_back(la);
	}
}

#else
static void lookahead_reserved_streams(LookAhead*) {}          // (reserved-CU mode: lab build only; g_potrf_reserve_below is the constant 0 here)
#endif

template <typename T>
static bool potrf_panel_can_strip(int64_t n, int64_t k, int64_t kb, const T* A, int64_t lda, const T* winv, int64_t ldp)
{
	return kb % IB == 0 && kb <= 8 * IB && k + kb < n && n - k - kb < (1 << 30) && trsm_strip_ok(sizeof(T), A, lda, winv) && ldp % (int64_t)(16 / sizeof(T)) == 0;
}

// Factor the nb-wide panel whose first column is k (its columns already carry every update from
// the panels to its left), left-looking over 128-column blocks.  Writes L into A and into the
// panel workspace P (n x nb, leading dimension nb, rows indexed globally).
template <typename T>
static int factor_panel(int64_t n, int64_t k, int64_t kb, T* A, int64_t lda, T* winv, T* P, int64_t ldp, int32_t* info, hipStream_t st,
                        int gflags, hipEvent_t first_diag = nullptr, hipStream_t diag_st = nullptr, hipEvent_t ev_diag = nullptr, hipEvent_t ev_gemm = nullptr,
                        int strip_mode = -1)          // -1: by stpy_tune key 18; 0: per-block products over all rows; 1: strip launch on `st`; 2: rows below left to the caller
{
	// diag_st != nullptr ("reserved" mode): the 128 x 128 diagonal-block kernel runs on its own stream, which is masked onto
	// CUs the trailing update cannot use, with an event hand-over in each direction; the panel GEMMs stay on `st`.
	const bool split = diag_st != nullptr && diag_st != st;
#define FP_EV(x) do { if ((x) != hipSuccess) { set_error("potrf: event hand-over between the panel streams failed"); return -1003; } } while (0)
	int rc;
	// Rows below the panel's diagonal block: one strip launch after that block is factored (solve.hip: trsm_strip_kernel)
	// instead of two products per 128 columns over all of them -- the chain of small kernels then only spans kb rows.
	const bool can_strip = potrf_panel_can_strip<T>(n, k, kb, A, lda, winv, ldp);
	if (strip_mode < 0) strip_mode = (g_potrf_strip == 1 || g_potrf_strip == 3 || (g_potrf_strip == 2 && !(gflags & GEMM_BESIDE))) ? 1 : 0;
	if (!can_strip) strip_mode = 0;
	const bool strip = strip_mode != 0;
	const int64_t rows_end = strip ? k + kb : n;           // the per-block chain below covers rows [c, rows_end)
	for (int64_t c = k; c < k + kb; c += IB) {
		const int64_t cb = (n - c < IB) ? (n - c) : IB;
		const int64_t jj = c - k;
		if (jj > 0) {       // A[c:n, c:c+cb] -= P[c:n, 0:jj] P[c:c+cb, 0:jj]^T
			ProfScope ps(TAG_PANEL_GEMM, 2.0 * (double)(rows_end - c) * (double)cb * (double)jj, st);
			rc = gemm_nt<T>(rows_end - c, cb, jj, P + c * ldp, ldp, P + c * ldp, ldp, A + c * lda + c, lda, (T*)nullptr, 0, 1, 0, st, nullptr, nullptr, nullptr, 1, nullptr, gflags);
			if (rc) return rc;
		}
		hipStream_t ds = split ? diag_st : st;
		if (split) { FP_EV(hipEventRecord(ev_gemm, st)); FP_EV(hipStreamWaitEvent(ds, ev_gemm, 0)); }
		{
			ProfScope ps(TAG_POTF2, (double)cb * cb * cb / 3.0, ds);
			rc = potf2_trtri<T>(A + c * lda + c, lda, (int)cb, winv + (c / IB) * IB * IB, P + c * ldp + jj, ldp, info, (int)c, ds, !split && (gflags & GEMM_BESIDE) != 0);
		}
		if (rc) return rc;
		if (c == k && first_diag && hipEventRecord(first_diag, ds) != hipSuccess) { set_error("potrf: event record failed"); return -1003; }
		if (split) { FP_EV(hipEventRecord(ev_diag, ds)); FP_EV(hipStreamWaitEvent(st, ev_diag, 0)); }
		if (c + cb < rows_end) {   // A[c+cb:n, c:c+cb] <- A[..] inverse(L_cc)^T, in place + copy into the panel
			ProfScope ps(TAG_PANEL_GEMM, (double)(rows_end - c - cb) * (double)cb * (double)cb, st);   // triangular operand: half of 2mnk
			rc = gemm_nt<T>(rows_end - c - cb, cb, cb, A + (c + cb) * lda + c, lda, winv + (c / IB) * IB * IB, IB,
			                A + (c + cb) * lda + c, lda, P + (c + cb) * ldp + jj, ldp, 0, 0, st, nullptr, nullptr, nullptr, 1, nullptr, gflags);
			if (rc) return rc;
		}
	}
	if (strip_mode == 1) {
		const int64_t mb = n - k - kb;
		ProfScope ps(TAG_PANEL_GEMM, (double)mb * (double)kb * (double)kb, st);
		rc = trsm_strip<T>(mb, A + k * lda + k, lda, winv + (k / IB) * IB * IB, A + (k + kb) * lda + k, lda, P + (k + kb) * ldp, ldp, kb, st);
		if (rc) return rc;
	}
#undef FP_EV
	return 0;
}

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error("potrf: %s failed: %s", #x, hipGetErrorString(e_)); return -1000 - (int)e_; } } while (0)

// Right-looking with one panel of look-ahead.  For panel k (already factored, in workspace Pk):
//   main stream : update the NEXT panel's block column  A[r:n, r:r+nb] -= Pk Pk^T       (r = k + nb)
//                 -> event col_ready
//                 update the rest of the trailing matrix A[r+nb:n, r+nb:n] -= Pk Pk^T (lower tiles)
//   side stream : wait col_ready; factor panel k+1 into the OTHER workspace -> event panel_done
//   main stream : wait panel_done before using panel k+1.
// The two regions written concurrently are disjoint (columns [r, r+nb) vs columns >= r+nb) and the
// trailing update reads only Pk, which the side stream never touches.
template <typename T>
int potrf(int64_t n, T* A, int64_t lda, T* winv, T* work, int nb, int32_t* info, hipStream_t st, int gflags)
{
	// nb = 0: the panel width follows the size of what is LEFT (potrf_auto_nb): wide panels while the trailing
	// update is long enough to hide the next panel's latency-bound chain, narrower ones towards the end
	const bool adaptive = nb <= 0;
	if (nb <= 0) nb = potrf_auto_nb(n);
	if (nb % IB != 0) { set_error("potrf: nb must be a multiple of %d", IB); return -7; }
	HIPCHK(hipMemsetAsync(info, 0, sizeof(int32_t), st));
	const int64_t ldp = nb;                                         // widest panel (the first one)
	T* Pbuf[2] = {work, work + n * ldp};
	// fp32 (route key 32): each finished panel is split ONCE into its three bf16 planes (gemm_bf3p.hip: plane t = n x ldp bf16 behind the
	// two panel buffers) and both of its trailing updates run from those planes -- no per-tile re-split, no vector arithmetic in the K loop
	unsigned short* const planes = (unsigned short*)(work + 2 * n * ldp);
	const int64_t pstride = n * ldp;
	LookAhead* la = nullptr;
	int rc = lookahead_acquire(st, &la);
	if (rc) return rc;
	hipStream_t side = la->side;
	auto width = [&](int64_t left) { const int64_t w = adaptive ? potrf_auto_nb(left) : nb; return left < w ? left : w; };

	int64_t wk = width(n);                                          // width of the current panel
	// (stpy_tune key 24: the FIRST panel has nothing to hide behind -- its n x wk left-looking products and wk / 128 diagonal blocks
	// run alone on the chip -- so it may be narrower than the panels that follow)
	if (adaptive && g_potrf_first_nb > 0 && g_potrf_first_nb % IB == 0 && g_potrf_first_nb < wk) wk = g_potrf_first_nb;
	rc = factor_panel<T>(n, 0, wk, A, lda, winv, Pbuf[0], ldp, info, st, gflags);
	if (rc) return rc;
	int cur = 0;
	// U: the stream the trailing updates run on -- the caller's, or (reserved mode) the one masked off one CU per XCD.  A
	// change of mode orders the two streams through ev_mode; the caller's stream waits for the last update at the end.
	hipStream_t U = st;
	auto switch_to = [&](hipStream_t nu) -> int {
		if (nu == U) return 0;
		HIPCHK(hipEventRecord(la->ev_mode, U));
		HIPCHK(hipStreamWaitEvent(nu, la->ev_mode, 0));
		U = nu;
		return 0;
	};
	// trailing update C[r0:n, r0:r0+cols] -= Pk[r0:n] Pk[r0:r0+cols]^T  (lower: cols = n - r0, lower tiles only), from the planes when `pre`
	auto update = [&](T* Pk, bool pre, int64_t r0, int64_t cols, int64_t kk, int lower) -> int {
		if constexpr (sizeof(T) == 4) {
			// (few tiles: the 32 x 128 slivers of gemm_nt on the fp32 MFMA are as fast or faster.  Lab knob 33 = the tile count above which the planes
			// take a lower-triangular update, tools/potrf_sweep.py "33=128|800|1600|3200|5000": N = 8192 4.50 / 4.43 / 4.45 / 4.44 / 4.43 ms,
			// 16 384 13.87 / 13.86 / 14.08 / 14.51 / 14.67, 32 768 64.9 / 65.0 / 65.1 / 67.1 / 67.4 -> 800 tiles, i.e. ~5000 rows)
			const int64_t t = (n - r0) / 128, tiles = lower ? t * (t + 1) / 2 : t * (cols / 128);
			const int64_t floor_t = g_gemm_sliver_tiles > 0 ? g_potrf_planes_min_tiles : 0;          // (slivers switched off: the planes from 128 / 64 tiles on)
			if (pre && tiles >= (lower ? 128 : 64) && tiles > (lower ? floor_t : floor_t / 2))
				return gemm_nt_bf3p(n - r0, cols, kk, planes, ldp, pstride, r0, r0, (float*)A + r0 * lda + r0, lda, lower, U);
		}
		return gemm_nt<T>(n - r0, cols, kk, Pk + r0 * ldp, ldp, Pk + r0 * ldp, ldp, A + r0 * lda + r0, lda, (T*)nullptr, 0, 1, lower, U);
	};
	for (int64_t k = 0; k + wk < n;) {
		const int64_t r = k + wk;                                   // first row/column of the trailing matrix
		const int64_t nkb = width(n - r);                           // width of the next panel
		T* Pk = Pbuf[cur];
		// Reserved mode: beside the real update the diagonal-block kernel is PLACED quickly but RUNS four to five times
		// slower (in-kernel stamps, tools/potf2_stamps.py: 245-300 us against 58 us alone, every phase of it stretched alike),
		// and it is the longest link of a chain that mid-size trailing matrices cannot hide.  With the update kept off one CU
		// per XCD (3 % of the chip) that kernel runs alone on a reserved CU at its stand-alone speed.
		const bool want_reserve = (n - r) <= g_potrf_reserve_below && (n - r) >= g_potrf_reserve_above && !(gflags & GEMM_BESIDE);
		if (want_reserve && !la->upd) lookahead_reserved_streams(la);
		const bool reserve = want_reserve && la->upd && (n - r) <= g_potrf_reserve_below && (n - r) >= g_potrf_reserve_above && !(gflags & GEMM_BESIDE);
		rc = switch_to(reserve ? la->upd : st);
		if (rc) return rc;
		bool pre = false;
		if constexpr (sizeof(T) == 4) {
			pre = g_potrf_presplit && (n - r) % 128 == 0 && (n - r) >= 2048 && ((n - r) / 128) * ((n - r) / 128 + 1) / 2 > (g_gemm_sliver_tiles > 0 ? g_potrf_planes_min_tiles : 0) && wk % 32 == 0 && wk >= 64 && nkb % 128 == 0 && lda < (1 << 24) &&
			      (((uintptr_t)Pk | (uintptr_t)planes) & 15) == 0 && n * ldp * 2 < ((int64_t)1 << 32);
			if (pre) {
				rc = bf3_split((const float*)Pk + r * ldp, ldp, n - r, wk, planes, ldp, pstride, r, U);
				if (rc) return rc;
			}
		}
		{   // next panel's block column (all rows below r)
			ProfScope ps(TAG_SYRK, 2.0 * (double)(n - r) * (double)nkb * (double)wk - (double)nkb * (double)nkb * (double)wk, U);
			rc = update(Pk, pre, r, nkb, wk, 0);
			if (rc) return rc;
		}
		// key 18 = 3: the look-ahead stream factors only the panel's nkb x nkb diagonal block (the latency-bound chain: diagonal-block
		// kernels + small products over <= nkb rows); the rows below it are ONE strip launch on the update's stream once the trailing
		// update has drained -- alone on the chip, at its stand-alone speed, instead of 2 x nkb/128 sliver products beside the update.
		// Together with the reserved mode the WHOLE chain runs on the stream masked onto the reserved CUs (no per-block hand-over).
		const bool defer = g_potrf_strip == 3 && potrf_panel_can_strip<T>(n, r, nkb, A, lda, winv, ldp);
		// key 21: below this many remaining rows no look-ahead at all -- the whole trailing update first, then the panel on the same
		// stream, each alone on the chip (beside the update the chain runs 3-5x slower, and a short update cannot hide it anyway)
		// (lab knob 31, off: serial order also in a band ABOVE the sliver threshold of the trailing update, where the update is the 128 x 128
		// direct-to-VGPR kernel and the diagonal-block kernel beside it runs 8x slower than alone -- 265 us against 33, kernel trace of
		// tools/potrf_only.py 16384 -- because every SIMD's fp64 ALUs are held by 64-cycle MFMAs.  Measured: it LOSES, monotonically with the band's
		// upper edge (N = 16 384: 27.96 ms off, 28.11 / 28.67 / 29.52 / 30.09 with the edge at 11 000 / 12 000 / 14 000 / 16 500 rows; N = 32 768 179.5 ->
		// 184.0 at 20 000): up there the update and the chain are about balanced (17 ms of update against 19 ms of chain over the first 6000 columns of
		// N = 16 384) and the serial order gives up the update's own ramp and tail, which the chain otherwise fills.)
		const int64_t rem_tiles = ((n - r) / IB) * ((n - r) / IB + 1) / 2;
		const bool serial = !reserve && ((n - r) <= g_potrf_serial_below || (sizeof(T) == 8 && rem_tiles > g_gemm_sliver_tiles && (n - r) <= g_potrf_serial_band));
		if (serial) {
			if (r + nkb < n) {
				const int64_t r2 = r + nkb;
				ProfScope ps(TAG_SYRK, (double)(n - r2) * (double)(n - r2) * (double)wk, U);
				rc = update(Pk, pre, r2, n - r2, wk, 1);
				if (rc) return rc;
			}
			rc = factor_panel<T>(n, r, nkb, A, lda, winv, Pbuf[cur ^ 1], ldp, info, U, gflags);
			if (rc) return rc;
			cur ^= 1;
			k = r;
			wk = nkb;
			continue;
		}
		hipStream_t chain = (reserve && defer) ? la->diag : side;
		HIPCHK(hipEventRecord(la->col_ready, U));
		HIPCHK(hipStreamWaitEvent(chain, la->col_ready, 0));
		// A short trailing update cannot hide the panel chain, and the chain's first kernel -- one workgroup that needs
		// 83 KiB of LDS -- is slowed down by the update's workgroups once they have flooded the chip (kernel trace,
		// tools/potrf_only.py).  Below the threshold the update therefore starts only after that kernel has run.
		const bool diag_first = !reserve && (n - r) <= g_potrf_diag_first_below;
		const int pflags = gflags | ((n - r - nkb) >= g_potrf_beside_min ? GEMM_BESIDE : 0);
		rc = factor_panel<T>(n, r, nkb, A, lda, winv, Pbuf[cur ^ 1], ldp, info, chain, (reserve && defer) ? gflags : pflags, diag_first ? la->trail_done : nullptr,
		                     (reserve && !defer) ? la->diag : nullptr, la->ev_diag, la->ev_gemm, defer ? 2 : -1);
		if (rc) return rc;
		HIPCHK(hipEventRecord(la->panel_done, chain));
		if (diag_first) HIPCHK(hipStreamWaitEvent(U, la->trail_done, 0));
		if (r + nkb < n) {  // rest of the trailing matrix, lower tiles only
			const int64_t r2 = r + nkb;
			ProfScope ps(TAG_SYRK, (double)(n - r2) * (double)(n - r2) * (double)wk, U);      // lower triangle: m^2 k
			rc = update(Pk, pre, r2, n - r2, wk, 1);
			if (rc) return rc;
		}
		HIPCHK(hipStreamWaitEvent(U, la->panel_done, 0));
		if (defer) {
			const int64_t mb = n - r - nkb;
			ProfScope ps(TAG_PANEL_GEMM, (double)mb * (double)nkb * (double)nkb, U);
			rc = trsm_strip<T>(mb, A + r * lda + r, lda, winv + (r / IB) * IB * IB, A + (r + nkb) * lda + r, lda, Pbuf[cur ^ 1] + (r + nkb) * ldp, ldp, nkb, U);
			if (rc) return rc;
		}
		// the side stream may not start overwriting workspace `cur` (panel k+2) before this
		// trailing update has finished reading it: it waits on the next col_ready, which is
		// recorded on `U` after this update -- stream order (and ev_mode across a change of mode) gives that.
		cur ^= 1;
		k = r;
		wk = nkb;
	}
	return switch_to(st);
}

template int potf2_trtri<double>(double*, int64_t, int, double*, double*, int64_t, int32_t*, int, hipStream_t, bool);
template int potf2_trtri<float>(float*, int64_t, int, float*, float*, int64_t, int32_t*, int, hipStream_t, bool);
template int potrf<double>(int64_t, double*, int64_t, double*, double*, int, int32_t*, hipStream_t, int);
template int potrf<float>(int64_t, float*, int64_t, float*, float*, int, int32_t*, hipStream_t, int);

}  // namespace stpy
