// The serial part of the diagonal-block kernel (potrf.hip: potf2_trtri_mfma_kernel::diag_block): Cholesky factor + inverse of a
// 16 x 16 fp64 block in ONE wave, lane (q, i) = row i, columns 4q .. 4q+3.  Variant 0: the shipped form (nine LDS lane permutes of
// a double per pivot).  Variant 1: the pivot row of the inverse through DPP row broadcasts, the column multiplier through
// v_permlane16_swap / v_permlane32_swap row broadcasts, the next pivot taken from its own one-FMA update (no permute on the chain).
// Variants 2 / 3: the same with the pivot made opaque to hipcc's uniformity analysis.  Variant 4: factor and inverse on TWO waves.
// Prints cycles per block and the largest error against a host factorisation.  (Measured: 8467 / 9031 / 8788 / 9183 / 9100-9500 with the factor wave alone at 8500-8800; variants 5-7: one Newton step 8207, fp32 estimate 8413, both 8260; 8: lane masks recomputed per pivot instead of hoisted + spilled 10741;
// 9 / 10: the pivot broadcast kept in vector registers and no select on a failing pivot 9564 / 10038.  11: lane conditions folded into the data (778 instead of 1300 vector instructions, 114 instead of 420 selects, nothing spilled): 8238-8442;
// 12: 11 + no LDS round trip on the pivot chain: 8098.  An in-order wave stalls at the wait for the l_kj gathers of pivot J before it can
// issue pivot J+1's chain, whatever is on the data path -- only a hand-pipelined order (next column updated through a scalar broadcast,
// bulk update one pivot behind) would change that -- variant 13 is exactly that order, correct, and 8263: it does not.  All thirteen forms
// land within 8000-10700.  Ablations (wrong numbers, timing only): no pivots at all 384; no rsqrt / Newton 7988; only the next pivot's column
// updated and no inverse 4136; both 3461; the shipped form with only the inverse's pivot row through DPP 7964.  So about half of a pivot is
// its chain (one LDS round trip) and half the bulk update.
// Counters of single variants (`diag16_probe <variant>` under rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
// SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU), per block: the one wave executes strictly one instruction at a
// time -- 4.3 cycles per vector instruction, 8 per LDS permute -- plus the waits:   variant 0: 983 VALU = 4126 cycles, 279 LDS = 2216,
// SALU 375, wait 1700 (sum 8484);   11 (lean): 604 / 2610, 279 / 2216, wait 3166 (fewer independent instructions to fill the permute latency);
// 12: 1003 / 4454, 128 / 1008, wait 2054;   18: 1234 / 5126, 154 / 1216, wait 1196;   19 (order pinned by hand: the bulk update of pivot
// J - 1 dealt out between the dependent instructions of pivot J's chain): 1079 / 4758, 128 / 1008, wait 1537 = 7777, the best form, -8 %.)
// build: hipcc --offload-arch=gfx950 -O3 -o diag16_probe tools/diag16_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double T;
__device__ __forceinline__ double bcast(double v, int src)
{
	const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
	const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
	return __hiloint2double(hi, lo);
}
// all four 16-lane rows <- row R of x
template <int R>
__device__ __forceinline__ unsigned row_bcast_u(unsigned x)
{
	auto t = __builtin_amdgcn_permlane16_swap(x, x, false, false);
	const unsigned h = t[R & 1];
	auto u = __builtin_amdgcn_permlane32_swap(h, h, false, false);
	return u[R >> 1];
}
template <int R>
__device__ __forceinline__ double row_bcast(double v)
{
	return __hiloint2double((int)row_bcast_u<R>((unsigned)__double2hiint(v)), (int)row_bcast_u<R>((unsigned)__double2loint(v)));
}
// every lane <- lane J of its own 16-lane row (DPP row_newbcast)
template <int J>
__device__ __forceinline__ double lane_bcast(double v)
{
	const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + J, 0xf, 0xf, false);
	const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + J, 0xf, 0xf, false);
	return __hiloint2double(hi, lo);
}

template <int VAR, int J>
__device__ __forceinline__ void pivot(T (&a)[4], T (&w)[4], T& dnext, int q, int i, int lane, int& first_bad)
{
	constexpr int qj = J >> 2, cj = J & 3;
	// bit 4: the lane coordinates are made opaque per pivot, so that hipcc cannot hoist the 16 x 7 lane masks out of the block loop
	// (they do not fit into the SGPR file and come back through v_readlane spills: .sgpr_spill_count 142 in the shipped kernel)
	if (VAR & 16) asm volatile("" : "+v"(q), "+v"(i));
	T d;
	if (VAR & 32) {          // bit 5: the pivot never leaves the vector registers (DPP lane broadcast inside its row, then the row to all rows) ...
		const T src = ((VAR & 1) && J > 0) ? dnext : a[cj];
		d = row_bcast<qj>(lane_bcast<J>(src));
	} else
		d = ((VAR & 1) && J > 0) ? bcast(dnext, 16 * qj + J) : bcast(a[cj], 16 * qj + J);
	if (VAR & 2) asm volatile("" : "+v"(d));          // opaque: hipcc otherwise keeps the whole uniform chain (rsqrt, Newton steps) in SGPRs, one v_readfirstlane pair per VALU result
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	if (!(VAR & 32)) d = bad ? T(1) : d;          // ... and is not replaced when it fails (compare -> lane mask -> select is a second trip through the SGPRs on the chain): NaN from there on
	T rl = (VAR & 8) ? (T)__builtin_amdgcn_rsqf((float)d) : (T)__builtin_amdgcn_rsq(d);          // bit 3: fp32 estimate
	if (VAR & 128) rl = d * T(0.5);          // bit 7 (timing ablation, wrong numbers): no reciprocal square root, no Newton steps
	else {
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	if (!(VAR & 4)) rl = rl * (T(1.5) - T(0.5) * d * rl * rl);          // bit 2: ONE Newton step
	}
	const T l = d * rl;
	const T colv = (i == J) ? l : a[cj] * rl;
	a[cj] = (q == qj && i >= J) ? colv : a[cj];
	T mi, lk[4], wj[4];
	if (!(VAR & 1)) {
		mi = __shfl(colv, 16 * qj + i, 64);
#pragma unroll
		for (int c = 0; c < 4; ++c) { lk[c] = __shfl(colv, 16 * qj + 4 * q + c, 64); wj[c] = (VAR & 512) ? lane_bcast<J>(w[c]) : __shfl(w[c], 16 * q + J, 64); }          // bit 9: ONLY the pivot row of the inverse through DPP
	} else {
		mi = row_bcast<qj>(colv);
		if (J + 1 < 16) {          // the next pivot's diagonal element, by the same FMA the update below applies to it (l_kj = l_ij on the diagonal)
			constexpr int cn = (J + 1) & 3;
			dnext = a[cn] - mi * mi;
		}
#pragma unroll
		for (int c = 0; c < 4; ++c) { lk[c] = __shfl(mi, 16 * q + 4 * q + c, 64); wj[c] = lane_bcast<J>(w[c]); }
	}
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const int k = 4 * q + c;
		const T na = a[c] - mi * lk[c];
		if (!(VAR & 256) || c == ((J + 1) & 3)) a[c] = (k > J && i >= k) ? na : a[c];          // bit 8 (timing ablation): only the next pivot's column is updated
		const T ws = wj[c] * rl;
		const T nw = w[c] - mi * ws;
		if (!(VAR & 256)) w[c] = (i == J) ? ws : ((i > J) ? nw : w[c]);
	}
}

// ---- "lean" pivot (variant 11): the same arithmetic with the lane conditions folded into the DATA instead of 26 selects per pivot.
//   * multiplier column zeroed on the rows at or above the pivot (ONE select, mask i > J): rows that are finished get a zero multiplier,
//     columns at or left of the pivot get a zero l_kj -- the (k > J && i >= k) guards of the update become unnecessary;
//   * the strict upper triangle of `a` is allowed to fill with garbage: it is only ever read through that zeroed multiplier column
//     and never stored;
//   * the inverse: w <- w * f - mi0 * (w_J * rl) with f = rl on the pivot row and 1 elsewhere (ONE select).
// Masks needed: (i > J), (i == J) per pivot and (q == qj): 36 SGPR pairs in all -- nothing spills.
template <int J>
__device__ __forceinline__ void pivot_lean(T (&a)[4], T (&w)[4], int q, int i, int& first_bad)
{
	constexpr int qj = J >> 2, cj = J & 3;
	T d = bcast(a[cj], 16 * qj + J);
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	d = bad ? T(1) : d;
	T rl = (T)__builtin_amdgcn_rsq(d);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	const T colv = a[cj] * rl;                        // on the pivot lane: d * rl = l_jj
	a[cj] = (q == qj) ? colv : a[cj];
	const T colv0 = (i > J) ? colv : T(0);
	const T mi0 = __shfl(colv0, 16 * qj + i, 64);
	const T f = (i == J) ? rl : T(1);
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const T lk0 = __shfl(colv0, 16 * qj + 4 * q + c, 64);
		const T wj = __shfl(w[c], 16 * q + J, 64);
		a[c] = a[c] - mi0 * lk0;
		w[c] = w[c] * f - mi0 * (wj * rl);
	}
}
// variant 12: lean + no LDS round trip on the pivot chain: the multiplier column reaches the other rows through the two row-swap
// instructions, the next pivot comes from its own FMA, the pivot row of the inverse through DPP; only the four l_kj gathers still use LDS
template <int J>
__device__ __forceinline__ void pivot_lean2(T (&a)[4], T (&w)[4], T& dnext, int q, int i, int& first_bad)
{
	constexpr int qj = J >> 2, cj = J & 3;
	T d = (J > 0) ? bcast(dnext, 16 * qj + J) : bcast(a[cj], 16 * qj + J);
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	d = bad ? T(1) : d;
	T rl = (T)__builtin_amdgcn_rsq(d);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	const T colv = a[cj] * rl;
	a[cj] = (q == qj) ? colv : a[cj];
	const T colv0 = (i > J) ? colv : T(0);
	const T mi0 = row_bcast<qj>(colv0);
	if constexpr (J + 1 < 16) { constexpr int cn = (J + 1) & 3; dnext = a[cn] - mi0 * mi0; }
	const T f = (i == J) ? rl : T(1);
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const T lk0 = __shfl(mi0, 16 * q + 4 * q + c, 64);
		const T wj = lane_bcast<J>(w[c]);
		a[c] = a[c] - mi0 * lk0;
		w[c] = w[c] * f - mi0 * (wj * rl);
	}
}
// ---- variant 13: the lean pivot, HAND-PIPELINED: the bulk update of pivot J - 1 (which waits for its four l_kj gathers) is issued
// AFTER pivot J's chain, and pivot J + 1's column is prepared from a scalar broadcast of l_{J+1,J}; the in-order wave never waits for an
// LDS gather issued in the same iteration.
struct PipeState { T mi0, rl, f, colS, lk0[4]; };
template <int J>
__device__ __forceinline__ void pivot_pipe(T (&a)[4], T (&w)[4], T& colJ, PipeState& prev, int q, int i, int& first_bad)
{
	constexpr int qj = J >> 2, cj = J & 3;
	// 1. chain of pivot J on its prepared column
	T d = bcast(colJ, 16 * qj + J);
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	d = bad ? T(1) : d;
	T rl = (T)__builtin_amdgcn_rsq(d);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	const T colv = colJ * rl;
	const T colv0 = (i > J) ? colv : T(0);
	const T mi0 = row_bcast<qj>(colv0);
	PipeState cur;
	cur.mi0 = mi0; cur.rl = rl; cur.f = (i == J) ? rl : T(1); cur.colS = colv;
	// 2. gathers of pivot J (consumed one iteration later)
#pragma unroll
	for (int c = 0; c < 4; ++c) cur.lk0[c] = __shfl(mi0, 16 * q + 4 * q + c, 64);
	// 3. bulk update of pivot J - 1, then the scaled column J - 1 goes into its place
	if constexpr (J > 0) {
		constexpr int qp = (J - 1) >> 2, cp = (J - 1) & 3;
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			const T wj = lane_bcast<J - 1>(w[c]);
			a[c] = a[c] - prev.mi0 * prev.lk0[c];
			w[c] = w[c] * prev.f - prev.mi0 * (wj * prev.rl);
		}
		a[cp] = (q == qp) ? prev.colS : a[cp];
	}
	// 4. column J + 1, brought up to date with pivot J through a scalar broadcast of l_{J+1,J}
	if constexpr (J + 1 < 16) {
		constexpr int cn = (J + 1) & 3;
		const T s = bcast(colv0, 16 * qj + J + 1);
		colJ = a[cn] - mi0 * s;
	}
	prev = cur;
}
template <int J>
__device__ __forceinline__ void pipe_run(T (&a)[4], T (&w)[4], T& colJ, PipeState& prev, int q, int i, int& fb)
{
	pivot_pipe<J>(a, w, colJ, prev, q, i, fb);
	if constexpr (J + 1 < 16) pipe_run<J + 1>(a, w, colJ, prev, q, i, fb);
}
__global__ __launch_bounds__(64) void diag_pipe_kernel(const T* __restrict__ in, T* __restrict__ outL, T* __restrict__ outW, int nblk, int reps, long long* cycles)
{
	extern __shared__ T S[];
	const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
	for (int t = lane; t < nblk * 256; t += 64) S[t] = in[t];
	__syncthreads();
	long long t0 = 0;
	for (int r = 0; r <= reps; ++r) {
		if (r == 1) t0 = __builtin_amdgcn_s_memtime();
		for (int b = 0; b < nblk; ++b) {
			T a[4], w[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				a[c] = (col <= i) ? S[b * 256 + i * 16 + col] : T(0);
				w[c] = (col == i) ? T(1) : T(0);
			}
			int first_bad = 0;
			T colJ = a[0];
			PipeState prev;
			pipe_run<0>(a, w, colJ, prev, q, i, first_bad);
			// drain: bulk update of pivot 15 (only its scaling of row 15 of the inverse and the scaled column matter)
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const T wj = lane_bcast<15>(w[c]);
				a[c] = a[c] - prev.mi0 * prev.lk0[c];
				w[c] = w[c] * prev.f - prev.mi0 * (wj * prev.rl);
			}
			a[3] = (q == 3) ? prev.colS : a[3];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				outL[b * 256 + i * 16 + col] = (col <= i) ? a[c] : T(0);
				outW[b * 256 + i * 16 + col] = (col <= i) ? w[c] : T(0);
			}
			if (first_bad) outL[0] = -1;
		}
	}
	if (lane == 0) *cycles = (long long)__builtin_amdgcn_s_memtime() - t0;
}

// ---- variant 19: variant 13 with the ORDER pinned by hand (sched_barrier): the bulk update of pivot J - 1 (26 independent vector
// instructions) is dealt out between the mutually dependent instructions of pivot J's chain (broadcast, rsqrt, two Newton steps, scale,
// row swaps), so that the one wave issues something every slot instead of stalling on each result; the inverse's pivot row through DPP.
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int J>
__device__ __forceinline__ void pivot_pipe2(T (&a)[4], T (&w)[4], T& colJ, PipeState& prev, int q, int i, int& first_bad)
{
	constexpr int qj = J >> 2, cj = J & 3;
	constexpr bool HAVE = J > 0;
	constexpr int JP = HAVE ? J - 1 : 0, qp = JP >> 2, cp = JP & 3;
	T wj[4], t[4];
	// chain: the pivot ...
	T d = bcast(colJ, 16 * qj + J);
	if (HAVE) { a[0] = a[0] - prev.mi0 * prev.lk0[0]; a[1] = a[1] - prev.mi0 * prev.lk0[1]; }
	SB();
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	d = bad ? T(1) : d;
	if (HAVE) { a[2] = a[2] - prev.mi0 * prev.lk0[2]; a[3] = a[3] - prev.mi0 * prev.lk0[3]; }
	SB();
	T rl = (T)__builtin_amdgcn_rsq(d);
	if (HAVE) { wj[0] = lane_bcast<JP>(w[0]); wj[1] = lane_bcast<JP>(w[1]); }
	SB();
	T u = d * rl;
	if (HAVE) { wj[2] = lane_bcast<JP>(w[2]); }
	SB();
	u = u * rl;
	if (HAVE) { wj[3] = lane_bcast<JP>(w[3]); }
	SB();
	u = T(1.5) - T(0.5) * u;
	if (HAVE) { t[0] = wj[0] * prev.rl; }
	SB();
	rl = rl * u;
	if (HAVE) { t[1] = wj[1] * prev.rl; }
	SB();
	u = d * rl;
	if (HAVE) { t[2] = wj[2] * prev.rl; }
	SB();
	u = u * rl;
	if (HAVE) { t[3] = wj[3] * prev.rl; }
	SB();
	u = T(1.5) - T(0.5) * u;
	if (HAVE) { w[0] = w[0] * prev.f; }
	SB();
	rl = rl * u;
	if (HAVE) { w[1] = w[1] * prev.f; }
	SB();
	const T colv = colJ * rl;
	if (HAVE) { w[2] = w[2] * prev.f; }
	SB();
	const T colv0 = (i > J) ? colv : T(0);
	if (HAVE) { w[3] = w[3] * prev.f; }
	SB();
	const T mi0 = row_bcast<qj>(colv0);
	if (HAVE) { w[0] = w[0] - prev.mi0 * t[0]; w[1] = w[1] - prev.mi0 * t[1]; }
	SB();
	PipeState cur;
	cur.mi0 = mi0; cur.rl = rl; cur.f = (i == J) ? rl : T(1); cur.colS = colv;
#pragma unroll
	for (int c = 0; c < 4; ++c) cur.lk0[c] = __shfl(mi0, 16 * q + 4 * q + c, 64);
	if (HAVE) { w[2] = w[2] - prev.mi0 * t[2]; w[3] = w[3] - prev.mi0 * t[3]; a[cp] = (q == qp) ? prev.colS : a[cp]; }
	SB();
	if constexpr (J + 1 < 16) {
		constexpr int cn = (J + 1) & 3;
		const T sN = bcast(colv0, 16 * qj + J + 1);
		colJ = a[cn] - mi0 * sN;
	}
	prev = cur;
}
template <int J>
__device__ __forceinline__ void pipe2_run(T (&a)[4], T (&w)[4], T& colJ, PipeState& prev, int q, int i, int& fb)
{
	pivot_pipe2<J>(a, w, colJ, prev, q, i, fb);
	if constexpr (J + 1 < 16) pipe2_run<J + 1>(a, w, colJ, prev, q, i, fb);
}
__global__ __launch_bounds__(64) void diag_pipe2_kernel(const T* __restrict__ in, T* __restrict__ outL, T* __restrict__ outW, int nblk, int reps, long long* cycles)
{
	extern __shared__ T S[];
	const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
	for (int t = lane; t < nblk * 256; t += 64) S[t] = in[t];
	__syncthreads();
	long long t0 = 0;
	for (int r = 0; r <= reps; ++r) {
		if (r == 1) t0 = __builtin_amdgcn_s_memtime();
		for (int b = 0; b < nblk; ++b) {
			T a[4], w[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				a[c] = (col <= i) ? S[b * 256 + i * 16 + col] : T(0);
				w[c] = (col == i) ? T(1) : T(0);
			}
			int first_bad = 0;
			T colJ = a[0];
			PipeState prev;
			pipe2_run<0>(a, w, colJ, prev, q, i, first_bad);
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const T wjl = lane_bcast<15>(w[c]);
				a[c] = a[c] - prev.mi0 * prev.lk0[c];
				w[c] = w[c] * prev.f - prev.mi0 * (wjl * prev.rl);
			}
			a[3] = (q == 3) ? prev.colS : a[3];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				outL[b * 256 + i * 16 + col] = (col <= i) ? a[c] : T(0);
				outW[b * 256 + i * 16 + col] = (col <= i) ? w[c] : T(0);
			}
			if (first_bad) outL[0] = -1;
		}
	}
	if (lane == 0) *cycles = (long long)__builtin_amdgcn_s_memtime() - t0;
}

template <int VARIANT, int J>
__device__ __forceinline__ void lean_run(T (&a)[4], T (&w)[4], T& dnext, int q, int i, int& fb)
{
	if constexpr (VARIANT == 0) pivot_lean<J>(a, w, q, i, fb); else pivot_lean2<J>(a, w, dnext, q, i, fb);
	if constexpr (J + 1 < 16) lean_run<VARIANT, J + 1>(a, w, dnext, q, i, fb);
}
template <int VARIANT>
__global__ __launch_bounds__(64) void diag_lean_kernel(const T* __restrict__ in, T* __restrict__ outL, T* __restrict__ outW, int nblk, int reps, long long* cycles)
{
	extern __shared__ T S[];
	const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
	for (int t = lane; t < nblk * 256; t += 64) S[t] = in[t];
	__syncthreads();
	long long t0 = 0;
	for (int r = 0; r <= reps; ++r) {
		if (r == 1) t0 = __builtin_amdgcn_s_memtime();
		for (int b = 0; b < nblk; ++b) {
			T a[4], w[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				a[c] = (col <= i) ? S[b * 256 + i * 16 + col] : T(0);
				w[c] = (col == i) ? T(1) : T(0);
			}
			int first_bad = 0;
			T dnext = 0;
			lean_run<VARIANT, 0>(a, w, dnext, q, i, first_bad);
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				outL[b * 256 + i * 16 + col] = (col <= i) ? a[c] : T(0);
				outW[b * 256 + i * 16 + col] = (col <= i) ? w[c] : T(0);
			}
			if (first_bad) outL[0] = -1;
		}
	}
	if (lane == 0) *cycles = (long long)__builtin_amdgcn_s_memtime() - t0;
}

template <int VAR>
__global__ __launch_bounds__(64) void diag_kernel(const T* __restrict__ in, T* __restrict__ outL, T* __restrict__ outW, int nblk, int reps, long long* cycles)
{
	extern __shared__ T S[];          // nblk blocks of 16 x 16
	const int lane = threadIdx.x, q = lane >> 4, i = lane & 15;
	for (int t = lane; t < nblk * 256; t += 64) S[t] = in[t];
	__syncthreads();
	long long t0 = 0;
	for (int r = 0; r <= reps; ++r) {
		if (r == 1) t0 = __builtin_amdgcn_s_memtime();          // repetition 0 warms the instruction cache
		for (int b = 0; b < nblk; ++b) {
			T a[4], w[4], dnext = 0;
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				a[c] = (col <= i) ? S[b * 256 + i * 16 + col] : T(0);
				w[c] = (col == i) ? T(1) : T(0);
			}
			int first_bad = 0;
			if constexpr (VAR != 64) {          // (64: no pivots at all -- the cost of the block loop around them)
			pivot<VAR, 0>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 1>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 2>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 3>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 4>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 5>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 6>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 7>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 8>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 9>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 10>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 11>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 12>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 13>(a, w, dnext, q, i, lane, first_bad);
			pivot<VAR, 14>(a, w, dnext, q, i, lane, first_bad); pivot<VAR, 15>(a, w, dnext, q, i, lane, first_bad);
			}
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				const int col = 4 * q + c;
				outL[b * 256 + i * 16 + col] = (col <= i) ? a[c] : T(0);
				outW[b * 256 + i * 16 + col] = (col <= i) ? w[c] : T(0);
			}
			if (first_bad) outL[0] = -1;
		}
	}
	if (lane == 0) *cycles = (long long)__builtin_amdgcn_s_memtime() - t0;
}

// ---- variant 4: TWO waves.  Wave 0 factors (the a-half of the pivot step), wave 1 builds the inverse (the w-half) one pivot behind:
// per pivot wave 0 publishes the scaled column (the 16 values of row group qj) and 1/l_jj in LDS and then bumps a counter; wave 1
// polls the counter (no barrier: both waves are resident in the same workgroup, and the LDS serves a wave's operations in order).
struct Comm { double colv[16][16]; double rl[16]; int flag; };
template <int J>
__device__ __forceinline__ void pivot_factor(T (&a)[4], int q, int i, int lane, int& first_bad, Comm* cm, int seq0)
{
	constexpr int qj = J >> 2, cj = J & 3;
	T d = bcast(a[cj], 16 * qj + J);
	const bool bad = !(d > T(0)) || !(d < T(1e300));
	first_bad = (bad && first_bad == 0) ? J + 1 : first_bad;
	d = bad ? T(1) : d;
	T rl = (T)__builtin_amdgcn_rsq(d);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	rl = rl * (T(1.5) - T(0.5) * d * rl * rl);
	const T l = d * rl;
	const T colv = (i == J) ? l : a[cj] * rl;
	a[cj] = (q == qj && i >= J) ? colv : a[cj];
	if (q == qj) cm->colv[J][i] = colv;
	if (lane == 0) cm->rl[J] = rl;
	// (the LDS serves one wave's operations in order, so the data stores are performed before the counter store without waiting for
	// them: a compiler-level fence is all that is needed -- a workgroup-scope release fence costs an s_waitcnt on the pivot chain)
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	if (lane == 0) __hip_atomic_store(&cm->flag, seq0 + J + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	const T mi = __shfl(colv, 16 * qj + i, 64);
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const T lk = __shfl(colv, 16 * qj + 4 * q + c, 64);
		const int k = 4 * q + c;
		const T na = a[c] - mi * lk;
		a[c] = (k > J && i >= k) ? na : a[c];
	}
}
template <int J>
__device__ __forceinline__ void pivot_inverse(T (&w)[4], int q, int i, int lane, Comm* cm, int seq0)
{
	while (__hip_atomic_load(&cm->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < seq0 + J + 1) __builtin_amdgcn_s_sleep(1);
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
	const T mi = cm->colv[J][i];
	const T rl = cm->rl[J];
#pragma unroll
	for (int c = 0; c < 4; ++c) {
		const T wj = __shfl(w[c], 16 * q + J, 64);
		const T ws = wj * rl;
		const T nw = w[c] - mi * ws;
		w[c] = (i == J) ? ws : ((i > J) ? nw : w[c]);
	}
}
__global__ __launch_bounds__(128) void diag2_kernel(const T* __restrict__ in, T* __restrict__ outL, T* __restrict__ outW, int nblk, int reps, long long* cycles)
{
	extern __shared__ T S[];          // nblk blocks of 16 x 16, then the hand-over area
	Comm* cm = reinterpret_cast<Comm*>(S + nblk * 256);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, i = lane & 15;
	for (int t = tid; t < nblk * 256; t += 128) S[t] = in[t];
	if (tid == 0) cm->flag = 0;
	__syncthreads();
	long long t0 = 0, busy = 0;
	int seq = 0;
	for (int r = 0; r <= reps; ++r) {
		if (r == 1) { t0 = __builtin_amdgcn_s_memtime(); busy = 0; }
		for (int b = 0; b < nblk; ++b, seq += 16) {
			const long long tb = __builtin_amdgcn_s_memtime();
			if (wave == 0) {
				T a[4];
#pragma unroll
				for (int c = 0; c < 4; ++c) { const int col = 4 * q + c; a[c] = (col <= i) ? S[b * 256 + i * 16 + col] : T(0); }
				int first_bad = 0;
				pivot_factor<0>(a, q, i, lane, first_bad, cm, seq); pivot_factor<1>(a, q, i, lane, first_bad, cm, seq); pivot_factor<2>(a, q, i, lane, first_bad, cm, seq); pivot_factor<3>(a, q, i, lane, first_bad, cm, seq);
				pivot_factor<4>(a, q, i, lane, first_bad, cm, seq); pivot_factor<5>(a, q, i, lane, first_bad, cm, seq); pivot_factor<6>(a, q, i, lane, first_bad, cm, seq); pivot_factor<7>(a, q, i, lane, first_bad, cm, seq);
				pivot_factor<8>(a, q, i, lane, first_bad, cm, seq); pivot_factor<9>(a, q, i, lane, first_bad, cm, seq); pivot_factor<10>(a, q, i, lane, first_bad, cm, seq); pivot_factor<11>(a, q, i, lane, first_bad, cm, seq);
				pivot_factor<12>(a, q, i, lane, first_bad, cm, seq); pivot_factor<13>(a, q, i, lane, first_bad, cm, seq); pivot_factor<14>(a, q, i, lane, first_bad, cm, seq); pivot_factor<15>(a, q, i, lane, first_bad, cm, seq);
#pragma unroll
				for (int c = 0; c < 4; ++c) { const int col = 4 * q + c; outL[b * 256 + i * 16 + col] = (col <= i) ? a[c] : T(0); }
				if (first_bad) outL[0] = -1;
			} else {
				T w[4];
#pragma unroll
				for (int c = 0; c < 4; ++c) w[c] = (4 * q + c == i) ? T(1) : T(0);
				pivot_inverse<0>(w, q, i, lane, cm, seq); pivot_inverse<1>(w, q, i, lane, cm, seq); pivot_inverse<2>(w, q, i, lane, cm, seq); pivot_inverse<3>(w, q, i, lane, cm, seq);
				pivot_inverse<4>(w, q, i, lane, cm, seq); pivot_inverse<5>(w, q, i, lane, cm, seq); pivot_inverse<6>(w, q, i, lane, cm, seq); pivot_inverse<7>(w, q, i, lane, cm, seq);
				pivot_inverse<8>(w, q, i, lane, cm, seq); pivot_inverse<9>(w, q, i, lane, cm, seq); pivot_inverse<10>(w, q, i, lane, cm, seq); pivot_inverse<11>(w, q, i, lane, cm, seq);
				pivot_inverse<12>(w, q, i, lane, cm, seq); pivot_inverse<13>(w, q, i, lane, cm, seq); pivot_inverse<14>(w, q, i, lane, cm, seq); pivot_inverse<15>(w, q, i, lane, cm, seq);
#pragma unroll
				for (int c = 0; c < 4; ++c) { const int col = 4 * q + c; outW[b * 256 + i * 16 + col] = (col <= i) ? w[c] : T(0); }
			}
			busy += (long long)__builtin_amdgcn_s_memtime() - tb;
			__syncthreads();          // (the real kernel has a workgroup barrier here as well: both halves are done before the panel below uses them)
		}
	}
	if (tid == 0) *cycles = (long long)__builtin_amdgcn_s_memtime() - t0;
	if (lane == 0) cycles[1 + wave] = busy;          // per wave: time from the start of a block to the end of its own half
}

__global__ void perm_check(unsigned* o)
{
	const unsigned x = threadIdx.x;
	o[threadIdx.x] = row_bcast_u<0>(x);
	o[64 + threadIdx.x] = row_bcast_u<1>(x);
	o[128 + threadIdx.x] = row_bcast_u<2>(x);
	o[192 + threadIdx.x] = row_bcast_u<3>(x);
	o[256 + threadIdx.x] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x150 + 5, 0xf, 0xf, false);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
	const int only = argc > 1 ? atoi(argv[1]) : -1;          // one variant only (target for rocprofv3 --pmc passes)
	unsigned* po; CK(hipMalloc(&po, 320 * 4));
	hipLaunchKernelGGL(perm_check, dim3(1), dim3(64), 0, 0, po);
	std::vector<unsigned> ph(320); CK(hipMemcpy(ph.data(), po, 320 * 4, hipMemcpyDeviceToHost));
	bool ok = true;
	for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) ok = ok && ph[r * 64 + l] == (unsigned)(16 * r + (l & 15));
	for (int l = 0; l < 64; ++l) ok = ok && ph[256 + l] == (unsigned)((l & ~15) + 5);
	printf("row_bcast (permlane16_swap + permlane32_swap) and row_newbcast DPP: %s\n", ok ? "as assumed" : "NOT as assumed");
	if (!ok) { for (int r = 0; r < 5; ++r) { for (int l = 0; l < 64; ++l) printf("%u ", ph[r * 64 + l]); printf("\n"); } }

	const int nblk = 8, reps = 200;
	std::vector<double> A(nblk * 256), Lr(nblk * 256), Wr(nblk * 256);
	srand(5);
	for (int b = 0; b < nblk; ++b) {
		double G[16][16];
		for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) G[i][j] = (rand() / (double)RAND_MAX) - 0.5;
		for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 0.5 : 0.0; for (int k = 0; k < 16; ++k) s += G[i][k] * G[j][k]; A[b * 256 + i * 16 + j] = s; }
		// host Cholesky + inverse
		double Lh[16][16] = {}, Wh[16][16] = {};
		for (int j = 0; j < 16; ++j) {
			double s = A[b * 256 + j * 16 + j]; for (int k = 0; k < j; ++k) s -= Lh[j][k] * Lh[j][k];
			Lh[j][j] = sqrt(s);
			for (int i = j + 1; i < 16; ++i) { double t = A[b * 256 + i * 16 + j]; for (int k = 0; k < j; ++k) t -= Lh[i][k] * Lh[j][k]; Lh[i][j] = t / Lh[j][j]; }
		}
		for (int j = 0; j < 16; ++j) { Wh[j][j] = 1.0 / Lh[j][j]; for (int i = j + 1; i < 16; ++i) { double t = 0; for (int k = j; k < i; ++k) t += Lh[i][k] * Wh[k][j]; Wh[i][j] = -t / Lh[i][i]; } }
		for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { Lr[b * 256 + i * 16 + j] = Lh[i][j]; Wr[b * 256 + i * 16 + j] = Wh[i][j]; }
	}
	double *din, *dL, *dW; long long* dc;
	CK(hipMalloc(&din, A.size() * 8)); CK(hipMalloc(&dL, A.size() * 8)); CK(hipMalloc(&dW, A.size() * 8)); CK(hipMalloc(&dc, 24));
	CK(hipMemcpy(din, A.data(), A.size() * 8, hipMemcpyHostToDevice));
	for (int var = 0; var < 20; ++var) {
		if (only >= 0 && var != only) continue;
		if (var == 0) hipLaunchKernelGGL(diag_kernel<0>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 1) hipLaunchKernelGGL(diag_kernel<1>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 2) hipLaunchKernelGGL(diag_kernel<2>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 3) hipLaunchKernelGGL(diag_kernel<3>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 4) hipLaunchKernelGGL(diag2_kernel, dim3(1), dim3(128), nblk * 256 * 8 + sizeof(Comm), 0, din, dL, dW, nblk, reps, dc);
		else if (var == 5) hipLaunchKernelGGL(diag_kernel<4>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 6) hipLaunchKernelGGL(diag_kernel<8>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 7) hipLaunchKernelGGL(diag_kernel<12>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 8) hipLaunchKernelGGL(diag_kernel<16>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 9) hipLaunchKernelGGL(diag_kernel<32>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 10) hipLaunchKernelGGL(diag_kernel<33>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 11) hipLaunchKernelGGL(diag_lean_kernel<0>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 12) hipLaunchKernelGGL(diag_lean_kernel<1>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 13) hipLaunchKernelGGL(diag_pipe_kernel, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 14) hipLaunchKernelGGL(diag_kernel<64>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 15) hipLaunchKernelGGL(diag_kernel<128>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 16) hipLaunchKernelGGL(diag_kernel<256>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 17) hipLaunchKernelGGL(diag_kernel<384>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else if (var == 18) hipLaunchKernelGGL(diag_kernel<512>, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		else hipLaunchKernelGGL(diag_pipe2_kernel, dim3(1), dim3(64), nblk * 256 * 8, 0, din, dL, dW, nblk, reps, dc);
		CK(hipDeviceSynchronize());
		std::vector<double> Lg(A.size()), Wg(A.size()); long long cyc = 0;
		CK(hipMemcpy(Lg.data(), dL, A.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(Wg.data(), dW, A.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost));
		double eL = 0, eW = 0;
		for (size_t t = 0; t < A.size(); ++t) { eL = fmax(eL, fabs(Lg[t] - Lr[t])); eW = fmax(eW, fabs(Wg[t] - Wr[t])); }
		if (var == 4) { long long pw[3]; CK(hipMemcpy(pw, dc, 24, hipMemcpyDeviceToHost)); printf("variant 4: factor wave %.0f, inverse wave %.0f ticks per block (start of block -> own half done)\n", (double)pw[1] / (reps * nblk), (double)pw[2] / (reps * nblk)); }
		printf("variant %d: %.0f memtime ticks per 16x16 block (at 2.4 GHz: %.2f us), max |L - L_host| %.2e, max |W - W_host| %.2e\n", var, (double)cyc / (reps * nblk), (double)cyc / (reps * nblk) / 2400.0, eL, eW);
	}
	return 0;
}
