// rff.hip -- random Fourier feature embed  Z[i][j] = c * {cos | sin}(<W_j, x_i> (+ b_j)).
//
// The phase matrix X W^T is an NT contraction with K = d, so it runs on the MFMA GEMM of gemm.hip.
//   fp32 (the performance configuration, BASELINE config 5): trig + scale are fused into the GEMM's
//        store epilogue (hardware v_sin_f32 / v_cos_f32 on the phase reduced to revolutions); the
//        n x m phase matrix never exists in memory (the reference materialises it four times,
//        embedding.py:234-241).  At d = 64 the contraction (1.1e12 flop) and the 34 GB of output
//        cost about the same, so neither a VALU dot product nor an unfused pass reaches the roofline.
//   fp64 (the reference's dtype, used for parity): GEMM into `out`, then one in-place elementwise
//        pass with libm-accurate sin/cos.  (Fusing the fp64 libm bodies into the GEMM epilogue makes
//        hipcc spill every accumulator to scratch.)
//
// Column layout quirk kept from the reference (embedding.py:236-239): without a bias the cos
// half uses frequency rows 0..m/2-1 and the sin half uses the *other* rows m/2..m-1.
#include <type_traits>
#include "common.h"

namespace stpy {

// g_rff_wgs = 0 (lab knob):    stpy_tune key 19: workgroups of the streaming kernel (0 = 512, two per CU)
int g_rff_tile = 1;             // dedicated fp32 kernels (stpy_tune key 9): 1 = streaming kernel for large d = 64 shapes, tile kernel for
                                // the other d = 32 / 64 shapes; 2 = tile kernel only; 0 = always the GEMM epilogue

__global__ __launch_bounds__(256)
void rff_trig_f64_kernel(double* __restrict__ out, int64_t ldo, int64_t n, int m, int half, const double* __restrict__ bias,
                         const double* __restrict__ fscale, double scale)
{
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int mp = (m + 1) / 2;                           // feature pairs per row (an odd m -- cosine-only quadrature grids -- leaves a single)
	const int64_t total = n * (int64_t)mp;
	if (idx >= total) return;
	const int64_t i = idx / mp;
	const int j = (int)(idx - i * mp) * 2;                // two adjacent features per thread
	double* o = out + i * ldo + j;
	const bool two = j + 1 < m;
	double q0 = o[0], q1 = two ? o[1] : 0.0;
	if (bias) { q0 += bias[j]; if (two) q1 += bias[j + 1]; }
	o[0] = scale * (fscale ? fscale[j] : 1.0) * ((bias || j < half) ? cos(q0) : sin(q0));
	if (two) o[1] = scale * (fscale ? fscale[j + 1] : 1.0) * ((bias || j + 1 < half) ? cos(q1) : sin(q1));
}

// transposed embedding Phi^T (m x n): the feature index is the row
__global__ __launch_bounds__(256)
void rff_trig_f64_t_kernel(double* __restrict__ out, int64_t ldo, int64_t n, int m, int half, const double* __restrict__ bias,
                           const double* __restrict__ fscale, double scale)
{
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= (int64_t)m * n) return;
	const int j = (int)(idx / n);
	const int64_t i = idx - (int64_t)j * n;
	double q = out[(int64_t)j * ldo + i];
	if (bias) q += bias[j];
	out[(int64_t)j * ldo + i] = scale * (fscale ? fscale[j] : 1.0) * ((bias || j < half) ? cos(q) : sin(q));
}

// ---- fp32, d = 32 or 64, tile-aligned shapes: 128 x 64 output tile per workgroup, the whole contraction
// staged ONCE (no K loop, one barrier), 52 KiB of LDS so that THREE workgroups share a CU: the phases of a
// tile (operand fetch - MFMA - trig + 64 KiB of stores) are strictly serial inside a workgroup, and with
// two resident workgroups the MFMA pipe sat idle 45 % of the time.  Waves 2 x 2, wave tile 64 x 32.
// Both sin and cos go through v_cos_f32 (argument in revolutions): t = q/2pi + b/2pi (- 1/4 for the sine
// columns), reduced by v_fract.  The k index is permuted consistently for both operands (lane group kq
// reads the float4 at k = 16 g + 4 kq), which a dot product does not notice and which makes every LDS
// read a 16-byte one.  Column tiles vary fastest, so the eight XCDs each keep one eighth of W in their L2.
// `exp` (stpy_tune key 1, 0 in production) switches phases off for tools/rff_ablate.py: 1 no stores, 2 no MFMA,
// 4 no operand loads.
template <int D>
__global__ __launch_bounds__(256, 3)
void rff_tile_f32_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                         float* __restrict__ out, int64_t ldo, int col_tiles, int half,
                         const float* __restrict__ bias, float scale, int exp)
{
#if !STPY_LAB
	exp = 0;          // (timing ablations exist in the lab build only: folds every `exp` test away)
#endif
	typedef float v4f __attribute__((ext_vector_type(4)));
	constexpr int LD = D + 4;             // rows shift by 4 banks: a quarter wave's ds_read_b128 touches every bank once
	constexpr int F4 = D / 4;
	constexpr int CLD = 36;               // row stride of the output staging (see the store phase)
	__shared__ __attribute__((aligned(16))) float smem[(192 * LD > 4 * 64 * CLD) ? 192 * LD : 4 * 64 * CLD];
	float* const As = smem;
	float* const Bs = smem + 128 * LD;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
	const int r16 = lane & 15, kq = lane >> 4;
	const int ct = (int)(blockIdx.x % (unsigned)col_tiles);
	const int64_t row0 = (int64_t)(blockIdx.x / (unsigned)col_tiles) * 128;
	const int col0 = ct * 64;

	const float* xa = x + row0 * ldx;
	const float* wb = W + (int64_t)col0 * ldw;
	if (!(exp & 4)) {
#pragma unroll
	for (int it = 0; it < (128 * F4) / 256; ++it) {
		const int idx = tid + it * 256, r = idx / F4, c = idx % F4;
		*(v4f*)&As[r * LD + 4 * c] = *(const v4f*)(xa + (int64_t)r * ldx + 4 * c);
	}
#pragma unroll
	for (int it = 0; it < (64 * F4) / 256; ++it) {
		const int idx = tid + it * 256, r = idx / F4, c = idx % F4;
		*(v4f*)&Bs[r * LD + 4 * c] = *(const v4f*)(wb + (int64_t)r * ldw + 4 * c);
	}
	}
	__syncthreads();

	v4f acc[4][2];
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = v4f{0.f, 0.f, 0.f, 0.f};
	if (!(exp & 2))
#pragma unroll
	for (int g = 0; g < D / 16; ++g) {
		v4f a[4], b[2];
#pragma unroll
		for (int tm = 0; tm < 4; ++tm) a[tm] = *(const v4f*)&As[(wm * 64 + tm * 16 + r16) * LD + 16 * g + 4 * kq];
#pragma unroll
		for (int tn = 0; tn < 2; ++tn) b[tn] = *(const v4f*)&Bs[(wn * 32 + tn * 16 + r16) * LD + 16 * g + 4 * kq];
#pragma unroll
		for (int c = 0; c < 4; ++c)
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int tn = 0; tn < 2; ++tn)
					acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][c], b[tn][c], acc[tm][tn], 0, 0, 0);
	}

	constexpr float INV_2PI = 0.15915494309189535f;
	float off[2];
#pragma unroll
	for (int tn = 0; tn < 2; ++tn) {
		const int col = col0 + wn * 32 + tn * 16 + r16;
		off[tn] = bias ? bias[col] * INV_2PI : (col < half ? 0.f : -0.25f);
	}
	// The MFMA result layout gives 64-byte row segments per store instruction (16 lanes x 4 B).  Each wave
	// turns its 64 x 32 sub-tile through LDS (the operand images are dead after the barrier) and stores
	// 16 bytes per lane: 8 lanes = one 128-byte line, 8 full lines per instruction, 8 instructions.
	// (CLD = 36: rows 4 apart are 16 banks apart, so the four kq groups of a write do not collide)
	__syncthreads();
	float* const cw = smem + wave * (64 * CLD);
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i)
#pragma unroll
			for (int tn = 0; tn < 2; ++tn) {
				const float t = __builtin_amdgcn_fractf(__builtin_fmaf(acc[tm][tn][i], INV_2PI, off[tn]));
				cw[(tm * 16 + 4 * kq + i) * CLD + tn * 16 + r16] = scale * __builtin_amdgcn_cosf(t);
			}
	if ((exp & 1) && scale != 12345.f) return;
	const int rr = lane >> 3, c4 = (lane & 7) * 4;
	float* const obase = out + (row0 + wm * 64 + rr) * ldo + (col0 + wn * 32 + c4);
#pragma unroll
	for (int j = 0; j < 8; ++j) {
		const v4f v = *(const v4f*)&cw[(rr + 8 * j) * CLD + c4];
		__builtin_nontemporal_store(v, (v4f*)(obase + (int64_t)(8 * j) * ldo));
	}
}

// ---- fp32, d = 64, large tile-aligned shapes (m % 512 == 0): persistent, barrier-free streaming kernel.
// The phases of a tile in the kernel above are serial inside a workgroup and the co-resident workgroups run in
// convoy, so MFMA, trig and memory time add up (tools/rff_ablate.py).  Here every WAVE is its own pipeline:
//   * it keeps the A fragments of its 64 rows (the whole K = 64: 64 VGPRs) for a full sweep over its columns;
//   * per 32-column tile it needs 8 float4 of W straight from L2 into VGPRs (prefetched one tile ahead) -- no LDS
//     operand staging, no barrier, nothing shared between waves;
//   * the trig + store of tile j-1 (results parked in registers) is interleaved, instruction by instruction, with
//     the 128 MFMAs of tile j, so the VALU work rides in the MFMA issue shadow of the same wave;
//   * the finished tile is turned through a wave-private 9 KiB LDS patch so each store writes full 128-byte lines.
// Work split: blockIdx % 8 selects one eighth of the columns (with round-robin workgroup placement one XCD's L2
// holds exactly that eighth of W), blockIdx / 8 strides over the 128-row blocks.
__global__ __launch_bounds__(256, 2)
void rff_stream_f32_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                           float* __restrict__ out, int64_t ldo, int row_blocks, int cols_per_part, int half,
                           const float* __restrict__ bias, float scale, int exp)
{
#if !STPY_LAB
	exp = 0;          // (timing ablations exist in the lab build only: folds every `exp` test away)
#endif
	typedef float v4f __attribute__((ext_vector_type(4)));
	constexpr int CLD = 36;
	__shared__ __attribute__((aligned(16))) float smem[4 * 64 * CLD];
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;     // wave-uniform: addresses below split into SGPR base + 32-bit lane offset
	const int r16 = lane & 15, kq = lane >> 4;
	float* const cw = smem + wave * (64 * CLD);
	const int rr = lane >> 3, c4 = (lane & 7) * 4;
	constexpr float INV_2PI = 0.15915494309189535f;
	const int part = blockIdx.x & 7;
	const int tiles = cols_per_part / 64;                 // 32-column tiles per wave and row block (the two wn waves alternate)
	const int colp = part * cols_per_part;
	const unsigned w_lane = (unsigned)r16 * (unsigned)ldw + 4u * kq;       // lane part of a W fragment address (elements)
	const unsigned o_lane = (unsigned)rr * (unsigned)ldo + (unsigned)c4;   // lane part of an output address
	const unsigned st_lane = (unsigned)(4 * kq) * CLD + r16;               // lane part of a staging write
	const unsigned ld_lane = (unsigned)rr * CLD + c4;                      // lane part of a staging read

	for (int rb = blockIdx.x >> 3; rb < row_blocks; rb += gridDim.x >> 3) {
		const int64_t row0 = (int64_t)rb * 128 + wm * 64;
		v4f a[4][4];
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int g = 0; g < 4; ++g) a[tm][g] = *(const v4f*)(x + (row0 + tm * 16 + r16) * ldx + 16 * g + 4 * kq) * INV_2PI;     // phases in revolutions

		auto load_b = [&](v4f (&b)[2][4], int j) {
			if ((exp & 4) && j > 1) return;
			const float* const wb = W + (int64_t)(colp + (2 * j + wn) * 32) * ldw;          // uniform
#pragma unroll
			for (int tn = 0; tn < 2; ++tn)
#pragma unroll
				for (int g = 0; g < 4; ++g) b[tn][g] = *(const v4f*)((wb + (int64_t)(tn * 16) * ldw + 16 * g) + w_lane);
		};
		// MFMAs of tile j into `acc`, interleaved with trig + LDS staging of the previous tile's `res` (HAVE: there is one).
		// The accumulation starts from the column's phase offset (b/2pi, -1/4 for sine columns): no separate add later.
		auto tile = [&](auto have_tag, v4f (&acc)[4][2], const v4f (&b)[2][4], const v4f (&res)[4][2], int j) {
			constexpr bool HAVE = decltype(have_tag)::value;
			v4f offv[2];
#pragma unroll
			for (int tn = 0; tn < 2; ++tn) {
				const int col = colp + (2 * j + wn) * 32 + tn * 16 + r16;
				const float o = bias ? bias[col] * INV_2PI : (col < half ? 0.f : -0.25f);
				offv[tn] = v4f{o, o, o, o};
			}
#pragma unroll
			for (int g = 0; g < 4; ++g)
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int step = g * 4 + c;               // 16 steps x 8 MFMAs; 2 result elements of the previous tile per step
#pragma unroll
					for (int tm = 0; tm < 4; ++tm)
#pragma unroll
						for (int tn = 0; tn < 2; ++tn)
							acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm][g][c], b[tn][g][c], step == 0 ? offv[tn] : acc[tm][tn], 0, 0, 0);
					if constexpr (HAVE) {
						const int tm = step >> 2, i = step & 3;
#pragma unroll
						for (int tn = 0; tn < 2; ++tn)
							cw[st_lane + (unsigned)((tm * 16 + i) * CLD + tn * 16)] = scale * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(res[tm][tn][i]));
#pragma unroll
						for (int q = 0; q < 8; ++q) {         // one MFMA, then a slice of the VALU / LDS work, eight times
							__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
							__builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
							if (q == 3 || q == 7) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
		};
		auto finish = [&](const v4f (&res)[4][2]) {        // the last tile of a sweep: nothing left to hide it under
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int tn = 0; tn < 2; ++tn)
						cw[st_lane + (unsigned)((tm * 16 + i) * CLD + tn * 16)] = scale * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(res[tm][tn][i]));
		};
		auto flush = [&](int j) {                           // wave-private patch -> eight full-line stores per instruction
			if ((exp & 1) && scale != 12345.f) return;
			float* const ob = out + row0 * ldo + (colp + (2 * j + wn) * 32);             // uniform
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const v4f v = *(const v4f*)&cw[ld_lane + (unsigned)(8 * q * CLD)];
				__builtin_nontemporal_store(v, (v4f*)((ob + (int64_t)(8 * q) * ldo) + o_lane));
			}
		};

		// two tiles per trip (tiles is even) so that accumulator / result and the two W buffers swap roles without copies
		v4f b0[2][4], b1[2][4], acc0[4][2], acc1[4][2];
		load_b(b0, 0);
		for (int j = 0; j < tiles; j += 2) {
			load_b(b1, j + 1);
			if (j == 0) {
				tile(std::false_type{}, acc0, b0, acc1, j);
			} else {
				tile(std::true_type{}, acc0, b0, acc1, j);           // tile j; finishes tile j-1
				flush(j - 1);
			}
			load_b(b0, j + 2 < tiles ? j + 2 : j);
			tile(std::true_type{}, acc1, b1, acc0, j + 1);           // tile j+1; finishes tile j
			flush(j);
		}
		finish(acc1);
		flush(tiles - 1);
	}
}

#if STPY_LAB
// ---- the same streaming kernel with the MFMA operands SWAPPED (W tile as the "row" operand, x rows as the "column" operand):
// the accumulator of tile (tn, tm) then holds, in lane (r16, kq), the four CONSECUTIVE features tn*16 + 4 kq .. + 3 of sample
// tm*16 + r16 -- exactly one 16-byte store per MFMA tile, straight from the registers.  No LDS patch, no ds_write per element,
// no ds_read / wait before the stores: the trig of tile j-1 is converted in place and stored between the MFMAs of tile j.
// A store instruction writes 16 segments of 64 bytes; the other half of each 128-byte line follows from the same wave's next
// tn tile.  (stpy_tune key 9 = 3; measured against the LDS-patch form in one process by tools/rff_routes.py.)
__global__ __launch_bounds__(256, 2)
void rff_stream_direct_f32_kernel(const float* __restrict__ x, int64_t ldx, const float* __restrict__ W, int64_t ldw,
                                  float* __restrict__ out, int64_t ldo, int row_blocks, int cols_per_part, int half,
                                  const float* __restrict__ bias, float scale, int exp)
{
#if !STPY_LAB
	exp = 0;          // (timing ablations exist in the lab build only: folds every `exp` test away)
#endif
	typedef float v4f __attribute__((ext_vector_type(4)));
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
	const int r16 = lane & 15, kq = lane >> 4;
	constexpr float INV_2PI = 0.15915494309189535f;
	const int part = blockIdx.x & 7;
	const int tiles = cols_per_part / 64;
	const int colp = part * cols_per_part;
	const unsigned w_lane = (unsigned)r16 * (unsigned)ldw + 4u * kq;
	const unsigned o_lane = (unsigned)r16 * (unsigned)ldo + 4u * kq;            // sample r16 of a 16-row tile, features 4 kq ..

	for (int rb = blockIdx.x >> 3; rb < row_blocks; rb += gridDim.x >> 3) {
		const int64_t row0 = (int64_t)rb * 128 + wm * 64;
		v4f a[4][4];
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int g = 0; g < 4; ++g) a[tm][g] = *(const v4f*)(x + (row0 + tm * 16 + r16) * ldx + 16 * g + 4 * kq) * INV_2PI;

		auto load_b = [&](v4f (&b)[2][4], int j) {
			if ((exp & 4) && j > 1) return;
			const float* const wb = W + (int64_t)(colp + (2 * j + wn) * 32) * ldw;
#pragma unroll
			for (int tn = 0; tn < 2; ++tn)
#pragma unroll
				for (int g = 0; g < 4; ++g) b[tn][g] = *(const v4f*)((wb + (int64_t)(tn * 16) * ldw + 16 * g) + w_lane);
		};
		// MFMAs of tile j into `acc` (indexed [tn][tm]); between them the previous tile's `res` is turned into outputs in place and
		// stored: one (tn, tm) tile = 4 fract + 4 cos + 4 mul + 1 store, spread over the 16 steps of 8 MFMAs
		auto tile = [&](auto have_tag, v4f (&acc)[2][4], const v4f (&b)[2][4], v4f (&res)[2][4], int j) {
			constexpr bool HAVE = decltype(have_tag)::value;
			v4f offv[2];
#pragma unroll
			for (int tn = 0; tn < 2; ++tn) {
				const int col = colp + (2 * j + wn) * 32 + tn * 16 + 4 * kq;
				if (bias) offv[tn] = *(const v4f*)(bias + col) * INV_2PI;
				else {
#pragma unroll
					for (int i = 0; i < 4; ++i) offv[tn][i] = (col + i < half) ? 0.f : -0.25f;
				}
			}
			float* const ob = out + row0 * ldo + (colp + (2 * (j - 1) + wn) * 32);      // previous tile's output columns (uniform)
#pragma unroll
			for (int g = 0; g < 4; ++g)
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					const int step = g * 4 + c;
#pragma unroll
					for (int tn = 0; tn < 2; ++tn)
#pragma unroll
						for (int tm = 0; tm < 4; ++tm)
							acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[tn][g][c], a[tm][g][c], step == 0 ? offv[tn] : acc[tn][tm], 0, 0, 0);
					if constexpr (HAVE) {
						if ((step & 1) == 0) {                 // eight (tn, tm) tiles over sixteen steps: convert on even steps, store on odd ones
							const int t = step >> 1, tn = t >> 2, tm = t & 3;
#pragma unroll
							for (int i = 0; i < 4; ++i) res[tn][tm][i] = scale * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(res[tn][tm][i]));
						} else if (!((exp & 1) && scale != 12345.f)) {
							const int t = step >> 1, tn = t >> 2, tm = t & 3;
							__builtin_nontemporal_store(res[tn][tm], (v4f*)((ob + (int64_t)(tm * 16) * ldo + tn * 16) + o_lane));
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
		};
		auto finish = [&](v4f (&res)[2][4], int j) {           // last tile of the sweep
			if ((exp & 1) && scale != 12345.f) return;
			float* const ob = out + row0 * ldo + (colp + (2 * j + wn) * 32);
#pragma unroll
			for (int tn = 0; tn < 2; ++tn)
#pragma unroll
				for (int tm = 0; tm < 4; ++tm) {
#pragma unroll
					for (int i = 0; i < 4; ++i) res[tn][tm][i] = scale * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(res[tn][tm][i]));
					__builtin_nontemporal_store(res[tn][tm], (v4f*)((ob + (int64_t)(tm * 16) * ldo + tn * 16) + o_lane));
				}
		};
		v4f b0[2][4], b1[2][4], acc0[2][4], acc1[2][4];
		load_b(b0, 0);
		for (int j = 0; j < tiles; j += 2) {
			load_b(b1, j + 1);
			if (j == 0) tile(std::false_type{}, acc0, b0, acc1, j);
			else tile(std::true_type{}, acc0, b0, acc1, j);             // tile j; stores tile j-1
			load_b(b0, j + 2 < tiles ? j + 2 : j);
			tile(std::true_type{}, acc1, b1, acc0, j + 1);              // tile j+1; stores tile j
		}
		finish(acc1, tiles - 1);
	}
}
#endif  // STPY_LAB

// ---- fp32 embed with the contraction on the bf16 MATRIX cores (d = 64, the shapes of the streaming kernel above, workspace given).
// Why: v_mfma_f32_16x16x4_f32 runs on the SIMD's own fp32 lanes -- tools/mfma_filler_probe.hip: every VALU instruction next to it costs
// its full issue time (32.6 cycles per MFMA alone, 45.8 with one v_fma_f32, 49.8 with one v_cos_f32, +4 / +8 for each further one),
// and its peak (157 TFLOP/s) IS the packed-fp32 vector peak.  So the fp32-MFMA kernel above pays MFMA + trig + staging serially
// (7.0 ms + 2.4 ms at config 5, before the clock drops under that load).  The bf16 matrix pipe is a separate unit at 16x the rate.
// How, without giving up fp32 accuracy: every fp32 operand is split EXACTLY into three bf16 parts (x = x1 + x2 + x3: 8 + 8 + 8
// significant bits, by truncation), and the six products of weight >= 2^-16 are accumulated in fp32 by the MFMA, smallest first:
//   x.w = x1.w3 + (x1 + x2).w2 + (x1 + x2 + x3).w1 + [x2.w3 + x3.w2 + x3.w3 <= 2^-23 |x||w|, dropped: below one fp32 rounding]
// 96 v_mfma_f32_16x16x32_bf16 per 64 x 32 tile (1536 cycles) instead of 128 fp32 MFMAs (4096 cycles), and the trig / staging of
// the previous tile now genuinely overlaps them.  Against the oracle the error is that of the fp32-MFMA kernel (tests/test_gpu_configs.py).
// Config 5 (tools/rff_routes.py, tools/rff_ablate.py ws): 7.4 ms against 10.6 ms for the fp32-MFMA kernel; without stores 4.6 ms; the store
// stream alone (this pattern: 8 rows x 128 bytes per instruction, non-temporal) 6.35 ms = 5.4 TB/s, so the kernel is now bound by the HBM
// write as SURVEY section 8d classed it.  Tried on top, no gain: the four waves side by side (512 contiguous bytes per row), plain
// instead of non-temporal stores (10.1 ms), the trig chain pipelined across slots (kept: it costs nothing).
// W is split once per call by rff_split_w_kernel into the workspace, already in MFMA fragment order:
//   block (c16, kh, part) = 64 lanes x 16 bytes; lane l: column 16 c16 + (l & 15), k = 32 kh + 8 (l >> 4) .. + 7.
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float x, unsigned& h1, unsigned& h2, unsigned& h3)      // bf16 patterns of the three parts
{
	const unsigned u1 = __float_as_uint(x) & 0xffff0000u;
	const float r1 = x - __uint_as_float(u1);                  // exact: the low 16 significant bits
	const unsigned u2 = __float_as_uint(r1) & 0xffff0000u;
	const float r2 = r1 - __uint_as_float(u2);                 // exact: at most 8 significant bits left
	h1 = u1 >> 16; h2 = u2 >> 16; h3 = __float_as_uint(r2) >> 16;
}
__device__ __forceinline__ void split3x8(const float (&v)[8], u4v& p1, u4v& p2, u4v& p3)
{
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		unsigned a1, a2, a3, b1, b2, b3;
		split3(v[2 * q], a1, a2, a3);
		split3(v[2 * q + 1], b1, b2, b3);
		p1[q] = a1 | (b1 << 16); p2[q] = a2 | (b2 << 16); p3[q] = a3 | (b3 << 16);
	}
}

__global__ __launch_bounds__(256)
void rff_split_w_kernel(const float* __restrict__ W, int64_t ldw, int64_t blocks, u4v* __restrict__ ws)
{
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int lane = (int)(t & 63);
	const int64_t blk = t >> 6;                 // = c16 * 2 + kh
	if (blk >= blocks) return;
	const float* src = W + ((blk >> 1) * 16 + (lane & 15)) * ldw + (blk & 1) * 32 + 8 * (lane >> 4);
	typedef float v4f __attribute__((ext_vector_type(4)));
	const v4f lo = *(const v4f*)src, hi = *(const v4f*)(src + 4);
	const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	u4v p1, p2, p3;
	split3x8(v, p1, p2, p3);
	ws[(blk * 3 + 0) * 64 + lane] = p1;
	ws[(blk * 3 + 1) * 64 + lane] = p2;
	ws[(blk * 3 + 2) * 64 + lane] = p3;
}

// EXP: timing ablations, compiled as separate kernels (tools/rff_ablate.py ws): 1 no stores, 4 no W loads after the first tiles, 8 no trig / staging
template <int EXP>
__global__ __launch_bounds__(256, 2)
void rff_stream_bf16x3_kernel(const float* __restrict__ x, int64_t ldx, const u4v* __restrict__ ws, float* __restrict__ out, int64_t ldo,
                              int row_blocks, int cols_per_part, int half, const float* __restrict__ bias, float scale)
{
	constexpr int exp = EXP;
	typedef float v4f __attribute__((ext_vector_type(4)));
	constexpr int CLD = 36;
	__shared__ __attribute__((aligned(16))) float smem[2 * 4 * 64 * CLD];          // two 9 KiB staging patches per wave
	const int tid = threadIdx.x, lane = tid & 63;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
	const int r16 = lane & 15, kq = lane >> 4;
	float* const cw0 = smem + wave * (2 * 64 * CLD);
	float* const cw1 = cw0 + 64 * CLD;
	const int rr = lane >> 3, c4 = (lane & 7) * 4;
	constexpr float INV_2PI = 0.15915494309189535f;
	const int part = blockIdx.x & 7;
	const int tiles = cols_per_part / 64;                 // 32-column tiles per wave and row block (the two wn waves alternate)
	const int colp = part * cols_per_part;
	const unsigned o_lane = ((unsigned)rr * (unsigned)ldo + (unsigned)c4) * 4u;   // lane part of an output address, in bytes (ldo < 2^26: the launcher)
	// non-temporal 16-byte store at SGPR base + 32-bit lane offset (written out: through a pointer expression hipcc keeps the zero-extended
	// lane offset as a 64-bit register pair and adds the base on the vector ALU for every store -- two more VGPRs than this kernel has)
	// (s_nop 1: a store of more than 8 bytes reads its data registers for a couple of cycles after issue, and the instruction behind
	// an asm statement may already overwrite them -- the wait states hipcc inserts by itself behind a store it knows)
	auto store_nt = [&](const float* ubase, v4f v) { asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" :: "v"(o_lane), "v"(v), "s"(ubase) : "memory"); };
	const unsigned st_lane = (unsigned)(4 * kq) * CLD + r16;               // lane part of a staging write
	const unsigned ld_lane = (unsigned)rr * CLD + c4;                      // lane part of a staging read
	const unsigned b_lane = (unsigned)lane * 16u;                          // lane part (bytes) of a W fragment address

	for (int rb = blockIdx.x >> 3; rb < row_blocks; rb += gridDim.x >> 3) {
		const int64_t row0 = (int64_t)rb * 128 + wm * 64;
		// every row block starts its sweep over the column part at a different tile: the 64 workgroups of an XCD then write 32 different
		// column positions at any moment instead of marching through the same one (rows are a power of two apart: tools/ld_pad_probe.py)
		const int rot = (2 * rb) % tiles;
		auto ct = [&](int j) { const int t = j + rot; return t >= tiles ? t - tiles : t; };
		bf8 a[3][4][2];                                   // [part][row tile][K half]: the wave's 64 rows, resident for the sweep
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int kh = 0; kh < 2; ++kh) {
				const float* src = x + (row0 + tm * 16 + r16) * ldx + kh * 32 + 8 * kq;
				const v4f lo = *(const v4f*)src * INV_2PI, hi = *(const v4f*)(src + 4) * INV_2PI;     // phases in revolutions
				const float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
				u4v p1, p2, p3;
				split3x8(v, p1, p2, p3);
				a[0][tm][kh] = __builtin_bit_cast(bf8, p1); a[1][tm][kh] = __builtin_bit_cast(bf8, p2); a[2][tm][kh] = __builtin_bit_cast(bf8, p3);
			}

		bf8 b[3][2][2];                                   // [part][column tile][K half]: ONE buffer; a part is re-loaded for the next tile
		auto load_b = [&](int p, int j) {                  // as soon as the products that read it have been issued
			if constexpr ((exp & 4) != 0) { if (j > 1) return; }
			const char* const wb = (const char*)(ws + (int64_t)((colp + (2 * ct(j) + wn) * 32) >> 4) * (2 * 3 * 64));          // uniform
#pragma unroll
			for (int tn = 0; tn < 2; ++tn)
#pragma unroll
				for (int kh = 0; kh < 2; ++kh) b[p][tn][kh] = __builtin_bit_cast(bf8, *(const u4v*)((wb + ((tn * 2 + kh) * 3 + p) * 1024) + b_lane));   // SGPR base + 32-bit lane offset
		};
		// The 96 MFMAs of tile j into `acc` (smallest products first).  Slipped in between them: after every third one the trig +
		// LDS staging of one result element of tile j-1 (`res`, into patch `cwt`); every twelfth, one 8-row slice of tile j-2 read
		// back from the OTHER patch (`cwf`) and, nine MFMAs later, stored -- full 128-byte lines, the stores spread over the tile
		// instead of a burst at its end (a burst stalls the wave at the store queue: 7.7 ms against 5.0 ms without stores);
		// each W part is re-loaded for tile jn once its last reader is out.
		auto tile = [&](auto have_tag, auto flush_tag, v4f (&acc)[4][2], const v4f (&res)[4][2], int j, int jn, float* cwt, const float* cwf) {
			constexpr bool HAVE = decltype(have_tag)::value, FLUSH = decltype(flush_tag)::value;
			constexpr int AP[6] = {0, 0, 1, 0, 1, 2}, BP[6] = {2, 1, 1, 0, 0, 0};
			float off[2];                                       // the accumulation starts from the column's phase offset (b / 2pi; -1/4 turns cos into sin)
#pragma unroll
			for (int tn = 0; tn < 2; ++tn) {
				const int col = colp + (2 * ct(j) + wn) * 32 + tn * 16 + r16;
				off[tn] = bias ? bias[col] * INV_2PI : (col < half ? 0.f : -0.25f);
			}
			float* const ob = out + row0 * ldo + (colp + (2 * ct(j >= 2 ? j - 2 : j) + wn) * 32);             // uniform: tile j-2's first element (unused while j < 2)
			v4f fv;
			float tf[2], tc[2];
#pragma unroll
			for (int grp = 0; grp < 6; ++grp) {
#pragma unroll
				for (int kh = 0; kh < 2; ++kh)
#pragma unroll
					for (int tm = 0; tm < 4; ++tm)
#pragma unroll
						for (int tn = 0; tn < 2; ++tn) {
							const int idx = ((grp * 2 + kh) * 4 + tm) * 2 + tn;          // 0 .. 95
							acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AP[grp]][tm][kh], b[BP[grp]][tn][kh], (grp == 0 && kh == 0) ? v4f{off[tn], off[tn], off[tn], off[tn]} : acc[tm][tn], 0, 0, 0);
							if constexpr (FLUSH) {
								if (idx % 12 == 1 && !(exp & 1)) fv = *(const v4f*)&cwf[ld_lane + (unsigned)(8 * (idx / 12) * CLD)];
								if (idx % 12 == 10 && !(exp & 1)) store_nt(ob + (int64_t)(8 * (idx / 12)) * ldo, fv);
							}
							if (idx % 3 == 2) {
								// the trig chain of an element (v_fract -> v_cos -> v_mul -> ds_write) as a three-stage pipeline over the
								// slots, so that no instruction waits for the one before it (an in-order wave stalled on a transcendental's
								// latency cannot issue its next MFMA either): slot e: fract of e, cos of e-1, scale + staging of e-2
								if (HAVE && !(exp & 8)) {
									const int e = idx / 3;
									if (e >= 2) { const int w = e - 2, em = w >> 3, ei = (w >> 1) & 3, en = w & 1; cwt[st_lane + (unsigned)((em * 16 + ei) * CLD + en * 16)] = scale * tc[w & 1]; }
									if (e >= 1) tc[(e - 1) & 1] = __builtin_amdgcn_cosf(tf[(e - 1) & 1]);
									{ const int em = e >> 3, ei = (e >> 1) & 3, en = e & 1; tf[e & 1] = __builtin_amdgcn_fractf(res[em][en][ei]); }
								}
							}
							__builtin_amdgcn_sched_barrier(0);          // program order is the schedule: the kernel sits at the 256-VGPR limit and any hoisting spills
						}
				// W parts: part 0 (the largest, read last) for THIS tile after group 0, parts 2 and 1 for the NEXT tile once their last
				// readers are out -- at the tile's start, when the previous results are still all live, only two parts are held
				if (grp == 0) load_b(0, j);
				if (grp == 2) load_b(2, jn);
				if (grp == 3) load_b(1, jn);
			}
			if (HAVE && !(exp & 8)) {        // drain the pipeline: elements 30 and 31
				cwt[st_lane + (unsigned)((3 * 16 + 3) * CLD + 0 * 16)] = scale * tc[0];
				cwt[st_lane + (unsigned)((3 * 16 + 3) * CLD + 1 * 16)] = scale * __builtin_amdgcn_cosf(tf[1]);
			}
		};
		auto finish = [&](const v4f (&res)[4][2], float* cwt) {        // the last tile of a sweep: nothing left to hide it under
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int i = 0; i < 4; ++i)
#pragma unroll
					for (int tn = 0; tn < 2; ++tn)
						cwt[st_lane + (unsigned)((tm * 16 + i) * CLD + tn * 16)] = scale * __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(res[tm][tn][i]));
		};
		auto flush = [&](int j, const float* cwf) {                     // ... and the last two tiles' stores
			if constexpr ((exp & 1) != 0) { if (scale != 12345.f) return; }
			float* const ob = out + row0 * ldo + (colp + (2 * ct(j) + wn) * 32);             // uniform
#pragma unroll
			for (int q = 0; q < 8; ++q) {
				const v4f v = *(const v4f*)&cwf[ld_lane + (unsigned)(8 * q * CLD)];
				store_nt(ob + (int64_t)(8 * q) * ldo, v);
				if (q == 3) __builtin_amdgcn_sched_barrier(0);          // two batches of four: 16 transient VGPRs, not 32 (the kernel sits at the 256 limit)
			}
		};

		v4f acc0[4][2], acc1[4][2];
		load_b(2, 0); load_b(1, 0);
		// two tiles per trip (tiles is even): accumulator / result and the two patches swap roles without copies.
		// Even tile j: results of j-1 (acc1) -> patch 1, stores of j-2 from patch 0; odd tile: the other way round.
		for (int j = 0; j < tiles; j += 2) {
			const int jn2 = j + 2 < tiles ? j + 2 : j + 1;
			if (j == 0) {
				tile(std::false_type{}, std::false_type{}, acc0, acc1, 0, 1, cw1, cw0);
				tile(std::true_type{}, std::false_type{}, acc1, acc0, 1, jn2, cw0, cw1);
			} else {
				tile(std::true_type{}, std::true_type{}, acc0, acc1, j, j + 1, cw1, cw0);
				tile(std::true_type{}, std::true_type{}, acc1, acc0, j + 1, jn2, cw0, cw1);
			}
		}
		flush(tiles - 2, cw0);               // tile T-2 (even) was staged into patch 0 during tile T-1
		finish(acc1, cw1);                   // tile T-1 (odd) -> patch 1
		flush(tiles - 1, cw1);
	}
}

int64_t rff_workspace_bytes(int elem, int64_t n, int d, int64_t m)
{
	// the three bf16 parts of W in fragment order (rff_stream_bf16x3_kernel): fp32, d = 64 only
	if (elem != 4 || d != 64 || m % 1024 != 0 || n % 128 != 0 || n < 8192) return 0;
	return m * 64 * 3 * 2;
}

template <typename T>
int rff_embed(const T* x, int64_t n, int64_t ldx, int d, const T* W, int64_t ldw, int64_t m,
              const T* bias, const T* feat_scale, double scale, T* out, int64_t ldo, int transposed, void* work, int64_t work_bytes, hipStream_t st);

// (an odd m only makes sense without the cos | sin split: biased features, or cosine-only quadrature grids that pass a
// zero bias -- embedding.py:84-85 demands an even m for everything else)
template <>
int rff_embed<float>(const float* x, int64_t n, int64_t ldx, int d, const float* W, int64_t ldw, int64_t m,
                     const float* bias, const float* feat_scale, double scale, float* out, int64_t ldo, int transposed, void* work, int64_t work_bytes, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (m % 2 != 0 && !bias) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	// per-feature amplitudes (quadrature embeddings: m is small there) take the GEMM-epilogue route
	if (!feat_scale && g_rff_tile >= 1 && g_rff_tile != 2 && !transposed && d == 64 && n % 128 == 0 && m % 1024 == 0 && n >= 8192 && ldw < ((int64_t)1 << 27) && ldo < ((int64_t)1 << 28) && ldx % 4 == 0 && ldw % 4 == 0 &&
	    ldo % 4 == 0 && (((uintptr_t)x | (uintptr_t)W | (uintptr_t)out) & 15) == 0 && n / 128 < (int64_t)INT32_MAX && m < (int64_t)INT32_MAX) {
		const int row_blocks = (int)(n / 128);
		int wgs = g_rff_wgs > 0 ? (g_rff_wgs + 7) / 8 * 8 : 512;          // two per CU; a multiple of eight (one column part per XCD); stpy_tune key 19
		if (wgs > 8 * row_blocks) wgs = 8 * row_blocks;
		// with a workspace: the contraction on the bf16 matrix cores from an exact three-way split of both operands (stpy_tune key 9 = 5
		// keeps the fp32-MFMA kernel for A/B runs)
		if (work && work_bytes >= rff_workspace_bytes(4, n, d, m) && g_rff_tile == 1 && (((uintptr_t)work) & 15) == 0 && m * 6 * 64 < ((int64_t)1 << 31) && ldo < ((int64_t)1 << 26)) {
			const int64_t blocks = (m / 16) * 2;
			hipLaunchKernelGGL(rff_split_w_kernel, dim3((unsigned)((blocks * 64 + 255) / 256)), dim3(256), 0, st, W, ldw, blocks, (u4v*)work);
			int rc = check_launch("rff_split_w");
			if (rc) return rc;
#define STPY_BF3(E) hipLaunchKernelGGL(rff_stream_bf16x3_kernel<E>, dim3((unsigned)wgs), dim3(256), 0, st, x, ldx, (const u4v*)work, out, ldo, row_blocks, (int)(m / 8), (int)(m / 2), bias, (float)scale)
#if STPY_LAB
			switch (g_gemm_exp) {
			case 1: STPY_BF3(1); break;
			case 8: STPY_BF3(8); break;
			default: STPY_BF3(0); break;
			}
#else
			STPY_BF3(0);
#endif
#undef STPY_BF3
			return check_launch("rff_stream_bf16x3");
		}
#if STPY_LAB
		if (g_rff_tile == 3) hipLaunchKernelGGL(rff_stream_direct_f32_kernel, dim3((unsigned)wgs), dim3(256), 0, st, x, ldx, W, ldw, out, ldo, row_blocks, (int)(m / 8), (int)(m / 2), bias, (float)scale, g_gemm_exp);
		else
#endif
		hipLaunchKernelGGL(rff_stream_f32_kernel, dim3((unsigned)wgs), dim3(256), 0, st, x, ldx, W, ldw, out, ldo, row_blocks, (int)(m / 8), (int)(m / 2), bias, (float)scale, g_gemm_exp);
		return check_launch("rff_stream_f32");
	}
	if (!feat_scale && g_rff_tile && !transposed && (d == 32 || d == 64) && n % 128 == 0 && m % 64 == 0 && ldx % 4 == 0 && ldw % 4 == 0 &&
	    (((uintptr_t)x | (uintptr_t)W | (uintptr_t)out) & 15) == 0 && ldo % 4 == 0 && (n / 128) * (m / 64) < (int64_t)INT32_MAX && m < (int64_t)INT32_MAX) {
		const int col_tiles = (int)(m / 64);
		const dim3 grid((unsigned)((n / 128) * col_tiles));
		if (d == 64) hipLaunchKernelGGL(rff_tile_f32_kernel<64>, grid, dim3(256), 0, st, x, ldx, W, ldw, out, ldo, col_tiles, (int)(m / 2), bias, (float)scale, g_gemm_exp);
		else         hipLaunchKernelGGL(rff_tile_f32_kernel<32>, grid, dim3(256), 0, st, x, ldx, W, ldw, out, ldo, col_tiles, (int)(m / 2), bias, (float)scale, g_gemm_exp);
		return check_launch("rff_tile_f32");
	}
	RffEpilogue<float> epi{(int)(m / 2), (float)scale, bias, transposed ? 1 : 0, feat_scale};
	if (transposed) return gemm_nt<float>(m, n, d, W, ldw, x, ldx, out, ldo, (float*)nullptr, 0, 2, 0, st, nullptr, &epi);
	return gemm_nt<float>(n, m, d, x, ldx, W, ldw, out, ldo, (float*)nullptr, 0, 2, 0, st, nullptr, &epi);
}

template <>
int rff_embed<double>(const double* x, int64_t n, int64_t ldx, int d, const double* W, int64_t ldw, int64_t m,
                      const double* bias, const double* feat_scale, double scale, double* out, int64_t ldo, int transposed, void* work, int64_t work_bytes, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (m % 2 != 0 && !bias) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	if (m > INT32_MAX) { set_error("rff_embed: m exceeds int32"); return -7; }
	if (transposed) {
		int rc = gemm_nt<double>(m, n, d, W, ldw, x, ldx, out, ldo, (double*)nullptr, 0, 0, 0, st);
		if (rc) return rc;
		const int64_t total = m * n;
		hipLaunchKernelGGL(rff_trig_f64_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, ldo, n, (int)m, (int)(m / 2), bias, feat_scale, scale);
		return check_launch("rff_trig_f64_t");
	}
	int rc = gemm_nt<double>(n, m, d, x, ldx, W, ldw, out, ldo, (double*)nullptr, 0, 0, 0, st);
	if (rc) return rc;
	const int64_t total = n * ((m + 1) / 2);
	hipLaunchKernelGGL(rff_trig_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, ldo, n, (int)m, (int)(m / 2), bias, feat_scale, scale);
	return check_launch("rff_trig_f64");
}

}  // namespace stpy
