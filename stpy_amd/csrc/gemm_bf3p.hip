// fp32 trailing updates of the factorisation from a panel that has been split into its three bf16 planes ONCE, in memory.
//
// gemm_nt_bf3_kernel (gemm.hip) splits every fp32 operand value on the fly: each workgroup re-splits its A and B rows at every K step
// (4 VALU operations per value + the byte permutes + 12 LDS stores per thread), and those phases are what keeps its matrix pipe
// at 55-60 % busy.  In the factorisation both operands of every trailing update are rows of ONE panel that is read by hundreds of tiles:
// here the panel is split once (bf3_split: a streaming pass, 4 B read + 6 B written per value, ~0.1 ms for a 65 536 x 1024 panel
// against 10-14 ms of update) into three row-major bf16 planes, and the update kernel below has no vector arithmetic in its K loop at
// all: the planes go from global memory straight into LDS (global_load_lds_dwordx4, the 16-byte chunk swizzle applied by the choice of
// SOURCE address, as in the fp64 kernels), and the waves only read fragments and issue MFMAs.
//
// Tile: 256 x 128 per workgroup of 8 waves (4 x 2, each 64 x 64), ONE workgroup per CU, two LDS stages of 3 planes x 384 rows x 64 B
// = 72 KiB each.  Per K step of 32 a wave reads 24 fragments (ds_read_b128) for 96 MFMAs; the CU's LDS moves 192 KiB (1536 cycles at
// 128 B/clk) under 3072 MFMA cycles per SIMD.  The arithmetic -- the exact truncation split, the six products smallest first, the
// two-level accumulation in chunks of four K steps, -C in the accumulator and the sign flipped at the store -- is that of
// gemm_nt_bf3_kernel instruction for instruction per output element, so both routes give bit-identical results
// (tests/test_gpu_kernels.py::test_f32_factor_presplit_route_is_bit_identical).
#include "common.h"
#include <atomic>

namespace stpy {

constexpr int P_TM = 256, P_TN = 128, P_BK = 32;
constexpr int P_ROWS = P_TM + P_TN;                   // rows of one plane in a stage: A rows 0..255, B rows 256..383
constexpr int P_PLANE = P_ROWS * 64;                  // 24 KiB
constexpr int P_STAGE = 3 * P_PLANE;                  // 72 KiB
constexpr int P_LDS = 2 * P_STAGE;                    // 144 KiB
constexpr int P_THREADS = 512;

struct Bf3pArgs {
	const unsigned short* pl;       // plane 0; plane t at pl + t * pstride (elements)
	int64_t ldp, pstride;
	int64_t arow, brow;             // first plane row of the A operand / of the B operand
	float* C;
	int64_t ldc;
	int m, n, k;
	int tri;                        // lower 128 x 128 tiles only (m == n)
	int nst_m, nst_n, nsuper;       // super-tiles of (256 << stl) x (256 << stl) elements
	int stl;                        // 2: 1024 x 1024 = 4 row pairs x 8 column tiles = 32 workgroups; 1: 512 x 512 = 2 x 4 = 8 workgroups
	int exp;                        // lab build: timing experiments (results wrong when != 0)
	// K split over blockIdx.y (few output tiles, long K): chunk c covers K blocks [c * kchunk, ...); chunk 0 applies (op) to C, chunk c >= 1
	// STORES its partial product to ws + (c - 1) * ws_stride (leading dimension ldws), summed into C afterwards in a fixed order
	int kchunk;
	float* ws;
	int64_t ldws, ws_stride;
};

// ---- the split: a thread takes 8 consecutive values of a row
__global__ __launch_bounds__(256)
void bf3_split_kernel(const float* __restrict__ X, int64_t ldx, int rows, int cols8, unsigned short* __restrict__ pl, int64_t ldp, int64_t pstride, int64_t prow0)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	typedef unsigned u4v __attribute__((ext_vector_type(4)));
	// 64 consecutive threads fill one 1 KiB image in order: thread -> (16-row group, K block, row in group, PHYSICAL chunk)
	const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int kbs = cols8 >> 2;                                   // K blocks of X
	const int64_t im = t >> 6;
	const int within = (int)(t & 63);
	const int grp = (int)(im / kbs), kb = (int)(im - (int64_t)grp * kbs);
	const int r = grp * 16 + (within >> 2);
	if (r >= rows) return;
	const int64_t prow = prow0 + r;
	const int c = (within & 3) ^ (int)((prow >> 1) & 3);          // the logical chunk that lives at this physical position
	const float* const src = X + (int64_t)r * ldx + kb * 32 + c * 8;
	const v4f a = *(const v4f*)src, b = *(const v4f*)(src + 4);
	const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
	unsigned part[3][4];
#pragma unroll
	for (int q = 0; q < 4; ++q) {
		unsigned h[2][3];
#pragma unroll
		for (int e = 0; e < 2; ++e) {
			const float v = x[2 * q + e];
			const unsigned u1 = __float_as_uint(v) & 0xffff0000u;
			const float r1 = v - __uint_as_float(u1);                  // exact: the low 16 significant bits
			const unsigned u2 = __float_as_uint(r1) & 0xffff0000u;
			const float r2 = r1 - __uint_as_float(u2);                 // exact: at most 8 significant bits left
			h[e][0] = u1; h[e][1] = u2; h[e][2] = __float_as_uint(r2);
		}
#pragma unroll
		for (int pt = 0; pt < 3; ++pt) part[pt][q] = __builtin_amdgcn_perm(h[1][pt], h[0][pt], 0x07060302u);          // high halves: [odd value | even value]
	}
	const int64_t img = ((prow >> 4) * (ldp >> 5) + kb) * 512 + within * 8;
#pragma unroll
	for (int pt = 0; pt < 3; ++pt)
		*(u4v*)(pl + pt * pstride + img) = u4v{part[pt][0], part[pt][1], part[pt][2], part[pt][3]};
}

// X: rows x cols fp32 (row i = plane row prow0 + i).  Planes: plane t at pl + t * pstride; a plane is TILE-MAJOR -- for every group of 16 rows and every
// block of 32 columns one contiguous 1 KiB image in exactly the (swizzled) order the update kernel wants in LDS, groups of a row block side by side
// (ldp / 32 images per group).  One LDS-DMA wave instruction of the update kernel then reads 1 KiB = eight whole cache lines; with row-major planes it
// read sixteen half lines, and the operand stream out of the L2s (the kernel's co-limit) moved twice the bytes it used.
int bf3_split(const float* X, int64_t ldx, int64_t rows, int64_t cols, unsigned short* pl, int64_t ldp, int64_t pstride, int64_t prow0, hipStream_t st)
{
	if (rows <= 0 || cols <= 0) return 0;
	if (cols % 32 != 0 || ldx % 4 != 0 || ldp % 32 != 0 || cols > ldp || pstride % 8 != 0 || ((uintptr_t)X & 15) || ((uintptr_t)pl & 15) || rows > INT32_MAX || prow0 < 0 || prow0 % 16 != 0 || rows % 16 != 0) {
		set_error("bf3_split: operands must be 16-byte aligned with cols a multiple of 32");
		return -6;
	}
	const int64_t threads = rows * (cols / 8);
	hipLaunchKernelGGL(bf3_split_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, X, ldx, (int)rows, (int)(cols / 8), pl, ldp, pstride, prow0);
	return check_launch("bf3_split");
}

// ---- C (op) A B^T  (A = plane rows arow .., B = plane rows brow ..), 256 x 128 tiles.  ACC 0: C = A B^T, 1: C -= A B^T, 2: C += A B^T
template <int ACC>
__global__ __launch_bounds__(P_THREADS, 1)
void gemm_bf3p_kernel(Bf3pArgs p)
{
	constexpr bool SUB = ACC == 1, LOADC = ACC != 0;
	typedef float v4f __attribute__((ext_vector_type(4)));
	typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
	typedef unsigned u4v __attribute__((ext_vector_type(4)));
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

	// block -> tile: 32 workgroups (4 row pairs x 8 column tiles = 1024 x 1024 elements) of one super-tile share an XCD (b & 7)
	const int b = blockIdx.x;
	// (small problems take 512 x 512 super-tiles: with 36 super-tiles of 1024 x 1024 -- n = 8192, lower triangle -- the XCDs get 116 to 148 tiles
	// each and the fullest one sets the time: 5 rounds' worth for 4.1 rounds of work)
	const int wbits = 2 * p.stl + 1;                              // log2(workgroups per super-tile)
	const int S = (b & 7) + 8 * (b >> (3 + wbits));
	const int w = (b >> 3) & ((1 << wbits) - 1);
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
	const int row0 = __builtin_amdgcn_readfirstlane(((si << p.stl) + (w >> (p.stl + 1))) * P_TM);
	const int col0 = __builtin_amdgcn_readfirstlane(((sj << (p.stl + 1)) + (w & ((2 << p.stl) - 1))) * P_TN);
	if (row0 >= p.m || col0 >= p.n) return;
	if (p.tri && col0 > row0 + 128) return;                       // both 128-row halves lie above the diagonal
	const int chunk = blockIdx.y;
	const int kbeg = chunk * p.kchunk;
	const int KT = ((p.k - kbeg < p.kchunk ? p.k - kbeg : p.kchunk)) / P_BK;
	const bool first_chunk = chunk == 0;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, kq = lane >> 4;
	// a wave computes iff its rows exist and its 128 x 128 tile is on or below the diagonal (the tiles gemm_nt_bf3_kernel computes)
	const int trow = row0 + (wm >> 1) * 128;
	const bool active = trow < p.m && !(p.tri && col0 > trow);

	// ---- LDS-DMA: a wave instruction moves 16 plane rows (64 B each) = 1 KiB; wave w moves row groups w, w + 8, w + 16 of every plane.
	// Lane l lands at row l >> 2, physical chunk l & 3 of its group.
	// (the planes are tile-major, see bf3_split: a group's image of K block kb is 1 KiB at ((row / 16) * (ldp / 32) + kb) * 1024 bytes, already swizzled)
	unsigned voff[3];
#pragma unroll
	for (int s = 0; s < 3; ++s) {
		const int g = wave + 8 * s;
		int64_t prow;                                             // first plane row of the group
		if (g < 16) { int rr = row0 + g * 16; if (rr > p.m - 16) rr = p.m - 16; prow = p.arow + rr; }
		else prow = p.brow + col0 + (g - 16) * 16;
		voff[s] = (unsigned)((prow >> 4) * (p.ldp >> 5) * 1024 + lane * 16);
	}
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
	auto dma_one = [&](const unsigned short* gbase, unsigned vo, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(vo), "s"(gbase), "s"(laddr) : "memory");
	};
	auto dma_plane = [&](int stage, int k0, int pt) {
		const unsigned short* const ub = p.pl + pt * p.pstride + ((kbeg + k0) >> 5) * 512;              // uniform: K block (kbeg + k0) / 32
#pragma unroll
		for (int s = 0; s < 3; ++s)
			dma_one(ub, voff[s], lds0 + (unsigned)(stage * P_STAGE + pt * P_PLANE + (wave + 8 * s) * 1024));
	};
	auto dma_stage = [&](int stage, int k0) {
#pragma unroll
		for (int pt = 0; pt < 3; ++pt) dma_plane(stage, k0, pt);
	};
	dma_stage(0, 0);

	// ---- accumulators: zero, or the C tile (negated when subtracting: the products are accumulated on -C and the sign flipped at the store)
	v4f acc[4][4];
	float* const ctile = first_chunk ? p.C + (int64_t)row0 * p.ldc + col0 : p.ws + (int64_t)(chunk - 1) * p.ws_stride + (int64_t)row0 * p.ldws + col0;
	const unsigned ldc32 = (unsigned)(first_chunk ? p.ldc : p.ldws);
	if (active) {
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const float* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + 4 * kq + i) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = (LOADC && first_chunk) ? crow[tn * 16] : 0.f;          // (subtracting: negated at the first chunk's addition)
			}
	}
	// stage 0 has landed: its nine DMA instructions are this wave's OLDEST vector-memory operations, the 64 loads of the C tile behind them
	// may stay in flight (the accumulators are not touched before the end of the first chunk, four K steps from here)
	if (active && LOADC && first_chunk) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
	else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

	const unsigned frag = (unsigned)r16 * 64u + (unsigned)((kq ^ ((r16 >> 1) & 3)) << 4);
	const unsigned a_off = (unsigned)(wm * 64) * 64u + frag, b_off = (unsigned)(P_TM + wn * 64) * 64u + frag;
	constexpr int AP[6] = {0, 0, 1, 0, 1, 2}, BP[6] = {2, 1, 1, 0, 0, 0};
	constexpr int CHUNK = 4;
	v4f part[4][4];
	// One K step: the DMA of stage kt + 1 goes out plane by plane behind the first three MFMA groups (the step's own fragment reads get the
	// LDS first; 1 % over issuing it all up front) and has the rest of the step to land; one barrier per step.
	// (tools/bf3p_bench.py, n = 32 768, K = 1024, all-zero operands so that the clock does not move: the step's MFMAs + fragment reads alone
	// 3.68 ms, its operand DMA alone 3.30 ms, both 4.14-4.19 ms, C tile + barriers alone 0.87 ms -- the operand stream, 72 KiB per CU and step
	// out of the L2s, is nearly as long as the arithmetic.  With ROW-major planes -- sixteen 64-byte half lines per DMA instruction instead of
	// eight whole lines -- the DMA alone took 3.73 ms and both 4.44.  Keeping TWO stages in flight -- fragments pulled into registers first, a
	// second barrier, stage kt + 2 fetched into the buffer just read -- shortened the DMA-only time by 9 % and lengthened the arithmetic by
	// 7 %: 4.80 ms together; not kept.  Random operands: 5.06 ms against 5.87 for gemm_nt_bf3_kernel -- the clock drops under real data.)
	auto kstep = [&](auto first_tag, int kt) {
		constexpr bool FIRST = decltype(first_tag)::value;
		const bool more = kt + 1 < KT && !(STPY_LAB && (p.exp & 1));
		if (active && !(STPY_LAB && (p.exp & 2))) {
			const unsigned char* const sb = smem + (kt & 1) * P_STAGE;
			bf8 fa[3][4], fb[3][4];
#pragma unroll
			for (int pt = 0; pt < 3; ++pt)
#pragma unroll
				for (int t = 0; t < 4; ++t) {
					fa[pt][t] = __builtin_bit_cast(bf8, *(const u4v*)(sb + pt * P_PLANE + a_off + t * 16 * 64));
					fb[pt][t] = __builtin_bit_cast(bf8, *(const u4v*)(sb + pt * P_PLANE + b_off + t * 16 * 64));
				}
#pragma unroll
			for (int g = 0; g < 6; ++g) {
#pragma unroll
				for (int tm = 0; tm < 4; ++tm)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						part[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[AP[g]][tm], fb[BP[g]][tn], (FIRST && g == 0) ? v4f{0.f, 0.f, 0.f, 0.f} : part[tm][tn], 0, 0, 0);
				if (more && g < 3) {
					__builtin_amdgcn_sched_barrier(0);
					dma_plane((kt + 1) & 1, (kt + 1) * P_BK, g);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
		}
		else if (more) dma_stage((kt + 1) & 1, (kt + 1) * P_BK);
		// the next stage has landed (this wave's share) and every wave has read this one
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
	};
	for (int kc = 0; kc < KT; kc += CHUNK) {
		kstep(std::true_type{}, kc);
		for (int j = 1; j < CHUNK && kc + j < KT; ++j) kstep(std::false_type{}, kc + j);
		if (active) {
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) acc[tm][tn] = (SUB && kc == 0) ? part[tm][tn] - acc[tm][tn] : acc[tm][tn] + part[tm][tn];          // (-C) + p = p - C, bit for bit
		}
	}
	if (active) {
#pragma unroll
		for (int tm = 0; tm < 4; ++tm)
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				float* const crow = ctile + ((unsigned)(wm * 64 + tm * 16 + 4 * kq + i) * ldc32 + (unsigned)(wn * 64 + r16));
#pragma unroll
				for (int tn = 0; tn < 4; ++tn) crow[tn * 16] = SUB ? -acc[tm][tn][i] : acc[tm][tn][i];
			}
	}
}

// K split (nch > 1, modes 0 and 5 only): chunk 0 applies (op) to C, chunk c >= 1 stores its partial product into ws[c - 1] (n-leading,
// m x n floats each); the caller adds them to C in a fixed order (syrk_planes below).
int gemm_nt_bf3p(int64_t m, int64_t n, int64_t k, const unsigned short* pl, int64_t ldp, int64_t pstride, int64_t arow, int64_t brow,
                 float* C, int64_t ldc, int tri, hipStream_t st, int mode, int nch, float* ws)
{
	if (mode != 0 && mode != 1 && mode != 5) { set_error("gemm_nt_bf3p: mode %d", mode); return -11; }
	if (nch < 1 || nch > 64 || (nch > 1 && (mode == 1 || !ws))) { set_error("gemm_nt_bf3p: K split %d not supported here", nch); return -14; }
	if (m <= 0 || n <= 0) return 0;
	if (m % 128 != 0 || n % P_TN != 0 || k % P_BK != 0 || k < P_BK || ldp % 32 != 0 || pstride % 8 != 0 || ((uintptr_t)pl & 15) || (tri && m != n) || arow % 16 != 0 || brow % 16 != 0 ||
	    m > (1 << 30) || n > (1 << 30) || ldc >= (1 << 24) || ((arow > brow ? arow : brow) + (m > n ? m : n)) * ldp * 2 >= ((int64_t)1 << 32)) {
		set_error("gemm_nt_bf3p: shape or alignment not supported (m=%lld n=%lld k=%lld)", (long long)m, (long long)n, (long long)k);
		return -6;
	}
	Bf3pArgs p;
	p.pl = pl; p.ldp = ldp; p.pstride = pstride; p.arow = arow; p.brow = brow; p.C = C; p.ldc = ldc;
	p.m = (int)m; p.n = (int)n; p.k = (int)k; p.tri = tri ? 1 : 0;
	p.stl = 2;
	{
		const int64_t sm = (m + 1023) / 1024, sn = (n + 1023) / 1024;
		if ((p.tri ? sm * (sm + 1) / 2 : sm * sn) < 128 && !(STPY_LAB && g_potrf_serial_band == -1)) p.stl = 1;          // fewer than 16 super-tiles per XCD: balance before locality (lab: knob 31 = -1 keeps the large ones, for the A/B)
	}
	const int64_t se = (int64_t)256 << p.stl;
	p.nst_m = (int)((m + se - 1) / se); p.nst_n = (int)((n + se - 1) / se);
	p.nsuper = p.tri ? p.nst_m * (p.nst_m + 1) / 2 : p.nst_m * p.nst_n;
	p.exp = g_gemm_exp;
	p.kchunk = nch > 1 ? (int)(((k / nch + 127) / 128) * 128) : (int)k;          // whole two-level accumulation chunks (4 K steps of 32)
	if (nch > 1 && (int64_t)p.kchunk * (nch - 1) >= k) { set_error("gemm_nt_bf3p: K split %d leaves an empty chunk (k = %lld)", nch, (long long)k); return -14; }
	p.ws = ws; p.ldws = n; p.ws_stride = m * n;
	const int64_t nblocks = (((int64_t)p.nsuper + 7) / 8) * (8 << (2 * p.stl + 1));
	if (nblocks > INT32_MAX) { set_error("gemm_nt_bf3p: grid too large"); return -2; }
	static std::atomic<bool> attr_set{false};
	if (!attr_set.load(std::memory_order_acquire)) {
		hipError_t e = hipFuncSetAttribute((const void*)gemm_bf3p_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf3p_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
		if (e == hipSuccess) e = hipFuncSetAttribute((const void*)gemm_bf3p_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
		if (e != hipSuccess) { set_error("gemm_nt_bf3p: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return -1000 - (int)e; }
		attr_set.store(true, std::memory_order_release);
	}
	const dim3 grid((unsigned)nblocks, (unsigned)nch);
	if (mode == 1) hipLaunchKernelGGL(gemm_bf3p_kernel<1>, grid, dim3(P_THREADS), P_LDS, st, p);
	else if (mode == 5) hipLaunchKernelGGL(gemm_bf3p_kernel<2>, grid, dim3(P_THREADS), P_LDS, st, p);
	else hipLaunchKernelGGL(gemm_bf3p_kernel<0>, grid, dim3(P_THREADS), P_LDS, st, p);
	return check_launch("gemm_nt_bf3p");
}

// ---- C (op) A A^T on the lower 128 x 128 tiles, fp32, A split once into the workspace (the feature-space normal equations: every one of the
// ---- n / 128 row tiles of A is an operand of n / 128 output tiles).  0 bytes = this route does not take the shape.
// Few output tiles and a long K (m = 8192 features, 65 536 rows per slab: 1056 workgroup tiles = 4.1 rounds on 256 CUs, each 2048 K steps long,
// so the fifth round runs on an eighth of the chip): the K range is cut into chunks that are dispatched as one grid (blockIdx.y), chunk 0 onto C
// and the others into their own n x n buffers, which one more launch adds to C in index order -- no workgroup waits for another, the result does
// not depend on the schedule.  The number of chunks minimises  ceil(tiles * chunks / 256) * (K / chunks)  + a charge per extra buffer.
static int64_t syrk_tiles(int64_t n)
{
	int64_t t = 0;
	for (int64_t i = 0; i < (n + 255) / 256; ++i) { const int64_t c = 2 * i + 2; t += c < n / 128 ? c : n / 128; }
	return t;
}

static int syrk_chunks(int64_t n, int64_t k)
{
	const int64_t tiles = syrk_tiles(n);
	int best = 1;
	int64_t best_cost = ((tiles + 255) / 256) * k;
	for (int c = 2; c <= 8; c *= 2) {
		const int64_t kc = ((k / c + 127) / 128) * 128;
		if (kc < 4096 || kc * (c - 1) >= k) break;
		const int64_t cost = ((tiles * c + 255) / 256) * kc + 3000 * (c - 1);
		if (cost < best_cost - best_cost / 50) { best = c; best_cost = cost; }          // (only for a gain of more than 2 %)
	}
	return best;
}

int64_t syrk_planes_workspace_bytes(int64_t n, int64_t k)
{
	if (n % 128 != 0 || n < 2048 || k % P_BK != 0 || k < 2 * P_BK || n * k * 2 >= ((int64_t)1 << 32)) return 0;
	return 3 * n * k * 2 + (int64_t)(syrk_chunks(n, k) - 1) * n * n * 4;
}

// C[i][j] += ws[0][i][j] + ws[1][i][j] + ...  on the lower 128 x 128 tiles (fixed order); thread = four consecutive columns
__global__ __launch_bounds__(256)
void syrk_chunk_sum_kernel(float* __restrict__ C, int64_t ldc, const float* __restrict__ ws, int64_t n, int nbuf)
{
	typedef float v4f __attribute__((ext_vector_type(4)));
	const int ti = blockIdx.y, tj = blockIdx.x >> 4;            // sixteen workgroups per 128 x 128 tile: 8 rows each
	if (tj > ti) return;
	const int r = ti * 128 + (blockIdx.x & 15) * 8 + (threadIdx.x >> 5), c = tj * 128 + (threadIdx.x & 31) * 4;
	v4f s = *(const v4f*)(ws + (int64_t)r * n + c);
	for (int b = 1; b < nbuf; ++b) s += *(const v4f*)(ws + (int64_t)b * n * n + (int64_t)r * n + c);
	float* const o = C + (int64_t)r * ldc + c;
	if ((((uintptr_t)o) & 15) == 0) *(v4f*)o = *(const v4f*)o + s;
	else { o[0] += s[0]; o[1] += s[1]; o[2] += s[2]; o[3] += s[3]; }
}

int syrk_planes(int64_t n, int64_t k, const float* A, int64_t lda, float* C, int64_t ldc, int mode, void* work, hipStream_t st)
{
	unsigned short* const pl = (unsigned short*)work;
	int rc = bf3_split(A, lda, n, k, pl, k, n * k, 0, st);
	if (rc) return rc;
	const int nch = mode == 1 ? 1 : syrk_chunks(n, k);
	float* const ws = (float*)((unsigned char*)work + 3 * n * k * 2);
	rc = gemm_nt_bf3p(n, n, k, pl, k, n * k, 0, 0, C, ldc, 1, st, mode, nch, nch > 1 ? ws : nullptr);
	if (rc || nch == 1) return rc;
	hipLaunchKernelGGL(syrk_chunk_sum_kernel, dim3((unsigned)(n / 128) * 16, (unsigned)(n / 128)), dim3(256), 0, st, C, ldc, (const float*)ws, n, nch - 1);
	return check_launch("syrk_planes (chunk sum)");
}

}  // namespace stpy

#if STPY_LAB
// lab build only: the two kernels alone, for tools/bf3p_bench.py
extern "C" __attribute__((visibility("default"))) int stpy_debug_bf3_split(const float* X, int64_t ldx, int64_t rows, int64_t cols, void* pl, int64_t ldp, int64_t pstride, void* stream)
{ return stpy::bf3_split(X, ldx, rows, cols, (unsigned short*)pl, ldp, pstride, 0, (hipStream_t)stream); }
extern "C" __attribute__((visibility("default"))) int stpy_debug_gemm_bf3p(int64_t m, int64_t n, int64_t k, const void* pl, int64_t ldp, int64_t pstride, int64_t arow, int64_t brow,
                                                                            float* C, int64_t ldc, int tri, void* stream)
{ return stpy::gemm_nt_bf3p(m, n, k, (const unsigned short*)pl, ldp, pstride, arow, brow, C, ldc, tri, (hipStream_t)stream, 1, 1, nullptr); }
#endif
