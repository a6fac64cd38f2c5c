// Prototype: fp64 NT GEMM C (-)= A B^T, 128 x 128 tile, two workgroups per CU, the A operand loaded DIRECT TO VGPR
// (no LDS), the B operand through LDS-DMA -- the structure the vendor library's kernel for these shapes uses
// (DTVA1, MT128x128x16).  Standalone: checks itself against a host loop on a small
// problem, then times the trailing-update shape.
//   hipcc --offload-arch=gfx950 -O3 tools/dtv_probe.hip -o /tmp/dtv && /tmp/dtv
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <type_traits>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TM = 128, TN = 128, BK = 16;

struct Args { const double* A; const double* B; double* C; long lda, ldb, ldc; int m, n, k, sub, tiles_m, tiles_n, nst_n, nsuper; };

// A: direct to VGPR.  Lane (r16, g) of wave (wm, wn) owns, for each of its four 16-row tiles, the 32 bytes
// A[row][k0 + 4g .. 4g+3] of a K tile -- exactly the two d2 fragments (halves h = 0, 1) its MFMAs consume -- and
// refills each fragment IN PLACE for the next K tile right after the last MFMA that reads it has been issued.
// B: LDS-DMA into two swizzled stages as in gemm_nt_kernel (half the LDS traffic, half the DMA instructions).
template <bool SUB, int VAR = 0>
__global__ __launch_bounds__(256, 2) void dtv_gemm(Args p)
{
	__shared__ __attribute__((aligned(16))) double smem[2 * TN * BK];      // 32 KiB
	const int b = blockIdx.x;
	const int S = (b & 7) + 8 * (b >> 9);
	const int w = (b >> 3) & 63;
	if (S >= p.nsuper) return;
	const int si = S / p.nst_n, sj = S - si * p.nst_n;
	const int ti = __builtin_amdgcn_readfirstlane(si * 8 + (w >> 3));
	const int tj = __builtin_amdgcn_readfirstlane(sj * 8 + (w & 7));
	if (ti >= p.tiles_m || tj >= p.tiles_n) return;
	const int row0 = ti * TM, col0 = tj * TN;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;

	// ---- B through LDS-DMA: wave w moves rows [32w, 32w+32), 8 rows (1 KiB) per piece
	const double* dsrc[4];
#pragma unroll
	for (int i = 0; i < 4; ++i) {
		const int r = wave * 32 + i * 8 + (lane >> 3);
		const int f = (((r >> 1) & 3) << 1) | ((r >> 3) & 1);
		dsrc[i] = p.B + (long)(col0 + r) * p.ldb + ((lane & 7) ^ f) * 2;
	}
	auto dma_one = [&](const double* gsrc, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(gsrc), "s"(laddr) : "memory");
	};
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)smem;
	auto dma_tile = [&](int buf, int k0) {
		const unsigned base = lds0 + (unsigned)(buf * TN + wave * 32) * (BK * 8);
#pragma unroll
		for (int i = 0; i < 4; ++i) dma_one(dsrc[i] + k0, base + i * 8 * BK * 8);
	};

	// ---- A straight into registers: uniform row-tile base + one 32-bit lane offset
	const double* const abase = p.A + (long)(row0 + wm * 64) * p.lda;
	const unsigned alane = (unsigned)r16 * (unsigned)p.lda + (unsigned)g * 4;       // elements
	d2 fa[4][2];
	// asm loads: hipcc's own vmcnt bookkeeping cannot see the LDS-DMA pieces, so the waits are placed by hand.  A wait is
	// tied to the fragment it guards through a "+v" operand, which keeps the MFMAs that read it behind the wait.
	auto lda_frag = [&](int tm, int h, int k0) {
		const double* ptr = abase + (long)tm * 16 * p.lda + (alane + (unsigned)k0 + (unsigned)h * 2);
		asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(fa[tm][h]) : "v"(ptr) : "memory");
	};
	// steady state, in issue order per tile: 4 DMA pieces, 4 first-half refills, 4 second-half refills.  When block
	// (h, tm) starts, exactly 11 younger operations than the load of fa[tm][h] may still be in flight.
	auto wait_frag = [&](int tm, int h) { asm volatile("s_waitcnt vmcnt(11)" : "+v"(fa[tm][h]) :: "memory"); };
	const int KT = p.k / BK;
	// prologue: tile 0's B pieces and A fragments first, the C tile behind them (so its latency overlaps theirs), then ONE
	// wait for everything that hipcc can see (a builtin, not asm): otherwise it parks its vmcnt waits for the C loads at their
	// first use -- inside the loop, where they would drain the hand-counted queue every iteration
	dma_tile(0, 0);
#pragma unroll
	for (int tm = 0; tm < 4; ++tm) lda_frag(tm, 0, 0);
#pragma unroll
	for (int tm = 0; tm < 4; ++tm) lda_frag(tm, 1, 0);
	d4 acc[4][4];
	double* const ctile = p.C + (long)row0 * p.ldc + col0;
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const double* crow = ctile + (long)(wm * 64 + tm * 16 + g + 4 * i) * p.ldc + wn * 64 + r16;
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = SUB ? crow[tn * 16] : 0.0;
		}
	__builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
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
			const double* bs = smem + buf * TN * BK + boff + ((2 * g + h) ^ fsw) * 2;
#pragma unroll
			for (int t = 0; t < 4; ++t) { fb[t] = *(const d2*)(bs + t * 16 * BK); if (SUB) fb[t] = -fb[t]; }      // C - A B^T: negate the operand that is waited for anyway
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
				for (int s = 0; s < 2; ++s)
#pragma unroll
					for (int tn = 0; tn < 4; ++tn)
						acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[tm][h][s], fb[tn][s], acc[tm][tn], 0, 0, 0);
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
#pragma unroll
	for (int tm = 0; tm < 4; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			double* crow = ctile + (long)(wm * 64 + tm * 16 + g + 4 * i) * p.ldc + wn * 64 + r16;
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) crow[tn * 16] = acc[tm][tn][i];
		}
}

__global__ void fill_kernel(double* x, size_t n, unsigned seed)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) { unsigned h = (unsigned)(i * 2654435761u) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; x[i] = (int)(h >> 8) * (1.0 / (1 << 24)) - 0.5; }
}

static int g_var = 0;
static void launch(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int m, int n, int k, int sub)
{
	Args p{A, B, C, lda, ldb, ldc, m, n, k, sub, m / TM, n / TN, 0, 0};
	const int nst_m = (p.tiles_m + 7) / 8;
	p.nst_n = (p.tiles_n + 7) / 8;
	p.nsuper = nst_m * p.nst_n;
	const int blocks = ((p.nsuper + 7) / 8) * 512;
	if (!sub) hipLaunchKernelGGL((dtv_gemm<false, 0>), dim3(blocks), dim3(256), 0, 0, p);
	else if (g_var == 0) hipLaunchKernelGGL((dtv_gemm<true, 0>), dim3(blocks), dim3(256), 0, 0, p);
	else if (g_var == 1) hipLaunchKernelGGL((dtv_gemm<true, 1>), dim3(blocks), dim3(256), 0, 0, p);
	else if (g_var == 2) hipLaunchKernelGGL((dtv_gemm<true, 2>), dim3(blocks), dim3(256), 0, 0, p);
	else hipLaunchKernelGGL((dtv_gemm<true, 3>), dim3(blocks), dim3(256), 0, 0, p);
}

int main()
{
	// ---- correctness on a small problem
	{
		const int m = 512, n = 384, k = 112;
		std::vector<double> A((size_t)m * k), B((size_t)n * k), C((size_t)m * n), R;
		srand(1);
		for (auto& v : A) v = rand() / (double)RAND_MAX - 0.5;
		for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
		for (auto& v : C) v = rand() / (double)RAND_MAX - 0.5;
		for (int sub = 0; sub < 2; ++sub) {
			R = C;
			for (int i = 0; i < m; ++i) for (int j = 0; j < n; ++j) {
				double s = 0; for (int q = 0; q < k; ++q) s += A[(size_t)i * k + q] * B[(size_t)j * k + q];
				R[(size_t)i * n + j] = sub ? C[(size_t)i * n + j] - s : s;
			}
			double *dA, *dB, *dC;
			hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8);
			hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
			hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
			launch(dA, k, dB, k, dC, n, m, n, k, sub);
			std::vector<double> O(C.size());
			hipError_t e = hipMemcpy(O.data(), dC, C.size() * 8, hipMemcpyDeviceToHost);
			double err = 0, nrm = 0;
			for (size_t i = 0; i < O.size(); ++i) { err = fmax(err, fabs(O[i] - R[i])); nrm = fmax(nrm, fabs(R[i])); }
			printf("check sub=%d: max abs err %.3e (max |ref| %.3f)  %s  [%s]\n", sub, err, nrm, err < 1e-12 * k ? "OK" : "WRONG", hipGetErrorString(e));
			hipFree(dA); hipFree(dB); hipFree(dC);
		}
	}
	// ---- timing on the trailing-update shape (full rectangle)
	const int shapes[][3] = {{32768, 32768, 1024}, {32768, 32768, 4096}};
	for (int var = 0; var < 4; ++var)
	for (auto& sh : shapes) {
		g_var = var;
		const int m = sh[0], n = sh[1], k = sh[2];
		double *dA, *dB, *dC;
		hipMalloc(&dA, (size_t)m * k * 8); hipMalloc(&dB, (size_t)n * k * 8); hipMalloc(&dC, (size_t)m * n * 8);
		fill_kernel<<<4096, 256>>>(dA, (size_t)m * k, 1u); fill_kernel<<<4096, 256>>>(dB, (size_t)n * k, 2u); fill_kernel<<<4096, 256>>>(dC, (size_t)m * n, 3u);
		hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
		launch(dA, k, dB, k, dC, n, m, n, k, 1);
		hipDeviceSynchronize();
		float best = 1e30f;
		for (int r = 0; r < 3; ++r) {
			hipEventRecord(e0);
			launch(dA, k, dB, k, dC, n, m, n, k, 1);
			hipEventRecord(e1); hipEventSynchronize(e1);
			float ms; hipEventElapsedTime(&ms, e0, e1);
			best = fminf(best, ms);
		}
		printf("dtv var=%d m=%d n=%d k=%d: %.3f ms  %.2f TFLOP/s\n", g_var, m, n, k, best, 2.0 * m * n * k / best / 1e9);
		hipFree(dA); hipFree(dB); hipFree(dC);
	}
	return 0;
}
