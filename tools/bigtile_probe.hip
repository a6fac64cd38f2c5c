// Prototype of a one-workgroup-per-CU fp64 NT GEMM for gfx950: C (-)= A B^T with a 256 x 128 tile,
// four waves of 128 x 64 (256 accumulator registers each), three LDS stages filled by LDS-DMA two
// and a half tiles ahead, fragment reads software-pipelined inside the wave (the barrier sits
// between the two halves of a K tile).  Standalone: checks itself against a host loop on a small
// problem, then times the trailing-update shape.
//   hipcc --offload-arch=gfx950 -O3 tools/bigtile_probe.hip -o /tmp/bigtile && /tmp/bigtile
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

constexpr int TM = 256, TN = 128, BK = 16, STAGES = 3, ROWS = TM + TN;

struct Args { const double* A; const double* B; double* C; long lda, ldb, ldc; int m, n, k, sub, tiles_m, tiles_n, nst_n, nsuper; };

template <bool SUB>
__global__ __launch_bounds__(256, 1) void big_gemm(Args p)
{
	__shared__ __attribute__((aligned(16))) double smem[STAGES * ROWS * BK];      // 144 KiB
	// ---- block -> tile: the 32 workgroups an XCD holds (b, b+8, ...) form a 4 x 8 super-tile (1024 x 1024)
	const int b = blockIdx.x;
	const int S = (b & 7) + 8 * (b >> 8);
	const int w = (b >> 3) & 31;
	if (S >= p.nsuper) return;
	const int si = S / p.nst_n, sj = S - si * p.nst_n;
	const int ti = __builtin_amdgcn_readfirstlane(si * 4 + (w >> 3));
	const int tj = __builtin_amdgcn_readfirstlane(sj * 8 + (w & 7));
	if (ti >= p.tiles_m || tj >= p.tiles_n) return;
	const int row0 = ti * TM, col0 = tj * TN;
	const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;

	// ---- LDS-DMA: wave w moves stage rows [96w, 96w + 96): A rows 0..255, then B rows 0..127; 8 rows (1 KiB) per piece
	const double* dsrc[12];
#pragma unroll
	for (int i = 0; i < 12; ++i) {
		const int sr = wave * 96 + i * 8 + (lane >> 3);          // stage row of this lane
		const bool isA = sr < TM;
		const int r = isA ? sr : sr - TM;                        // row inside the operand tile
		const int f = (((r >> 1) & 3) << 1) | ((r >> 3) & 1);
		const int c = (lane & 7) ^ f;                            // source chunk that lands at physical chunk lane & 7
		dsrc[i] = isA ? p.A + (long)(row0 + r) * p.lda + c * 2 : p.B + (long)(col0 + r) * p.ldb + c * 2;
	}
	auto dma_one = [&](const double* gsrc, unsigned laddr) {
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(gsrc), "s"(laddr) : "memory");
	};
	const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)smem;
	auto dma_tile = [&](int stage, int k0) {
		const unsigned base = lds0 + (unsigned)(stage * ROWS + wave * 96) * (BK * 8);
#pragma unroll
		for (int i = 0; i < 12; ++i) dma_one(dsrc[i] + k0, base + i * 8 * BK * 8);
	};

	const int KT = p.k / BK;
	auto ktile = [&](int t) { return (t < KT ? t : KT - 1) * BK; };      // past the end: re-load the last tile into a dead stage
	// ---- accumulators: the C tile itself when subtracting (loaded straight into the accumulator registers; the A
	// ---- fragments are negated after each LDS read, so the product comes out as C - A B^T and the epilogue is store-only).
	// ---- Negating C instead would stage all 256 values through VGPRs and spill the DMA pointers; hipcc's reloads then
	// ---- put vmcnt waits on them INSIDE the loop, which drain the LDS-DMA queue every tile.
	d4 acc[8][4];
	double* const ctile = p.C + (long)row0 * p.ldc + col0;
#pragma unroll
	for (int tm = 0; tm < 8; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const double* crow = ctile + (long)(wm * 128 + tm * 16 + g + 4 * i) * p.ldc + wn * 64 + r16;
#pragma unroll
			for (int tn = 0; tn < 4; ++tn) acc[tm][tn][i] = SUB ? crow[tn * 16] : 0.0;
		}

	dma_tile(0, ktile(0));
	dma_tile(1, ktile(1));
	dma_tile(2, ktile(2));

	// ---- fragment reads: lane group g takes chunks 2g (first half) and 2g+1 (second half) of its rows
	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);
	const int aoff = (wm * 128 + r16) * BK, boff = (TM + wn * 64 + r16) * BK;
	d2 fa0[8], fb0[4], fa1[8], fb1[4];
	auto rd0 = [&](int stage) {
		const double* base = smem + stage * ROWS * BK + ((2 * g + 0) ^ fsw) * 2;
#pragma unroll
		for (int t = 0; t < 8; ++t) { fa0[t] = *(const d2*)(base + aoff + t * 16 * BK); if (SUB) fa0[t] = -fa0[t]; }
#pragma unroll
		for (int t = 0; t < 4; ++t) fb0[t] = *(const d2*)(base + boff + t * 16 * BK);
	};
	auto rd1 = [&](int stage) {
		const double* base = smem + stage * ROWS * BK + ((2 * g + 1) ^ fsw) * 2;
#pragma unroll
		for (int t = 0; t < 8; ++t) { fa1[t] = *(const d2*)(base + aoff + t * 16 * BK); if (SUB) fa1[t] = -fa1[t]; }
#pragma unroll
		for (int t = 0; t < 4; ++t) fb1[t] = *(const d2*)(base + boff + t * 16 * BK);
	};

	__builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): C tile and the first three operand tiles have landed
	__syncthreads();
	rd0(0);
	int st = 0;                                            // stage of the tile being multiplied
	for (int kt = 0; kt < KT; ++kt) {
		const int st1 = (st == 2) ? 0 : st + 1;
		rd1(st);
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int tm = 0; tm < 8; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0[tm][s], fb0[tn][s], acc[tm][tn], 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		// mid tile: this wave's pieces of tile kt+1 have landed (tile kt+2's twelve may still fly); every wave has
		// finished reading stage st, so it can take tile kt+3
		__builtin_amdgcn_s_waitcnt(0x007C);                  // vmcnt(12) lgkmcnt(0)
		__syncthreads();
		dma_tile(st, ktile(kt + 3));
		rd0(st1);
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int tm = 0; tm < 8; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1[tm][s], fb1[tn][s], acc[tm][tn], 0, 0, 0);
		__builtin_amdgcn_sched_barrier(0);
		st = st1;
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // nothing may still be writing LDS when the workgroup retires

#pragma unroll
	for (int tm = 0; tm < 8; ++tm)
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			double* crow = ctile + (long)(wm * 128 + tm * 16 + g + 4 * i) * p.ldc + wn * 64 + r16;
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

static void launch(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int m, int n, int k, int sub)
{
	Args p{A, B, C, lda, ldb, ldc, m, n, k, sub, m / TM, n / TN, 0, 0};
	const int nst_m = (p.tiles_m + 3) / 4;
	p.nst_n = (p.tiles_n + 7) / 8;
	p.nsuper = nst_m * p.nst_n;
	const int blocks = ((p.nsuper + 7) / 8) * 256;
	if (sub) hipLaunchKernelGGL(big_gemm<true>, dim3(blocks), dim3(256), 0, 0, p);
	else hipLaunchKernelGGL(big_gemm<false>, dim3(blocks), dim3(256), 0, 0, p);
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
	const int shapes[][3] = {{32768, 32768, 1024}, {32768, 32768, 4096}, {32768, 1024, 32768}, {16384, 16384, 512}};
	for (auto& sh : shapes) {
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
		printf("big tile m=%d n=%d k=%d: %.3f ms  %.2f TFLOP/s\n", m, n, k, best, 2.0 * m * n * k / best / 1e9);
		hipFree(dA); hipFree(dB); hipFree(dC);
	}
	return 0;
}
