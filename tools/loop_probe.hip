// Structure probe for the GEMM main loop (no global memory): per "K tile" a wave does what gemm_nt_kernel
// does -- 2 x [8 ds_read_b128 fragments -> 32 v_mfma_f64_16x16x4] -- on LDS contents that never change.
// Variants isolate what keeps the matrix pipes from 100 %: the LDS reads, the workgroup barrier, the
// number of resident waves.  build+run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/loop_probe.hip -o /tmp/loop_probe && /tmp/loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// MODE bit 0: read fragments from LDS every half tile (else once, before the loop)
// MODE bit 1: __syncthreads() per K tile
// MODE bit 2: fragments of BOTH halves read up front (register-pipelined form)
// MODE bit 3: s_setprio 1 around the MFMA block
template <int MODE>
__global__ __launch_bounds__(256, 2) void loop_probe(double* out, int ktiles, const double* src, long src_elems)
{
	__shared__ __attribute__((aligned(16))) double smem[2 * 256 * 16];      // [2 buffers][A 128 rows + B 128 rows][16]
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	for (int i = tid; i < 2 * 256 * 16; i += 256) smem[i] = 1e-3 * ((i * 2654435761u >> 20) & 1023) - 0.5;
	__syncthreads();
	const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, g = lane >> 4;
	d4 acc[4][4];
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = d4{0, 0, 0, 0};
	const int fsw = (((r16 >> 1) & 3) << 1) | ((r16 >> 3) & 1);
	d2 fa[2][4], fb[2][4];
	auto rd = [&](int buf, int h) {
		const double* as = smem + (buf * 256 + wm * 64 + r16) * 16;
		const double* bs = smem + (buf * 256 + 128 + wn * 64 + r16) * 16;
		const int hoff = ((2 * g + h) ^ fsw) * 2;
#pragma unroll
		for (int t = 0; t < 4; ++t) { fa[h][t] = *(const d2*)(as + t * 16 * 16 + hoff); fb[h][t] = *(const d2*)(bs + t * 16 * 16 + hoff); }
	};
	auto mm = [&](int h) {
		if (MODE & 8) __builtin_amdgcn_s_setprio(1);
#pragma unroll
		for (int s = 0; s < 2; ++s)
#pragma unroll
			for (int tm = 0; tm < 4; ++tm)
#pragma unroll
				for (int tn = 0; tn < 4; ++tn)
					acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[h][tm][s], fb[h][tn][s], acc[tm][tn], 0, 0, 0);
		if (MODE & 8) __builtin_amdgcn_s_setprio(0);
	};
	// LDS-DMA of the "next tile": wave w moves rows [32w, 32w+32) of A and of B, 8 rows (1 KiB) per instruction
	const double* gsrc = src + (long)blockIdx.x * 4096 + (long)lane * 2 + wave * 512;
	const long tile_stride = (MODE & 64) ? 512L * 4096 : 0;      // every workgroup walks its own stream / or re-reads 32 KiB
	auto dma_one = [&](const double* g, double* l) {
		const unsigned laddr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) double*)l;
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
		             : "=&s"(keep) : "v"(g), "s"(laddr) : "memory");
	};
	long goff = 0;
	auto dma_tile = [&](int b) {
		const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			dma_one(gsrc + goff + i * 128, smem + (b * 256 + wv * 32 + i * 8) * 16);
			dma_one(gsrc + goff + 2048 + i * 128, smem + (b * 256 + 128 + wv * 32 + i * 8) * 16);
		}
		goff += tile_stride;
		if (goff + 512L * 4096 + 4096 > src_elems) goff = 0;
	};
	rd(0, 0); rd(0, 1);
	int buf = 0;
	if (MODE & 16) {
		// mid-tile barrier pipeline: F1 (second half of this tile) is read under the MFMAs of F0, the barrier
		// sits between the halves, F0 of the NEXT tile is read under the MFMAs of F1
		for (int kt = 0; kt < ktiles; ++kt) {
			rd(buf, 1);
			mm(0);
			__builtin_amdgcn_sched_barrier(0);
			if (MODE & 2) __syncthreads();
			rd(buf ^ 1, 0);
			mm(1);
			__builtin_amdgcn_sched_barrier(0);
			buf ^= 1;
		}
	} else
	for (int kt = 0; kt < ktiles; ++kt) {
		if (MODE & 32) dma_tile(buf ^ 1);
		if (MODE & 4) {
			if (MODE & 1) { rd(buf, 0); rd(buf, 1); }
			mm(0); mm(1);
		} else {
			if (MODE & 1) rd(buf, 0);
			mm(0);
			__builtin_amdgcn_sched_barrier(0);
			if (MODE & 1) rd(buf, 1);
			mm(1);
		}
		if (MODE & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		if (MODE & 2) __syncthreads();
		buf ^= 1;
	}
	double s = 0;
	for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
	out[blockIdx.x * 256 + tid] = s;
}

template <int MODE> void run(const char* what, int blocks, int ktiles)
{
	double* out; hipMalloc(&out, (size_t)blocks * 256 * 8);
	static double* src = nullptr;
	const long src_elems = 1L << 30;                           // 8 GiB stream
	if (!src) { hipMalloc(&src, src_elems * 8); hipMemset(src, 0, src_elems * 8); }
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	loop_probe<MODE><<<blocks, 256>>>(out, ktiles, src, src_elems);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	loop_probe<MODE><<<blocks, 256>>>(out, ktiles, src, src_elems);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	const double flops = (double)blocks * 4 * ktiles * 64 * 2048.0;
	printf("%-58s blocks=%4d (%d per CU): %8.2f ms  %6.2f TFLOP/s\n", what, blocks, blocks / 256, ms, flops / ms / 1e9);
	hipFree(out);
}

int main()
{
	const int KT = 20000;
	for (int blocks = 256; blocks <= 512; blocks += 256) {
		run<0>("MFMA only (fragments read once)", blocks, KT);
		run<1>("+ LDS fragment reads per half tile", blocks, KT);
		run<3>("+ LDS reads + barrier per K tile", blocks, KT);
		run<2>("barrier only", blocks, KT);
		run<5>("reads of both halves up front", blocks, KT);
		run<7>("reads up front + barrier", blocks, KT);
		run<9>("LDS reads, s_setprio around MFMAs", blocks, KT);
		run<11>("LDS reads + barrier, s_setprio", blocks, KT);
		run<35>("reads + barrier + DMA re-reading 32 KiB (cache hits)", blocks, KT);
		run<99>("reads + barrier + DMA streaming memory", blocks, KT);
		run<32>("MFMA + DMA (cache hits) only", blocks, KT);
		run<17>("mid-tile pipeline, no barrier", blocks, KT);
		run<19>("mid-tile pipeline + barrier", blocks, KT);
		run<27>("mid-tile pipeline + barrier + s_setprio", blocks, KT);
	}
	return 0;
}
