// Bare fp64 MFMA issue-rate probe: what one MI355X actually sustains on v_mfma_f64_16x16x4_f64 with
// operands in registers (no LDS, no memory) -- the practical ceiling the GEMM is measured against.
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// random-operand variant: 8 different (a, b) pairs per lane drawn from a hash, cycled through the
// accumulators -- multiplier inputs toggle like real data (DVFS check, MI355X_MICROARCH.md give-back)
template <int NACC>
__global__ __launch_bounds__(256, 2) void probe_rand(double* out, int iters, unsigned long long* clk)
{
	d4 acc[NACC];
	double a[8], b[8];
	unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
	for (int i = 0; i < 8; ++i) {
		h = h * 1664525u + 1013904223u; a[i] = ((int)(h >> 8) - (1 << 23)) * (1.0 / (1 << 23)) + 1e-9 * i;
		h = h * 1664525u + 1013904223u; b[i] = ((int)(h >> 8) - (1 << 23)) * (1.0 / (1 << 23)) * 0.01;
	}
	for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 7], b[(i + (i >> 3)) & 7], acc[i], 0, 0, 0);
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ __launch_bounds__(256, 2) void probe(double* out, int iters, unsigned long long* clk)
{
	d4 acc[NACC];
	double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-3;
	for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
	unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
	double s = 0;
	for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC> void run(int blocks, int iters, bool rnd = false)
{
	double* out; unsigned long long* clk;
	hipMalloc(&out, (size_t)blocks * 256 * 8); hipMalloc(&clk, (size_t)blocks * 16);
	hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
	if (rnd) probe_rand<NACC><<<blocks, 256>>>(out, iters, clk); else probe<NACC><<<blocks, 256>>>(out, iters, clk);
	hipDeviceSynchronize();
	hipEventRecord(e0);
	if (rnd) probe_rand<NACC><<<blocks, 256>>>(out, iters, clk); else probe<NACC><<<blocks, 256>>>(out, iters, clk);
	hipEventRecord(e1); hipEventSynchronize(e1);
	float ms; hipEventElapsedTime(&ms, e0, e1);
	unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
	double flops = (double)blocks * 4 * iters * NACC * 2048.0;
	double cyc_per_mfma = (double)h[0] / ((double)iters * NACC);
	printf("%s blocks=%d waves/SIMD=%d nacc=%d: %.2f ms  %.1f TFLOP/s  | wave0: %.1f shader-cycles per MFMA, clock %.0f MHz\n",
	       rnd ? "random  " : "constant", blocks, blocks / 256, NACC, ms, flops / ms / 1e9, cyc_per_mfma, (double)h[0] / (double)h[1] * 100.0);
	hipFree(out); hipFree(clk);
}

int main()
{
	run<16>(256, 20000);
	run<16>(512, 20000);
	run<4>(256, 80000);
	run<1>(256, 200000);
	run<16>(256, 200000);
	run<16>(256, 200000, true);
	run<16>(512, 100000, true);
	run<16>(256, 1000000, true);
	return 0;
}
