// Cycles per element of the table-assisted fp64 exp of the Gram epilogue (gemm.hip: gram_exp_tab), alone on the chip:
// CH independent chains per scheduling window, one or two waves per SIMD.   build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double exp_tab(double x, const double* tab)
{
	x = fmax(x, -745.0);
	const double k = rint(x * 369.3299304675746);
	double r = fma(-k, 0x1.62e42ffp-9, x);
	r = fma(-k, -1.6409824502660487e-13, r);
	double q = fma(r, 4.1666666666666664e-02, 1.6666666666666666e-01);
	q = fma(q, r, 0.5);
	q = fma(q, r, 1.0);
	q *= r;
	const int ki = (int)k;
	const double t = tab[ki & 255];
	return ldexp(fma(t, q, t), ki >> 8);
}

template <int CH>
__global__ __launch_bounds__(256) void probe(double* out, long long* cyc, int iters, double seed)
{
	__shared__ double tab[256];
	tab[threadIdx.x] = exp2((double)threadIdx.x / 256.0);
	__syncthreads();
	double acc[64];
	for (int i = 0; i < 64; ++i) acc[i] = -seed * (threadIdx.x + 1) * (i + 1) * 1e-3;
	const double na = 0.1 * threadIdx.x, nb = 0.01;
	long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 64; ++i) {
			acc[i] = 1.5 * exp_tab((acc[i] + na) + nb, tab) + 0.25;
			acc[i] = -acc[i] - 0.1 * i;
			if (i % CH == CH - 1) __builtin_amdgcn_sched_barrier(0);
		}
	}
	long long t1 = __builtin_amdgcn_s_memtime();
	double s = 0;
	for (int i = 0; i < 64; ++i) s += acc[i];
	out[blockIdx.x * 256 + threadIdx.x] = s;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	double* out; long long* cyc;
	hipMalloc(&out, 512 * 256 * 8); hipMalloc(&cyc, 512 * 8);
	const int iters = 200;
	for (int wgs : {256, 512}) {
		for (int ch : {2, 4, 8}) {
			for (int rep = 0; rep < 2; ++rep) {
				if (ch == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1.0);
				if (ch == 4) hipLaunchKernelGGL(probe<4>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1.0);
				if (ch == 8) hipLaunchKernelGGL(probe<8>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters, 1.0);
				hipDeviceSynchronize();
			}
			std::vector<long long> h(wgs);
			hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
			double s = 0; for (auto v : h) s += v;
			printf("%d waves per SIMD, %d chains per window: %7.1f ticks per element row (wave-level, per wave)\n", wgs / 256, ch, s / wgs / iters / 64);
		}
	}
	return 0;
}
