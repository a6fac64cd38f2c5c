// Issue cost of fp64 VALU instructions on gfx950: one wave per SIMD (256-thread workgroup, one per CU), N independent
// instructions of one kind in a loop, cycles per instruction from s_memtime.   build: hipcc --offload-arch=gfx950 -O3 -o fp64_valu_probe fp64_valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
template <int OP>
__global__ __launch_bounds__(256) void probe(double* out, long long* cyc, int iters, double seed)
{
	double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
	const double c = 1.0000001, d = 1e-9;
	int e0 = threadIdx.x & 3;
	long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#define ONE(a) \
		if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(c), "v"(d)); \
		if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(c)); \
		if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(d)); \
		if (OP == 3) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(e0)); \
		if (OP == 4) asm volatile("v_rndne_f64 %0, %0" : "+v"(a)); \
		if (OP == 5) { int k; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(k) : "v"(a)); e0 ^= k; } \
		if (OP == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(d)); \
		if (OP == 7) { float f = (float)e0; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); e0 += (int)f; } \
		if (OP == 8) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(e0) : "v"(e0)); \
		if (OP == 9) asm volatile("v_rsq_f64 %0, %0" : "+v"(a)); \
		if (OP == 10) asm volatile("v_sqrt_f64 %0, %0" : "+v"(a)); \
		if (OP == 11) asm volatile("v_rcp_f64 %0, %0" : "+v"(a));
		REP8(ONE(a0) ONE(a1) ONE(a2) ONE(a3) ONE(a4) ONE(a5) ONE(a6) ONE(a7))
	}
	long long t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
	const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_ldexp_f64", "v_rndne_f64", "v_cvt_i32_f64 (+v_xor)", "v_max_f64", "cvt+v_fma_f32+cvt+add", "v_lshl_add_u32", "v_rsq_f64", "v_sqrt_f64", "v_rcp_f64"};
	double* out; long long* cyc;
	hipMalloc(&out, 256 * 256 * 8); hipMalloc(&cyc, 256 * 8);
	const int iters = 2000;
	for (int op = 0; op < 12; ++op) {
		for (int rep = 0; rep < 2; ++rep) {
			switch (op) {
#define L(o) case o: hipLaunchKernelGGL(probe<o>, dim3(256), dim3(256), 0, 0, out, cyc, iters, 1.0); break;
				L(0) L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11)
			}
			hipDeviceSynchronize();
		}
		std::vector<long long> h(256);
		hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
		double s = 0; for (auto v : h) s += v;
		// s_memtime counts at 100 MHz (constant) on this part: report both raw ticks and ns
		printf("%-26s %8.3f memtime ticks per instruction (64 per iteration, one wave per SIMD)\n", names[op], s / 256 / iters / 64);
	}
	return 0;
}
