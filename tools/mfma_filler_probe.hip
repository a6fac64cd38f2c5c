// How many VALU instructions hide in the issue shadow of an MFMA on gfx950, per MFMA kind?  One wave per SIMD (256-thread workgroup,
// one per CU); a loop of 16 independent MFMAs (16 accumulators) with F filler instructions after each; s_memtime ticks per MFMA.
// build: hipcc --offload-arch=gfx950 -O3 -o _ab/mfma_filler_probe mfma_filler_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v4d __attribute__((ext_vector_type(4)));

// KIND 0: v_mfma_f32_16x16x4_f32 + v_fma_f32 fillers; 1: same MFMA + v_cos_f32 fillers; 2: v_mfma_f64_16x16x4_f64 + v_fma_f64 fillers;
// 3: f64 MFMA + v_fma_f32 fillers
template <int KIND, int F>
__global__ __launch_bounds__(256) void probe(float* out, long long* cyc, int iters)
{
	v4f af[16]; v4d ad[16];
	for (int i = 0; i < 16; ++i) { af[i] = v4f{0, 0, 0, 0}; ad[i] = v4d{0, 0, 0, 0}; }
	float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
	double da = a, db = b;
	float f[8]; double g[8];
	for (int i = 0; i < 8; ++i) { f[i] = a + i; g[i] = a + i; }
	long long t0 = __builtin_amdgcn_s_memtime();
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			if (KIND <= 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(af[i]) : "v"(a), "v"(b));
			else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ad[i]) : "v"(da), "v"(db));
#pragma unroll
			for (int k = 0; k < F; ++k) {
				const int s = (i * F + k) & 7;
				if (KIND == 0 || KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[s]) : "v"(b));
				if (KIND == 1) asm volatile("v_cos_f32 %0, %0" : "+v"(f[s]));
				if (KIND == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(g[s]) : "v"(db));
			}
		}
	}
	long long t1 = __builtin_amdgcn_s_memtime();
	float acc = 0;
	for (int i = 0; i < 16; ++i) acc += af[i][0] + (float)ad[i][0];
	for (int i = 0; i < 8; ++i) acc += f[i] + (float)g[i];
	out[blockIdx.x * 256 + threadIdx.x] = acc;
	if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int F> static void run(float* out, long long* cyc, const char* what)
{
	const int iters = 1000;
	for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((probe<KIND, F>), dim3(256), dim3(256), 0, 0, out, cyc, iters); hipDeviceSynchronize(); }
	std::vector<long long> h(256);
	hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
	double s = 0; for (auto v : h) s += v;
	printf("%-44s %d fillers per MFMA: %7.2f ticks per MFMA\n", what, F, s / 256 / iters / 16);
}

int main()
{
	float* out; long long* cyc;
	hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
#define ROW(K, W) run<K, 0>(out, cyc, W); run<K, 1>(out, cyc, W); run<K, 2>(out, cyc, W); run<K, 3>(out, cyc, W); run<K, 4>(out, cyc, W); run<K, 6>(out, cyc, W); run<K, 8>(out, cyc, W); run<K, 12>(out, cyc, W);
	ROW(0, "v_mfma_f32_16x16x4_f32 + v_fma_f32")
	ROW(1, "v_mfma_f32_16x16x4_f32 + v_cos_f32")
	ROW(2, "v_mfma_f64_16x16x4_f64 + v_fma_f64")
	ROW(3, "v_mfma_f64_16x16x4_f64 + v_fma_f32")
	return 0;
}
