// How fast does a dependent VALU chain (the diagonal-block kernel's pivot loop) run on a SIMD that two other waves keep
// saturated with back-to-back fp64 MFMAs (the trailing update)?  And does s_setprio change it?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_beside_mfma_probe.hip -o /tmp/vbm && /tmp/vbm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4 __attribute__((ext_vector_type(4)));

// PRIO 0: MFMAs only.  PRIO 1: plus the LDS traffic of the real update's K loop (16 ds_read_b128 per 64 MFMAs per wave).
template <int PRIO>
__global__ __launch_bounds__(256) void mfma_flood(double* sink, int iters)
{
	extern __shared__ float lds[];
	typedef double d2 __attribute__((ext_vector_type(2)));
	asm volatile("v_mov_b32 v223, 0" ::: "v223");          // 224 VGPRs like the trailing-update kernel
	v4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
	double x = 1.0 + threadIdx.x * 1e-9, y = 1.0 - threadIdx.x * 1e-9;
	const d2* l2 = (const d2*)lds + threadIdx.x;
	for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1.f;
	__syncthreads();
	for (int i = 0; i < iters; ++i) {
		if (PRIO == 1 && (i & 3) == 0) {          // every 16 MFMAs: four 16-byte reads (= 16 per 64 MFMAs)
			d2 r0 = l2[0], r1 = l2[256], r2 = l2[512], r3 = l2[768];
			x += (r0[0] + r1[1] + r2[0] + r3[1]) * 1e-30;
		}
		a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
		a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
		a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
		a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
	}
	if (a0[0] + a1[1] + a2[2] + a3[3] == 123.456) sink[0] = 1;
}

// kind 0: dependent fp64 FMA chain; 1: dependent fp32 FMA chain; 2: dependent lane permutes (ds_bpermute); 3: dependent LDS round trips
template <int PRIO, int KIND>
__global__ __launch_bounds__(256) void chain(double* sink, int n, unsigned long long* ticks)
{
	extern __shared__ float lds[];
	asm volatile("v_mov_b32 v63, 0" ::: "v63");
	if (PRIO >= 0) __builtin_amdgcn_s_setprio(PRIO);
	lds[threadIdx.x] = (float)threadIdx.x;
	__syncthreads();
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	double acc = threadIdx.x * 1e-3;
	float accf = threadIdx.x * 1e-3f;
	int idx = threadIdx.x;
	if (threadIdx.x < 64) {
		for (int i = 0; i < n; ++i) {
			if (KIND == 0) acc = __builtin_fma(acc, 1.0000001, 1e-9);
			if (KIND == 1) accf = __builtin_fmaf(accf, 1.0000001f, 1e-9f);
			if (KIND == 2) idx = __builtin_amdgcn_ds_bpermute(((idx + 1) & 63) << 2, idx);
			if (KIND == 3) { lds[(idx & 63) + 256] = accf; __builtin_amdgcn_s_waitcnt(0xC07F); accf = lds[((idx + 1) & 63) + 256] + 1.f; __builtin_amdgcn_s_waitcnt(0xC07F); }
		}
	}
	const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0) ticks[0] = t1 - t0;
	if (acc + accf + idx == 123.456) sink[0] = 1;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PRIO, int KIND>
int run(const char* name, int n, hipStream_t mainst, hipStream_t side, double* sink, unsigned long long* ticks, int flood_prio)
{
	for (int beside = 0; beside < 2; ++beside) {
		double us = 0;
		for (int r = 0; r < 4; ++r) {
			if (beside) {
				if (flood_prio == 0) hipLaunchKernelGGL((mfma_flood<0>), dim3(4096), dim3(256), 32 * 1024, mainst, sink, 2048);
				else hipLaunchKernelGGL((mfma_flood<1>), dim3(4096), dim3(256), 32 * 1024, mainst, sink, 2048);
				for (volatile int w = 0; w < 200000; ++w) {}
			}
			hipLaunchKernelGGL((chain<PRIO, KIND>), dim3(1), dim3(256), 8 * 1024, side, sink, n, ticks);
			CHK(hipDeviceSynchronize());
			unsigned long long t; CHK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost));
			us += t / 100.0;
		}
		printf("%-34s prio %2d  %s: %8.1f us\n", name, PRIO, beside ? (flood_prio == 0 ? "beside an fp64-MFMA flood           " : "beside an fp64-MFMA + LDS-read flood") : "alone                               ", us / 4);
	}
	return 0;
}

int main()
{
	double* sink; CHK(hipMalloc(&sink, 64));
	unsigned long long* ticks; CHK(hipMalloc(&ticks, 64));
	int lo, hi; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t mainst, side; CHK(hipStreamCreateWithPriority(&mainst, hipStreamNonBlocking, lo)); CHK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
	
	// how long does one flood workgroup take?  2048 x 4 MFMAs x 64 cycles = 524k cycles ~ 220 us alone on its SIMD, ~440 us with two per SIMD
	for (int fl = 0; fl < 2; ++fl) {
		if (run<-1, 0>("fp64 FMA chain (8000 deep)", 8000, mainst, side, sink, ticks, fl)) return 1;
		if (run<-1, 2>("lane permute chain (2000 deep)", 2000, mainst, side, sink, ticks, fl)) return 1;
		if (run<3, 2>("lane permute chain (2000 deep)", 2000, mainst, side, sink, ticks, fl)) return 1;
		if (run<-1, 3>("LDS round-trip chain (1000 deep)", 1000, mainst, side, sink, ticks, fl)) return 1;
	}
	return 0;
}
