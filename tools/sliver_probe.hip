// Does a small workgroup that FITS INTO THE RESOURCES TWO TRAILING-UPDATE WORKGROUPS LEAVE OVER ON A CU start at once
// while the update floods the chip, where a workgroup that needs one update workgroup to exit waits?
//   hipcc --offload-arch=gfx950 -O3 tools/sliver_probe.hip -o /tmp/sliver && /tmp/sliver
// "update": 256 threads, 224 VGPRs (forced), 32 KiB LDS, ~70 us per workgroup, 7168 workgroups (fourteen rounds of 512)
// probes on a high-priority stream, issued 150 us after the update started, each timed from enqueue to completion:
//   fat    512 threads, 128 VGPRs, 83 KiB LDS   (the diagonal-block kernel's shape today)
//   sliver 256 threads,  64 VGPRs, 80 KiB LDS   (fits beside two update workgroups: 512 - 2*224 = 64 VGPRs, 160 - 64 = 96 KiB)
//   gemmy  256 threads, 224 VGPRs, 32 KiB LDS, 100 workgroups (a chain GEMM today)
//   slivergemm 256 threads, 64 VGPRs, 48 KiB LDS, 750 workgroups of ~5 us
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>

__device__ __forceinline__ void spin_us(double us)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
	while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)(us * 100.0)) __builtin_amdgcn_s_sleep(8);
}

template <int VG>
__device__ __forceinline__ void force_vgprs()
{
	if constexpr (VG == 224) asm volatile("v_mov_b32 v223, 0" ::: "v223");
	if constexpr (VG == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
	if constexpr (VG == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
}

template <int THREADS, int VG>
__global__ __launch_bounds__(THREADS) void busy(float* sink, double us, unsigned long long* stamps)
{
	extern __shared__ float lds[];
	force_vgprs<VG>();
	if (threadIdx.x == 0 && stamps) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
	lds[threadIdx.x] = (float)threadIdx.x;
	__syncthreads();
	spin_us(us);
	if (lds[(threadIdx.x + 1) % THREADS] < 0) sink[0] = 1;
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
	float* sink; CHK(hipMalloc(&sink, 64));
	unsigned long long* stamps; CHK(hipMalloc(&stamps, 8 * 8192));
	int lo, hi; CHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
	hipStream_t mainst, side; CHK(hipStreamCreateWithPriority(&mainst, hipStreamNonBlocking, lo)); CHK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
	CHK(hipFuncSetAttribute((const void*)busy<512, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, 83 * 1024));
	CHK(hipFuncSetAttribute((const void*)busy<256, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
	CHK(hipFuncSetAttribute((const void*)busy<256, 224>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
	hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	auto run = [&](const char* name, int which, bool with_update) -> int {
		double tot = 0, worst = 0;
		const int reps = 8;
		for (int r = 0; r < reps; ++r) {
			if (with_update) hipLaunchKernelGGL((busy<256, 224>), dim3(7168), dim3(256), 32 * 1024, mainst, sink, 70.0, (unsigned long long*)nullptr);
			std::this_thread::sleep_for(std::chrono::microseconds(150 + 37 * r));
			CHK(hipEventRecord(e0, side));
			if (which == 0) hipLaunchKernelGGL((busy<512, 128>), dim3(1), dim3(512), 83 * 1024, side, sink, 57.0, (unsigned long long*)nullptr);
			if (which == 1) hipLaunchKernelGGL((busy<256, 64>), dim3(1), dim3(256), 80 * 1024, side, sink, 75.0, (unsigned long long*)nullptr);
			if (which == 2) hipLaunchKernelGGL((busy<256, 224>), dim3(100), dim3(256), 32 * 1024, side, sink, 45.0, (unsigned long long*)nullptr);
			if (which == 3) hipLaunchKernelGGL((busy<256, 64>), dim3(750), dim3(256), 48 * 1024, side, sink, 5.0, (unsigned long long*)nullptr);
			CHK(hipEventRecord(e1, side));
			CHK(hipEventSynchronize(e1));
			float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
			tot += ms; if (ms > worst) worst = ms;
			CHK(hipDeviceSynchronize());
		}
		printf("%-44s %s: mean %7.1f us  worst %7.1f us\n", name, with_update ? "beside the update" : "alone            ", tot / reps * 1e3, worst * 1e3);
		return 0;
	};
	for (int w = 0; w < 2; ++w) {
		if (run("fat    (512 thr, 128 VGPR, 83 KiB; 57 us)", 0, w)) return 1;
		if (run("sliver (256 thr,  64 VGPR, 80 KiB; 75 us)", 1, w)) return 1;
		if (run("gemmy  (100 wg, 224 VGPR, 32 KiB; 45 us)", 2, w)) return 1;
		if (run("sliver gemm (750 wg, 64 VGPR, 48 KiB; 5 us)", 3, w)) return 1;
	}
	// how long does the update itself take with / without sliver traffic beside it?
	for (int w = 0; w < 2; ++w) {
		CHK(hipDeviceSynchronize());
		auto t0 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL((busy<256, 224>), dim3(7168), dim3(256), 32 * 1024, mainst, sink, 70.0, (unsigned long long*)nullptr);
		if (w) for (int i = 0; i < 8; ++i) {
			hipLaunchKernelGGL((busy<256, 64>), dim3(1), dim3(256), 80 * 1024, side, sink, 75.0, (unsigned long long*)nullptr);
			hipLaunchKernelGGL((busy<256, 64>), dim3(750), dim3(256), 48 * 1024, side, sink, 5.0, (unsigned long long*)nullptr);
		}
		CHK(hipDeviceSynchronize());
		printf("update of 7168 x 70 us workgroups %s: %.1f us\n", w ? "with 8 x (sliver + sliver gemm) beside" : "alone", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
	}
	return 0;
}
