// Does this platform support stream memory operations (hipStreamWaitValue32 / hipStreamWriteValue32), on which kind of memory,
// and what does a kernel -> stream hand-off through them cost?   hipcc --offload-arch=gfx950 -O2 tools/waitvalue_probe.hip -o /tmp/wv && /tmp/wv
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void set_flag(volatile unsigned* f, unsigned v, unsigned long long* stamp)
{
	__threadfence_system();
	*f = v;
	__threadfence_system();
	if (stamp) *stamp = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(unsigned long long* stamp) { *stamp = __builtin_amdgcn_s_memrealtime(); }
__global__ void spin_kernel(volatile unsigned* f, unsigned want, unsigned long long* stamp)
{
	unsigned n = 0;
	while (__hip_atomic_load((unsigned*)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && n < 2000000u) { __builtin_amdgcn_s_sleep(8); ++n; }
	*stamp = __builtin_amdgcn_s_memrealtime();
}

int main()
{
	int attr = -1;
	hipError_t e = hipDeviceGetAttribute(&attr, hipDeviceAttributeCanUseStreamWaitValue, 0);
	printf("hipDeviceAttributeCanUseStreamWaitValue: %s, value %d\n", hipGetErrorString(e), attr);
	hipStream_t a, b;
	CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
	CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
	unsigned long long* stamps;
	CK(hipHostMalloc(&stamps, 64 * sizeof(unsigned long long), hipHostMallocDefault));
	for (int kind = 0; kind < 3; ++kind) {
		unsigned* flag = nullptr;
		const char* name = kind == 0 ? "hipExtMallocWithFlags(signal memory)" : (kind == 1 ? "hipMalloc" : "hipHostMalloc (pinned host)");
		if (kind == 0) e = hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory);
		else if (kind == 1) e = hipMalloc((void**)&flag, 8);
		else e = hipHostMalloc((void**)&flag, 8, hipHostMallocDefault);
		if (e != hipSuccess) { printf("%s: allocation failed: %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
		CK(hipMemset(flag, 0, 8));
		CK(hipDeviceSynchronize());
		// stream a: wait until *flag >= 1, then stamp; stream b: a kernel sets the flag (and stamps)
		e = hipStreamWaitValue32(a, flag, 1, hipStreamWaitValueGte, 0xffffffffu);
		if (e != hipSuccess) { printf("%s: hipStreamWaitValue32 -> %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
		hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, a, stamps + 1);
		auto t0 = std::chrono::steady_clock::now();
		hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, b, flag, 1u, stamps + 0);
		e = hipStreamSynchronize(a);
		auto t1 = std::chrono::steady_clock::now();
		printf("%s: kernel store -> hipStreamWaitValue32 -> next kernel: %s, device hand-off %.2f us (host wall %.1f us)\n", name, hipGetErrorString(e),
		       (double)(stamps[1] - stamps[0]) / 100.0, std::chrono::duration<double, std::micro>(t1 - t0).count());
		// the other direction: stream b writes the value with hipStreamWriteValue32, a resident kernel on stream a polls it
		CK(hipMemset(flag, 0, 8));
		CK(hipDeviceSynchronize());
		hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, a, flag, 7u, stamps + 3);
		hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(1), 0, b, stamps + 2);
		e = hipStreamWriteValue32(b, flag, 7, 0);
		if (e != hipSuccess) { printf("%s: hipStreamWriteValue32 -> %s\n", name, hipGetErrorString(e)); (void)hipGetLastError(); hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, b, flag, 7u, (unsigned long long*)nullptr); }
		CK(hipDeviceSynchronize());
		printf("%s: hipStreamWriteValue32 -> polling kernel: %.2f us after the kernel in front of it\n", name, (double)(stamps[3] - stamps[2]) / 100.0);
		// chain of 20 hand-offs kernel(b) -> wait(a) -> kernel(a) -> wait(b) ... : average per hop
		CK(hipMemset(flag, 0, 8));
		CK(hipDeviceSynchronize());
		t0 = std::chrono::steady_clock::now();
		bool ok = true;
		for (int i = 0; i < 20 && ok; ++i) {
			hipStream_t w = (i & 1) ? b : a, s = (i & 1) ? a : b;
			hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, (unsigned)(i + 1), (unsigned long long*)nullptr);
			ok = hipStreamWaitValue32(w, flag, (unsigned)(i + 1), hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
		}
		CK(hipDeviceSynchronize());
		t1 = std::chrono::steady_clock::now();
		printf("%s: 20 alternating hand-offs: %.1f us each (host wall, includes launches)\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / 20.0);
	}
	// for comparison: the same ping-pong with events
	hipEvent_t ev[20];
	for (auto& x : ev) CK(hipEventCreateWithFlags(&x, hipEventDisableTiming));
	unsigned* flag;
	CK(hipMalloc((void**)&flag, 8));
	CK(hipDeviceSynchronize());
	auto t0 = std::chrono::steady_clock::now();
	for (int i = 0; i < 20; ++i) {
		hipStream_t w = (i & 1) ? b : a, s = (i & 1) ? a : b;
		hipLaunchKernelGGL(set_flag, dim3(1), dim3(1), 0, s, flag, (unsigned)(i + 1), (unsigned long long*)nullptr);
		CK(hipEventRecord(ev[i], s));
		CK(hipStreamWaitEvent(w, ev[i], 0));
	}
	CK(hipDeviceSynchronize());
	auto t1 = std::chrono::steady_clock::now();
	printf("events: 20 alternating hand-offs: %.1f us each (host wall, includes launches)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 20.0);
	return 0;
}
