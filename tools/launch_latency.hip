// Dependent-launch latency probe: N tiny kernels in one stream, plain launches vs one captured hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/launch_latency.hip -o /tmp/ll && /tmp/ll
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void tiny(double* x, int step) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += step; }
__global__ void wide(double* x, int step) { x[blockIdx.x * blockDim.x + threadIdx.x] += step; }   // 64 workgroups
int main()
{
	double* x; hipMalloc(&x, 1 << 20); hipMemset(x, 0, 1 << 20);
	hipStream_t st; hipStreamCreate(&st);
	const int N = 300;
	for (int variant = 0; variant < 2; ++variant) {
		auto launch = [&](int i) { if (variant == 0) tiny<<<1, 64, 0, st>>>(x, i); else wide<<<64, 256, 0, st>>>(x, i); };
		for (int i = 0; i < 20; ++i) launch(i);
		hipStreamSynchronize(st);
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < N; ++i) launch(i);
		hipStreamSynchronize(st);
		double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
		printf("%s: %d dependent plain launches: %.1f us each\n", variant ? "64-workgroup kernel" : "1-workgroup kernel ", N, us / N);
		hipGraph_t g; hipGraphExec_t ge;
		hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
		for (int i = 0; i < N; ++i) launch(i);
		hipStreamEndCapture(st, &g);
		hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
		hipGraphLaunch(ge, st); hipStreamSynchronize(st);
		t0 = std::chrono::steady_clock::now();
		hipGraphLaunch(ge, st);
		hipStreamSynchronize(st);
		us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
		printf("%s: same %d kernels as one graph launch: %.1f us each\n", variant ? "64-workgroup kernel" : "1-workgroup kernel ", N, us / N);
		hipGraphExecDestroy(ge); hipGraphDestroy(g);
	}
	return 0;
}
