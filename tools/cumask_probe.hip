// Which CUs does a hipExtStreamCreateWithCUMask bit select on MI355X?  Launches a flood of short workgroups on a masked
// stream and histograms (XCC_ID, SE_ID, CU_ID) as the waves themselves read them from the hardware registers.
//   hipcc --offload-arch=gfx950 -O3 tools/cumask_probe.hip -o /tmp/cumask && /tmp/cumask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <set>
#include <map>

__global__ void where(unsigned* out)
{
	if (threadIdx.x == 0) {
		const unsigned hwid = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);        // HW_REG_HW_ID bits [15:0]
		const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);         // HW_REG_XCC_ID bits [3:0]
		out[blockIdx.x] = (xcc << 16) | hwid;
	}
	const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) __builtin_amdgcn_s_sleep(8);       // 20 us: let the grid spread
}

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
	const int G = 8192;
	unsigned* d; CHK(hipMalloc(&d, G * 4));
	std::vector<unsigned> h(G);
	auto run = [&](const char* name, const std::vector<unsigned>& mask) -> int {
		hipStream_t st;
		if (mask.empty()) CHK(hipStreamCreate(&st));
		else CHK(hipExtStreamCreateWithCUMask(&st, (unsigned)mask.size(), mask.data()));
		hipLaunchKernelGGL(where, dim3(G), dim3(64), 0, st, d);
		CHK(hipStreamSynchronize(st));
		CHK(hipMemcpy(h.data(), d, G * 4, hipMemcpyDeviceToHost));
		std::map<unsigned, std::set<unsigned>> per_xcc;
		for (unsigned v : h) {
			const unsigned xcc = v >> 16, hw = v & 0xffff;
			const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;      // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
			per_xcc[xcc].insert(se * 100 + sh * 50 + cu);
		}
		printf("%-40s", name);
		int total = 0;
		for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
		printf("  total %d CUs\n", total);
		if (total <= 16) for (auto& kv : per_xcc) { printf("      xcc%u:", kv.first); for (unsigned c : kv.second) printf(" se%u.cu%u", c / 100, c % 50); printf("\n"); }
		CHK(hipStreamDestroy(st));
		return 0;
	};
	if (run("no mask", {})) return 1;
	std::vector<unsigned> all(8, 0xffffffffu);
	if (run("all 256 bits", all)) return 1;
	std::vector<unsigned> low8(8, 0); low8[0] = 0xff;
	if (run("bits 0-7 only", low8)) return 1;
	std::vector<unsigned> not_low8(8, 0xffffffffu); not_low8[0] = 0xffffff00u;
	if (run("all but bits 0-7", not_low8)) return 1;
	std::vector<unsigned> bit0(8, 0); bit0[0] = 1;
	if (run("bit 0 only", bit0)) return 1;
	std::vector<unsigned> hi8(8, 0); hi8[7] = 0xff000000u;
	if (run("bits 248-255 only", hi8)) return 1;
	std::vector<unsigned> b8_15(8, 0); b8_15[0] = 0xff00;
	if (run("bits 8-15 only", b8_15)) return 1;
	return 0;
}
