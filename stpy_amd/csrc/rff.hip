// rff.hip -- random Fourier feature embed  Z[i][j] = c * {cos | sin}(<W_j, x_i> (+ b_j)).
// HBM-write bound (n*m outputs against (n+m)*d inputs): 64 x 128 output tiles, lanes along the
// contiguous feature index j (two adjacent features per lane -> 16-byte stores for f64, 8 for
// f32), W block staged k-major in LDS, x block read as broadcasts; the trig epilogue is fused so
// the n x m phase matrix never exists in memory (the reference materialises it 4 times,
// embedding.py:234-241).
//
// Column layout quirk kept from the reference (embedding.py:236-239): without a bias the cos
// half uses frequency rows 0..m/2-1 and the sin half uses the *other* rows m/2..m-1.
#include "common.h"

namespace stpy {

constexpr int RT_J = 128, RT_I = 64, RT_K = 16, R_THREADS = 256;

template <typename T>
struct RffArgs {
	const T* x; const T* W; const T* bias; T* out;
	int64_t ldx, ldw, ldo;
	int n, m, d;
	T scale;
};

template <typename T>
__global__ __launch_bounds__(R_THREADS)
void rff_kernel(RffArgs<T> p)
{
	const int j0 = blockIdx.x * RT_J, i0 = blockIdx.y * RT_I;
	__shared__ T ws[RT_K][RT_J];
	__shared__ T xs[RT_I][RT_K + 1];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int jc = lane * 2, ir = wave * 16;

	T acc[16][2];
#pragma unroll
	for (int r = 0; r < 16; ++r) acc[r][0] = acc[r][1] = T(0);

	for (int k0 = 0; k0 < p.d; k0 += RT_K) {
		{
			const int pt = tid & 127, kh = tid >> 7;
			const int gj = min(j0 + pt, p.m - 1);
#pragma unroll
			for (int kk = 0; kk < RT_K / 2; ++kk) {
				const int k = kh * (RT_K / 2) + kk;
				ws[k][pt] = (k0 + k < p.d) ? p.W[(int64_t)gj * p.ldw + k0 + k] : T(0);
			}
		}
		{
			const int pt = tid & 63, kq = tid >> 6;
			const int gi = min(i0 + pt, p.n - 1);
#pragma unroll
			for (int kk = 0; kk < RT_K / 4; ++kk) {
				const int k = kq * (RT_K / 4) + kk;
				xs[pt][k] = (k0 + k < p.d) ? p.x[(int64_t)gi * p.ldx + k0 + k] : T(0);
			}
		}
		__syncthreads();
#pragma unroll
		for (int k = 0; k < RT_K; ++k) {
			const T w0 = ws[k][jc], w1 = ws[k][jc + 1];
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const T xv = xs[ir + r][k];
				acc[r][0] += w0 * xv;
				acc[r][1] += w1 * xv;
			}
		}
		__syncthreads();
	}

	const int gj = j0 + jc;
	const int half = p.m / 2;
	T b0 = T(0), b1 = T(0);
	if (p.bias) { if (gj < p.m) b0 = p.bias[gj]; if (gj + 1 < p.m) b1 = p.bias[gj + 1]; }
	const bool cos0 = p.bias || gj < half, cos1 = p.bias || gj + 1 < half;
	const bool vec_ok = (gj + 1 < p.m) && ((p.ldo & 1) == 0) && ((((uintptr_t)p.out) & (2 * sizeof(T) - 1)) == 0);
#pragma unroll
	for (int r = 0; r < 16; ++r) {
		const int gi = i0 + ir + r;
		if (gi >= p.n) break;
		const T q0 = acc[r][0] + b0, q1 = acc[r][1] + b1;
		const T v0 = p.scale * (cos0 ? cos(q0) : sin(q0));
		const T v1 = p.scale * (cos1 ? cos(q1) : sin(q1));
		T* o = p.out + (int64_t)gi * p.ldo + gj;
		if (vec_ok) {
			typedef T v2 __attribute__((ext_vector_type(2)));
			*(v2*)o = v2{v0, v1};
		} else {
			if (gj < p.m) o[0] = v0;
			if (gj + 1 < p.m) o[1] = v1;
		}
	}
}

template <typename T>
int rff_embed(const T* x, int64_t n, int64_t ldx, int d, const T* W, int64_t ldw, int64_t m,
              const T* bias, double scale, T* out, int64_t ldo, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (n > INT32_MAX || m > INT32_MAX) { set_error("rff_embed: dimension exceeds int32"); return -3; }
	if (m % 2 != 0) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	RffArgs<T> p;
	p.x = x; p.W = W; p.bias = bias; p.out = out;
	p.ldx = ldx; p.ldw = ldw; p.ldo = ldo;
	p.n = (int)n; p.m = (int)m; p.d = d;
	p.scale = (T)scale;
	dim3 grid((unsigned)((m + RT_J - 1) / RT_J), (unsigned)((n + RT_I - 1) / RT_I));
	hipLaunchKernelGGL((rff_kernel<T>), grid, dim3(R_THREADS), 0, st, p);
	return check_launch("rff_embed");
}

template int rff_embed<double>(const double*, int64_t, int64_t, int, const double*, int64_t, int64_t, const double*, double, double*, int64_t, hipStream_t);
template int rff_embed<float>(const float*, int64_t, int64_t, int, const float*, int64_t, int64_t, const float*, double, float*, int64_t, hipStream_t);

}  // namespace stpy
