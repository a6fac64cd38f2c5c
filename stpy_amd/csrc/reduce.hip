// Small reductions and layout helpers around the evidence gradient, the samplers and the scalar summaries of the estimator
// (stpy_tril, stpy_trace_dot, stpy_scaled_points_t, stpy_lml_grad_reduce).  None of them is on a roofline: each touches
// O(n^2) bytes once (tril) or O(n d) bytes; they exist so that no arithmetic on device data is left to torch on the product path.
// Every reduction has a fixed summation order (one workgroup, strided partial sums, shuffle tree, 16 wave partials added in
// index order): results are bit-reproducible from run to run.
#include "common.h"

namespace stpy {

// ------------------------------------------------------------------------------------------
// zero the strict upper triangle (the in-place Cholesky leaves scratch there): 64 x 64 tiles, tiles below the diagonal untouched
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void tril_kernel(T* __restrict__ A, int64_t lda, int n)
{
	const int ti = blockIdx.y, tj = blockIdx.x;
	if (tj < ti) return;
	const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
	const int gc = tj * 64 + tx;
	if (gc >= n) return;
	for (int r = ty; r < 64; r += 4) {
		const int gr = ti * 64 + r;
		if (gr < n && gc > gr) A[(int64_t)gr * lda + gc] = T(0);
	}
}

template <typename T>
int tril(int64_t n, T* A, int64_t lda, hipStream_t st)
{
	if (n <= 0) return 0;
	if (n > INT32_MAX) { set_error("tril: n exceeds int32"); return -2; }
	const unsigned t = (unsigned)((n + 63) / 64);
	hipLaunchKernelGGL((tril_kernel<T>), dim3(t, t), dim3(256), 0, st, A, lda, (int)n);
	return check_launch("tril");
}

// block-wide sum in a fixed order; valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum_1024(T v, T* red16)
{
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
	__syncthreads();                                   // (red16 may still be read from the previous call)
	if ((threadIdx.x & 63) == 0) red16[threadIdx.x >> 6] = v;
	__syncthreads();
	T s = T(0);
	if (threadIdx.x == 0)
		for (int w = 0; w < 16; ++w) s += red16[w];
	return s;
}

// out2[0] = sum_i A_ii (A may be null: 0), out2[1] = <u, v> (u may be null: 0)
template <typename T>
__global__ __launch_bounds__(1024)
void trace_dot_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ u, const T* __restrict__ v, int n, T* __restrict__ out2)
{
	__shared__ T red[16];
	T s1 = T(0), s2 = T(0);
	for (int i = threadIdx.x; i < n; i += 1024) {
		if (A) s1 += A[(int64_t)i * lda + i];
		if (u) s2 += u[i] * v[i];
	}
	s1 = block_sum_1024(s1, red);
	s2 = block_sum_1024(s2, red);
	if (threadIdx.x == 0) { out2[0] = s1; out2[1] = s2; }
}

template <typename T>
int trace_dot(int64_t n, const T* A, int64_t lda, const T* u, const T* v, T* out2, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("trace_dot: n exceeds int32"); return -2; }
	hipLaunchKernelGGL((trace_dot_kernel<T>), dim3(1), dim3(1024), 0, st, A, lda, u, v, (int)n, out2);
	return check_launch("trace_dot");
}

// out[k*ldo + i] = x[i*ldx + cols[k]] * inv_ls[k]  (k < d),  and out[d*ldo + i] = 1 when ones_row: the NT operand [Xs | 1]^T
template <typename T>
__global__ __launch_bounds__(256)
void scaled_points_t_kernel(const T* __restrict__ x, int64_t ldx, int n, int d, const int32_t* __restrict__ cols, const T* __restrict__ inv_ls,
                            T* __restrict__ out, int64_t ldo)
{
	const int i = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y;
	if (i >= n) return;
	if (k == d) { out[(int64_t)k * ldo + i] = T(1); return; }
	const int c = cols ? cols[k] : k;
	out[(int64_t)k * ldo + i] = x[(int64_t)i * ldx + c] * inv_ls[k];
}

template <typename T>
int scaled_points_t(const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, T* out, int64_t ldo, int ones_row, hipStream_t st)
{
	if (n <= 0) return 0;
	if (n > INT32_MAX) { set_error("scaled_points_t: n exceeds int32"); return -2; }
	hipLaunchKernelGGL((scaled_points_t_kernel<T>), dim3((unsigned)((n + 255) / 256), (unsigned)(d + (ones_row ? 1 : 0))), dim3(256), 0, st,
	                   x, ldx, (int)n, d, cols, inv_ls, out, ldo);
	return check_launch("scaled_points_t");
}

// With P = H [Xs | 1]  (n x (d+1); column d = h = H 1):
//   S_k = sum_i xs_ik^2 h_i - xs_ik P_ik  ( = 1/2 sum_ij H_ij (xs_ik - xs_jk)^2 for symmetric H ),
//   acc[pidx[k]] += S_k * inv_ls[k]                 -- d/d(lengthscale) of the evidence per coordinate of a kernel term
// One workgroup, coordinates in order: several coordinates that share a parameter (an isotropic 'gamma') are added in a fixed order.
template <typename T>
__global__ __launch_bounds__(1024)
void lml_grad_reduce_kernel(const T* __restrict__ x, int64_t ldx, int n, int d, const int32_t* __restrict__ cols, const T* __restrict__ inv_ls,
                            const T* __restrict__ P, int64_t ldp, const int32_t* __restrict__ pidx, T* __restrict__ acc)
{
	__shared__ T red[16];
	for (int k = 0; k < d; ++k) {
		const int c = cols ? cols[k] : k;
		const T il = inv_ls[k];
		T s = T(0);
		for (int i = threadIdx.x; i < n; i += 1024) {
			const T xs = x[(int64_t)i * ldx + c] * il;
			const T* Pi = P + (int64_t)i * ldp;
			s += xs * xs * Pi[d] - xs * Pi[k];
		}
		s = block_sum_1024(s, red);
		if (threadIdx.x == 0) acc[pidx ? pidx[k] : k] += s * il;
	}
}

template <typename T>
int lml_grad_reduce(const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, const T* P, int64_t ldp,
                    const int32_t* pidx, T* acc, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("lml_grad_reduce: n exceeds int32"); return -2; }
	hipLaunchKernelGGL((lml_grad_reduce_kernel<T>), dim3(1), dim3(1024), 0, st, x, ldx, (int)n, d, cols, inv_ls, P, ldp, pidx, acc);
	return check_launch("lml_grad_reduce");
}

// Full-covariance kernel items (kernels.py:464-549: z = x[:, cols] cov, then a stationary kernel of |z_i - z_j|): with P = H [Z | 1]
// (n x (p+1), H = (w K^-1 - alpha alpha^T) o kappa F as stpy_lml_weight forms it for the mapped points Z with unit lengthscales),
//   out[a * p + m] -= sum_i x[i, cols[a]] * (P[i][p] * z[i][m] - P[i][m])       ( = -1/2 sum_ij H_ij (z_i - z_j)_m (x_i - x_j)_a )
// which is d/dcov[a][m] of the evidence.  One workgroup, the dg x p entries in order, fixed summation order.
template <typename T>
__global__ __launch_bounds__(1024)
void lml_grad_cov_reduce_kernel(const T* __restrict__ x, int64_t ldx, int n, int dg, const int32_t* __restrict__ cols,
                                const T* __restrict__ z, int64_t ldz, int pdim, const T* __restrict__ P, int64_t ldp, T* __restrict__ out)
{
	__shared__ T red[16];
	for (int a = 0; a < dg; ++a) {
		const int c = cols ? cols[a] : a;
		for (int m = 0; m < pdim; ++m) {
			T s = T(0);
			for (int i = threadIdx.x; i < n; i += 1024) {
				const T* Pi = P + (int64_t)i * ldp;
				s += x[(int64_t)i * ldx + c] * (Pi[pdim] * z[(int64_t)i * ldz + m] - Pi[m]);
			}
			s = block_sum_1024(s, red);
			if (threadIdx.x == 0) out[a * pdim + m] -= s;
		}
	}
}

template <typename T>
int lml_grad_cov_reduce(const T* x, int64_t n, int64_t ldx, int dg, const int32_t* cols, const T* z, int64_t ldz, int pdim, const T* P, int64_t ldp,
                        T* out, hipStream_t st)
{
	if (n > INT32_MAX) { set_error("lml_grad_cov_reduce: n exceeds int32"); return -2; }
	hipLaunchKernelGGL((lml_grad_cov_reduce_kernel<T>), dim3(1), dim3(1024), 0, st, x, ldx, (int)n, dg, cols, z, ldz, pdim, P, ldp, out);
	return check_launch("lml_grad_cov_reduce");
}

#define INST(T) \
	template int lml_grad_cov_reduce<T>(const T*, int64_t, int64_t, int, const int32_t*, const T*, int64_t, int, const T*, int64_t, T*, hipStream_t); \
	template int tril<T>(int64_t, T*, int64_t, hipStream_t); \
	template int trace_dot<T>(int64_t, const T*, int64_t, const T*, const T*, T*, hipStream_t); \
	template int scaled_points_t<T>(const T*, int64_t, int64_t, int, const int32_t*, const T*, T*, int64_t, int, hipStream_t); \
	template int lml_grad_reduce<T>(const T*, int64_t, int64_t, int, const int32_t*, const T*, const T*, int64_t, const int32_t*, T*, hipStream_t);
INST(double)
INST(float)

}  // namespace stpy
