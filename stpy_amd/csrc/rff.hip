// rff.hip -- random Fourier feature embed  Z[i][j] = c * {cos | sin}(<W_j, x_i> (+ b_j)).
//
// The phase matrix X W^T is an NT contraction with K = d, so it runs on the MFMA GEMM of gemm.hip.
//   fp32 (the performance configuration, BASELINE config 5): trig + scale are fused into the GEMM's
//        store epilogue (hardware v_sin_f32 / v_cos_f32 on the phase reduced to revolutions); the
//        n x m phase matrix never exists in memory (the reference materialises it four times,
//        embedding.py:234-241).  At d = 64 the contraction (1.1e12 flop) and the 34 GB of output
//        cost about the same, so neither a VALU dot product nor an unfused pass reaches the roofline.
//   fp64 (the reference's dtype, used for parity): GEMM into `out`, then one in-place elementwise
//        pass with libm-accurate sin/cos.  (Fusing the fp64 libm bodies into the GEMM epilogue makes
//        hipcc spill every accumulator to scratch.)
//
// Column layout quirk kept from the reference (embedding.py:236-239): without a bias the cos
// half uses frequency rows 0..m/2-1 and the sin half uses the *other* rows m/2..m-1.
#include "common.h"

namespace stpy {

__global__ __launch_bounds__(256)
void rff_trig_f64_kernel(double* __restrict__ out, int64_t ldo, int64_t n, int m, int half, const double* __restrict__ bias, double scale)
{
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	const int64_t total = n * (int64_t)(m / 2);
	if (idx >= total) return;
	const int64_t i = idx / (m / 2);
	const int j = (int)(idx - i * (m / 2)) * 2;           // two adjacent features per thread: 16-byte accesses
	double* o = out + i * ldo + j;
	double q0 = o[0], q1 = o[1];
	if (bias) { q0 += bias[j]; q1 += bias[j + 1]; }
	o[0] = scale * ((bias || j < half) ? cos(q0) : sin(q0));
	o[1] = scale * ((bias || j + 1 < half) ? cos(q1) : sin(q1));
}

// transposed embedding Phi^T (m x n): the feature index is the row
__global__ __launch_bounds__(256)
void rff_trig_f64_t_kernel(double* __restrict__ out, int64_t ldo, int64_t n, int m, int half, const double* __restrict__ bias, double scale)
{
	const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (idx >= (int64_t)m * n) return;
	const int j = (int)(idx / n);
	const int64_t i = idx - (int64_t)j * n;
	double q = out[(int64_t)j * ldo + i];
	if (bias) q += bias[j];
	out[(int64_t)j * ldo + i] = scale * ((bias || j < half) ? cos(q) : sin(q));
}

template <typename T>
int rff_embed(const T* x, int64_t n, int64_t ldx, int d, const T* W, int64_t ldw, int64_t m,
              const T* bias, double scale, T* out, int64_t ldo, int transposed, hipStream_t st);

template <>
int rff_embed<float>(const float* x, int64_t n, int64_t ldx, int d, const float* W, int64_t ldw, int64_t m,
                     const float* bias, double scale, float* out, int64_t ldo, int transposed, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (m % 2 != 0) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	RffEpilogue<float> epi{(int)(m / 2), (float)scale, bias, transposed ? 1 : 0};
	if (transposed) return gemm_nt<float>(m, n, d, W, ldw, x, ldx, out, ldo, (float*)nullptr, 0, 2, 0, st, nullptr, &epi);
	return gemm_nt<float>(n, m, d, x, ldx, W, ldw, out, ldo, (float*)nullptr, 0, 2, 0, st, nullptr, &epi);
}

template <>
int rff_embed<double>(const double* x, int64_t n, int64_t ldx, int d, const double* W, int64_t ldw, int64_t m,
                      const double* bias, double scale, double* out, int64_t ldo, int transposed, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (m % 2 != 0) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	if (m > INT32_MAX) { set_error("rff_embed: m exceeds int32"); return -7; }
	if (transposed) {
		int rc = gemm_nt<double>(m, n, d, W, ldw, x, ldx, out, ldo, (double*)nullptr, 0, 0, 0, st);
		if (rc) return rc;
		const int64_t total = m * n;
		hipLaunchKernelGGL(rff_trig_f64_t_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, ldo, n, (int)m, (int)(m / 2), bias, scale);
		return check_launch("rff_trig_f64_t");
	}
	int rc = gemm_nt<double>(n, m, d, x, ldx, W, ldw, out, ldo, (double*)nullptr, 0, 0, 0, st);
	if (rc) return rc;
	const int64_t total = n * (m / 2);
	hipLaunchKernelGGL(rff_trig_f64_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, out, ldo, n, (int)m, (int)(m / 2), bias, scale);
	return check_launch("rff_trig_f64");
}

}  // namespace stpy
