// rff.hip -- random Fourier feature embed  Z[i][j] = c * {cos | sin}(<W_j, x_i> (+ b_j)).
//
// The phase matrix X W^T is an NT contraction with K = d, so it runs on the MFMA GEMM of gemm.hip
// (fp32: v_mfma_f32_16x16x4_f32, fp64: v_mfma_f64_16x16x4_f64) with the trig + scale fused into the
// store epilogue: the n x m phase matrix never exists in memory (the reference materialises it four
// times, embedding.py:234-241) and the vector ALU is left for the sin/cos.  At C5 (d = 64, fp32) the
// contraction (1.1e12 flop) and the 34 GB of output are about equally expensive, so neither a
// VALU dot product nor an unfused GEMM + elementwise pass can reach the HBM roofline.
//
// Column layout quirk kept from the reference (embedding.py:236-239): without a bias the cos
// half uses frequency rows 0..m/2-1 and the sin half uses the *other* rows m/2..m-1.
#include "common.h"

namespace stpy {

template <typename T>
int rff_embed(const T* x, int64_t n, int64_t ldx, int d, const T* W, int64_t ldw, int64_t m,
              const T* bias, double scale, T* out, int64_t ldo, hipStream_t st)
{
	if (n <= 0 || m <= 0) return 0;
	if (m % 2 != 0) { set_error("rff_embed: m must be even (embedding.py:84-85)"); return -8; }
	RffEpilogue<T> epi{(int)(m / 2), (T)scale, bias};
	return gemm_nt<T>(n, m, d, x, ldx, W, ldw, out, ldo, (T*)nullptr, 0, 2, 0, st, nullptr, &epi);
}

template int rff_embed<double>(const double*, int64_t, int64_t, int, const double*, int64_t, int64_t, const double*, double, double*, int64_t, hipStream_t);
template int rff_embed<float>(const float*, int64_t, int64_t, int, const float*, int64_t, int64_t, const float*, double, float*, int64_t, hipStream_t);

}  // namespace stpy
