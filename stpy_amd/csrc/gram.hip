// gram.hip -- kernel (Gram) matrices  out[j][i] = kappa * phi(b_j, a_i),  the (|b|,|a|) orientation
// of stpy's KernelFunction.kernel.
//
// HBM-write bound by construction (q*n*sizeof(T) bytes out, (q+n)*d in), so the layout rules are
// the ones that matter: a 64 x 128 output tile per workgroup, lanes along the contiguous `i`
// dimension (each lane owns two adjacent columns -> 16-byte stores, 1 KiB contiguous per wave and
// row), the two point blocks staged once per tile in LDS already multiplied by the inverse
// lengthscales (the `a` block k-major so lanes read consecutive LDS words, the `b` block read as
// wave-wide broadcasts), row/column squared norms accumulated while staging.
//
// Distance forms (they follow the reference per kernel family):
//   SE / ARD      : ||a||^2 + ||b||^2 - 2<a,b>  (norm expansion, kernels.py:390-398, no clamp)
//   Matern 1/2..5/2: sum_k (a_k - b_k)^2        (direct differences as scipy/torch cdist,
//                    kernels.py:843, :944 -- exact zero on coincident points, which nu = 1/2 needs)
#include "common.h"

namespace stpy {

constexpr int GT_I = 128, GT_J = 64, GT_K = 16, G_THREADS = 256;

template <typename T>
struct GramArgs {
	const T* a; const T* b; const int32_t* cols; const T* inv_ls; T* out;
	int64_t lda, ldb, ldo;
	int n, q, d;
	T kappa, offset, diag_add;
	int kind, lower_only, combine, degree;
};

template <typename T> __device__ __forceinline__ T int_power(T base, int degree)
{
	T r = T(1);
	for (int e = 0; e < degree; ++e) r *= base;
	return r;
}

template <typename T> __device__ __forceinline__ T phi(int kind, T acc, T na, T nb, T offset, int degree)
{
	switch (kind) {
	case STPY_K_POLY: return int_power(acc + offset, degree);           // (<b,a> + c)^p, kernels.py:760
	case STPY_K_SE: {
		const T sq = na + nb - T(2) * acc;
		return exp(T(-0.5) * sq);
	}
	case STPY_K_MATERN12: return exp(-sqrt(acc));
	case STPY_K_MATERN32: { const T r = sqrt(acc) * T(1.7320508075688772935); return (T(1) + r) * exp(-r); }
	case STPY_K_MATERN52: { const T r = sqrt(acc) * T(2.2360679774997896964); return (T(1) + r + r * r / T(3)) * exp(-r); }
	default: return acc;   // LINEAR: plain dot product
	}
}

template <typename T, bool DIRECT>
__global__ __launch_bounds__(G_THREADS)
void gram_kernel(GramArgs<T> p)
{
	const int i0 = blockIdx.x * GT_I, j0 = blockIdx.y * GT_J;
	if (p.lower_only && i0 > j0 + GT_J - 1) return;       // tile strictly above the diagonal
	__shared__ T as[GT_K][GT_I];       // k-major: lanes read consecutive words
	__shared__ T bs[GT_J][GT_K + 1];
	__shared__ T na_s[GT_I], nb_s[GT_J];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int ic = lane * 2;            // this lane's two columns inside the tile
	const int jr = wave * 16;           // this wave's 16 rows inside the tile

	T acc[16][2];
#pragma unroll
	for (int r = 0; r < 16; ++r) acc[r][0] = acc[r][1] = T(0);
	T nsum = T(0);                      // squared norm of the point this thread stages (tid < 192)

	for (int k0 = 0; k0 < p.d; k0 += GT_K) {
		// ---- stage: a block 128 x 16 (thread t -> point t & 127, k half t >> 7), b block 64 x 16
		{
			const int pt = tid & 127, kh = tid >> 7;
			const int gi = min(i0 + pt, p.n - 1);
#pragma unroll
			for (int kk = 0; kk < GT_K / 2; ++kk) {
				const int k = kh * (GT_K / 2) + kk;
				const int kg = k0 + k;
				T v = T(0);
				if (kg < p.d) v = p.a[(int64_t)gi * p.lda + (p.cols ? p.cols[kg] : kg)] * p.inv_ls[kg];
				as[k][pt] = v;
			}
		}
		{
			const int pt = tid & 63, kq = tid >> 6;
			const int gj = min(j0 + pt, p.q - 1);
#pragma unroll
			for (int kk = 0; kk < GT_K / 4; ++kk) {
				const int k = kq * (GT_K / 4) + kk;
				const int kg = k0 + k;
				T v = T(0);
				if (kg < p.d) v = p.b[(int64_t)gj * p.ldb + (p.cols ? p.cols[kg] : kg)] * p.inv_ls[kg];
				bs[pt][k] = v;
			}
		}
		__syncthreads();
		if (!DIRECT) {                  // norms: threads 0..127 own an a point, 128..191 a b point
			if (tid < GT_I) {
#pragma unroll
				for (int k = 0; k < GT_K; ++k) nsum += as[k][tid] * as[k][tid];
			} else if (tid < GT_I + GT_J) {
#pragma unroll
				for (int k = 0; k < GT_K; ++k) nsum += bs[tid - GT_I][k] * bs[tid - GT_I][k];
			}
		}
#pragma unroll
		for (int k = 0; k < GT_K; ++k) {
			const T a0 = as[k][ic], a1 = as[k][ic + 1];
#pragma unroll
			for (int r = 0; r < 16; ++r) {
				const T bv = bs[jr + r][k];
				if (DIRECT) {
					const T d0 = a0 - bv, d1 = a1 - bv;
					acc[r][0] += d0 * d0;
					acc[r][1] += d1 * d1;
				} else {
					acc[r][0] += a0 * bv;
					acc[r][1] += a1 * bv;
				}
			}
		}
		__syncthreads();
	}
	if (!DIRECT) {
		if (tid < GT_I) na_s[tid] = nsum;
		else if (tid < GT_I + GT_J) nb_s[tid - GT_I] = nsum;
		__syncthreads();
	}

	const T na0 = DIRECT ? T(0) : na_s[ic], na1 = DIRECT ? T(0) : na_s[ic + 1];
	const int gi = i0 + ic;
	const bool vec_ok = (gi + 1 < p.n) && ((p.ldo & 1) == 0) && ((((uintptr_t)p.out) & (2 * sizeof(T) - 1)) == 0);
#pragma unroll
	for (int r = 0; r < 16; ++r) {
		const int gj = j0 + jr + r;
		if (gj >= p.q) break;
		const T nb = DIRECT ? T(0) : nb_s[jr + r];
		T v0 = p.kappa * phi<T>(p.kind, acc[r][0], na0, nb, p.offset, p.degree);
		T v1 = p.kappa * phi<T>(p.kind, acc[r][1], na1, nb, p.offset, p.degree);
		if (p.kind == STPY_K_LINEAR) { v0 += p.offset; v1 += p.offset; }
		T* o = p.out + (int64_t)gj * p.ldo + gi;
		if (p.combine == STPY_OUT_ADD) { if (gi < p.n) v0 += o[0]; if (gi + 1 < p.n) v1 += o[1]; }
		else if (p.combine == STPY_OUT_MUL) { if (gi < p.n) v0 *= o[0]; if (gi + 1 < p.n) v1 *= o[1]; }
		if (gi == gj) v0 += p.diag_add;
		if (gi + 1 == gj) v1 += p.diag_add;
		if (vec_ok) {
			typedef T v2 __attribute__((ext_vector_type(2)));
			*(v2*)o = v2{v0, v1};
		} else {
			if (gi < p.n) o[0] = v0;
			if (gi + 1 < p.n) o[1] = v1;
		}
	}
}

// ------------------------------------------------------------------------------------------
// MFMA route (SE / ARD, Matern 3/2 and 5/2, linear).  <b_j, a_i> over the d scaled coordinates is an
// NT contraction, so it runs on the GEMM of gemm.hip with the kernel function fused into its store
// epilogue: the fp64 vector ALU -- the bottleneck of the tile kernel above, which needs ~60
// instructions per element -- is left with the ~25 of the exp.  The points are first gathered
// (column subset), scaled by the inverse lengthscales and zero-padded to a whole K tile (16 doubles / 32 floats)
// of coordinates into the caller's workspace, together with their squared norms.
// Matern 1/2 keeps the direct-difference tile kernel: exp(-r) has a first-order term in r, and
// r from the norm expansion is only good to ~1e-8 on coincident points.
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256)
void prep_points_kernel(const T* __restrict__ x, int64_t n, int64_t ldx, int d, int dpad, const int32_t* cols, const T* __restrict__ inv_ls,
                        T* __restrict__ xs, T* __restrict__ nx)
{
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= n) return;
	T s = T(0);
	for (int k = 0; k < dpad; ++k) {
		T v = T(0);
		if (k < d) v = x[i * ldx + (cols ? cols[k] : k)] * inv_ls[k];
		xs[i * dpad + k] = v;
		s += v * v;
	}
	nx[i] = s;
}

static inline int64_t align16(int64_t b) { return (b + 15) & ~(int64_t)15; }

int64_t gram_workspace_bytes(int64_t n, int64_t q, int d, size_t esz)
{
	const int64_t kp = 128 / (int64_t)esz;           // one K tile of the GEMM: 16 doubles / 32 floats
	const int64_t dpad = (d + kp - 1) / kp * kp;
	return align16(n * dpad * esz) + align16(q * dpad * esz) + align16(n * esz) + align16(q * esz);
}

template <typename T>
int gram(int kind, const T* a, int64_t n, int64_t lda, const T* b, int64_t q, int64_t ldb, int d,
         const int32_t* cols, const T* inv_ls, double kappa, double offset, double diag_add,
         int lower_only, int combine, T* out, int64_t ldo, void* work, hipStream_t st)
{
	if (n <= 0 || q <= 0) return 0;
	const int degree = kind >> 8;           // STPY_K_POLY carries its degree above the family byte
	kind &= 0xff;
	if (kind == STPY_K_POLY && (degree < 1 || degree > 64)) { set_error("gram: polynomial degree %d out of range [1, 64]", degree); return -1; }
	if (kind != STPY_K_POLY && degree != 0) { set_error("gram: unknown kernel kind %d", kind | (degree << 8)); return -1; }
	if (work && kind != STPY_K_MATERN12 && kind >= STPY_K_SE && kind <= STPY_K_LINEAR) {
		constexpr int KP = 128 / (int)sizeof(T);
		const int dpad = (d + KP - 1) / KP * KP;
		char* w = (char*)work;
		T* as = (T*)w; w += align16(n * (int64_t)dpad * sizeof(T));
		T* bs = (T*)w; w += align16(q * (int64_t)dpad * sizeof(T));
		T* na = (T*)w; w += align16(n * sizeof(T));
		T* nb = (T*)w;
		const bool same = (a == b) && (n == q) && (lda == ldb);
		hipLaunchKernelGGL((prep_points_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, n, lda, d, dpad, cols, inv_ls, as, na);
		if (same) { bs = as; nb = na; }
		else hipLaunchKernelGGL((prep_points_kernel<T>), dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, b, q, ldb, d, dpad, cols, inv_ls, bs, nb);
		int rc = check_launch("gram prep");
		if (rc) return rc;
		if constexpr (sizeof(T) == 8) {          // aligned overwriting fp64 fills: the dedicated kernel (three small workgroups per CU)
			rc = gram_fill_f64(kind, (const double*)as, (const double*)bs, (const double*)na, (const double*)nb, dpad, n, q, kappa, offset, diag_add, lower_only, combine,
			                   (double*)out, ldo, st);
			if (rc != 0) return rc < 0 ? rc : 0;
		}
		GramEpilogue<T> epi{kind, combine, (T)kappa, (T)(kind == STPY_K_LINEAR ? offset : 0.0), (T)diag_add, na, nb, nullptr, T(1)};
		return gemm_nt<T>(q, n, dpad, bs, dpad, as, dpad, out, ldo, (T*)nullptr, 0, 3, lower_only, st, nullptr, nullptr, &epi);
	}
	if (n > INT32_MAX || q > INT32_MAX) { set_error("gram: dimension exceeds int32"); return -4; }
	if (kind < STPY_K_SE || kind > STPY_K_POLY) { set_error("gram: unknown kernel kind %d", kind); return -1; }
	GramArgs<T> p;
	p.a = a; p.b = b; p.cols = cols; p.inv_ls = inv_ls; p.out = out;
	p.lda = lda; p.ldb = ldb; p.ldo = ldo;
	p.n = (int)n; p.q = (int)q; p.d = d;
	p.kappa = (T)kappa; p.offset = (T)offset; p.diag_add = (T)diag_add;
	p.kind = kind; p.lower_only = lower_only; p.combine = combine; p.degree = degree;
	dim3 grid((unsigned)((n + GT_I - 1) / GT_I), (unsigned)((q + GT_J - 1) / GT_J));
	if (grid.y > 65535u) { set_error("gram: q too large for one launch"); return -7; }
	const bool direct = (kind == STPY_K_MATERN12 || kind == STPY_K_MATERN32 || kind == STPY_K_MATERN52);
	if (direct) hipLaunchKernelGGL((gram_kernel<T, true>), grid, dim3(G_THREADS), 0, st, p);
	else hipLaunchKernelGGL((gram_kernel<T, false>), grid, dim3(G_THREADS), 0, st, p);
	return check_launch("gram");
}

// out (op)= src elementwise over an m x n window (+ diag_add on the diagonal afterwards): the kernel algebra's fold of a
// multi-term item (kernels.py:146-157) and the M_i factors of the evidence gradient.  HBM-bound: one 16-byte access per
// lane where alignment allows.
template <typename T>
__global__ __launch_bounds__(256)
void combine_kernel(T* out, int64_t ldo, const T* src, int64_t lds,          // (src may be out itself: each element reads its own position)
                    int64_t row0, int64_t n, int combine, T diag_add, int vec)
{
	constexpr int CH = 16 / (int)sizeof(T);
	typedef T vch __attribute__((ext_vector_type(CH)));
	const int64_t row = row0 + blockIdx.y;
	T* o = out + row * ldo;
	const T* s = src + row * lds;
	if (vec) {
		const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * CH;
		if (c0 >= n) return;
		vch a = *(const vch*)(o + c0);
		const vch b = *(const vch*)(s + c0);
		a = combine == STPY_OUT_ADD ? a + b : (combine == STPY_OUT_MUL ? a * b : b);
#pragma unroll
		for (int e = 0; e < CH; ++e) if (c0 + e == row) a[e] += diag_add;
		*(vch*)(o + c0) = a;
	} else {
		const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
		if (c >= n) return;
		T a = o[c];
		const T b = s[c];
		a = combine == STPY_OUT_ADD ? a + b : (combine == STPY_OUT_MUL ? a * b : b);
		if (c == row) a += diag_add;
		o[c] = a;
	}
}

template <typename T>
int combine_into(int64_t m, int64_t n, T* out, int64_t ldo, const T* src, int64_t lds, int combine, double diag_add, hipStream_t st)
{
	if (m <= 0 || n <= 0) return 0;
	constexpr int CH = 16 / (int)sizeof(T);
	const int vec = (n % CH == 0) && (ldo % CH == 0) && (lds % CH == 0) && ((((uintptr_t)out | (uintptr_t)src) & 15) == 0);
	const int64_t per_row = vec ? n / CH : n;
	// rows ride on gridDim.y (<= 65535 per launch)
	for (int64_t r0 = 0; r0 < m; r0 += 65535) {
		const int64_t rows = (m - r0 < 65535) ? (m - r0) : 65535;
		hipLaunchKernelGGL((combine_kernel<T>), dim3((unsigned)((per_row + 255) / 256), (unsigned)rows), dim3(256), 0, st,
		                   out, ldo, src, lds, r0, n, combine, (T)diag_add, vec);
	}
	return check_launch("combine");
}

// H = (weight * Kinv - alpha alpha^T) o F, F the derivative factor of the kernel family (see gemm.hip, mode 4); Kinv is the
// symmetric inverse (n x n).  Hsrc == nullptr: in place over H; otherwise Kinv is read from Hsrc and H only written, so
// several terms can be formed from one inverse without copying it.  Same workspace layout as gram().
template <typename T>
int lml_weight(int kind, const T* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const T* inv_ls, double kappa, double weight,
               const T* alpha, const T* Hsrc, int64_t ldhs, T* H, int64_t ldh, void* work, hipStream_t st)
{
	if (n <= 0) return 0;
	if (kind < STPY_K_SE || kind > STPY_K_MATERN52) { set_error("lml_weight: kernel kind %d has no lengthscale gradient", kind); return -1; }
	constexpr int KP = 128 / (int)sizeof(T);
	const int dpad = (d + KP - 1) / KP * KP;
	char* w = (char*)work;
	T* as = (T*)w; w += align16(n * (int64_t)dpad * sizeof(T));
	w += align16(n * (int64_t)dpad * sizeof(T));
	T* na = (T*)w;
	hipLaunchKernelGGL((prep_points_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, n, ldx, d, dpad, cols, inv_ls, as, na);
	int rc = check_launch("lml_weight prep");
	if (rc) return rc;
	GramEpilogue<T> epi{kind, 0, (T)kappa, T(0), T(0), na, na, alpha, (T)weight};
	// (mode 4 reads the old tile through C2 when it is set; it never takes the second-copy store)
	return gemm_nt<T>(n, n, dpad, as, dpad, as, dpad, H, ldh, const_cast<T*>(Hsrc), Hsrc ? ldhs : 0, 4, 0, st, nullptr, nullptr, &epi);
}

// k(x_i, x_i): stationary kernels give kappa * phi(0); LINEAR gives kappa ||x_i[cols] * inv_ls||^2 + offset,
// POLY kappa (||x_i[cols] * inv_ls||^2 + offset)^degree
template <typename T>
__global__ void gram_diag_kernel(int kind, const T* __restrict__ x, int64_t m, int64_t ldx, int d, const int32_t* cols,
                                 const T* inv_ls, T kappa, T offset, int combine, T* __restrict__ out)
{
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= m) return;
	T v;
	const int degree = kind >> 8;
	kind &= 0xff;
	if (kind == STPY_K_LINEAR || kind == STPY_K_POLY) {
		T s = T(0);
		for (int k = 0; k < d; ++k) { const T t = x[i * ldx + (cols ? cols[k] : k)] * inv_ls[k]; s += t * t; }
		v = kind == STPY_K_LINEAR ? kappa * s + offset : kappa * int_power(s + offset, degree);
	} else {
		// -2<x,x> + ||x||^2 + ||x||^2 is exactly 0 in the reference's own evaluation order
		v = kappa;
	}
	if (combine == STPY_OUT_ADD) v += out[i];
	else if (combine == STPY_OUT_MUL) v *= out[i];
	out[i] = v;
}

template <typename T>
int gram_diag(int kind, const T* x, int64_t m, int64_t ldx, int d, const int32_t* cols, const T* inv_ls,
              double kappa, double offset, int combine, T* out, hipStream_t st)
{
	if (m <= 0) return 0;
	hipLaunchKernelGGL((gram_diag_kernel<T>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, st,
	                   kind, x, m, ldx, d, cols, inv_ls, (T)kappa, (T)offset, combine, out);
	return check_launch("gram_diag");
}

#define INST(T) \
	template int lml_weight<T>(int, const T*, int64_t, int64_t, int, const int32_t*, const T*, double, double, const T*, const T*, int64_t, T*, int64_t, void*, hipStream_t); \
	template int combine_into<T>(int64_t, int64_t, T*, int64_t, const T*, int64_t, int, double, hipStream_t); \
	template int gram<T>(int, const T*, int64_t, int64_t, const T*, int64_t, int64_t, int, const int32_t*, const T*, double, double, double, int, int, T*, int64_t, void*, hipStream_t); \
	template int gram_diag<T>(int, const T*, int64_t, int64_t, int, const int32_t*, const T*, double, double, int, T*, hipStream_t);
INST(double)
INST(float)

}  // namespace stpy
