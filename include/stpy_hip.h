/*
 * stpy_hip.h -- C ABI of libstpy_hip.so: the MI355X (gfx950) implementation of the dense
 * linear-algebra hot path of Mojusko/stpy (kernel Gram matrices, blocked Cholesky, triangular
 * solves, GP prediction epilogue, log-marginal reductions, random-Fourier-feature embed).
 *
 * The reference has no FFI of its own: the path sits behind Python methods that call torch CPU
 * ops.  Each entry point below replaces the torch/scipy call sequence cited next to it
 * (file:line relative to the reference root); INTEGRATION.md shows the ctypes binding a
 * maintainer would add on the reference side.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer into caller-owned memory (PyTorch-ROCm storage in the
 *     shipped host code); matrices are row-major with an explicit leading dimension in ELEMENTS;
 *   - dtype: 0 = float64, 1 = float32 (all operands of one call share it);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - no allocation of data, no ownership transfer; workspaces are sized by the *_workspace_bytes queries and
 *     passed with their size (`work_bytes`), the inverse-diagonal-block array `winv` with its element count
 *     (`winv_elems`): a buffer smaller than the query's answer for the same arguments is refused (-20 / -21)
 *     instead of being overrun;
 *   - state kept by the library, all of it documented here:
 *       * a thread-local last-error string;
 *       * per (device, caller stream): one high-priority side stream + three events, created on the first
 *         stpy_potrf / stpy_trsm_right_lt call on that stream and kept for the life of the process (the panel
 *         look-ahead).  Host threads that drive DIFFERENT streams may call concurrently; calls that share a
 *         stream must be issued by one thread at a time (they are ordered by the stream, like any HIP work); with them 64 bytes
 *         of device memory (the ticket / counter words of the one-launch vector solve), the library's only allocation;
 *       * the launch profiler's record table (stpy_profile_*), guarded by a mutex, off by default;
 *       * the nine ROUTE switches of stpy_tune (which shipped kernel serves a call where the library normally decides by
 *         size): process-wide integers read at launch time, never written by the shipped host code -- tests/ use them to
 *         reach every shipped path at small sizes; they must not be changed while another thread is inside the library.
 *         Behaviour a caller may legitimately want per call is a `flags` argument instead (STPY_FLAG_*).  The timing
 *         experiments of tools/ (ablation bits, measured-and-dropped kernel variants, reserved-CU streams, in-kernel
 *         stamps) are NOT in this library: they live in the lab build (make EXPERIMENTS=1 -> libstpy_hip_lab.so);
 *       * one sticky device error word per caller stream (part of the 64 bytes above), read by stpy_async_status;
 *   - exported symbols: exactly the functions declared in this header (the build hides everything else);
 *   - return value: 0 = ok, <0 = invalid argument (-(index of the argument), 1-based) or
 *     -1000-hipError for a failed launch; numerical failure of the factorisation is reported
 *     through the device word `info_dev` (0 = ok, j>0 = leading minor j not positive definite),
 *     exactly LAPACK's potrf convention, so the host decides when to synchronise and read it;
 *   - an empty problem (an output with a zero dimension: no test points, no rows) returns 0 without
 *     looking at any pointer -- empty tensors have null data pointers.
 */
#ifndef STPY_HIP_H
#define STPY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

enum { STPY_F64 = 0, STPY_F32 = 1 };

/* stationary / dot-product kernel families; lengthscales arrive as inv_ls[k] = 1/ell_k */
enum {
	STPY_K_SE = 0,        /* kappa * exp(-r^2/2)                      kernels.py:368-398, :552-583 (ard) */
	STPY_K_MATERN12 = 1,  /* kappa * exp(-r)                          kernels.py:844-845                 */
	STPY_K_MATERN32 = 2,  /* kappa * (1+sqrt3 r) exp(-sqrt3 r)        kernels.py:846-848                 */
	STPY_K_MATERN52 = 3,  /* kappa * (1+sqrt5 r+5r^2/3) exp(-sqrt5 r) kernels.py:849-851, :946-962       */
	STPY_K_LINEAR = 4,    /* kappa * <b_j, a_i> + offset              kernels.py:300-320                 */
	STPY_K_POLY = 5       /* kappa * (<b_j, a_i> + offset)^p          kernels.py:744-761 (offset = 1 there);
	                         the degree p (1..64) rides above the family byte: kind = STPY_K_POLY | (p << 8) */
};

/* per-call flags of stpy_potrf / stpy_trsm_right_lt */
enum {
	STPY_FLAG_BESIDE_UPDATE = 1   /* the call is enqueued while another stream's trailing update occupies the chip
	                                 (multi-GPU panel look-ahead): its small K = 128 products take the 32 KiB-LDS
	                                 kernels that fit on a CU beside one update workgroup, not the 128 KiB one-volley
	                                 kernel that would wait for a whole CU to drain */
};

/* how a kernel evaluation is combined into `out` -- the + and * kernel algebra of kernels.py:146-157 */
enum { STPY_OUT_SET = 0, STPY_OUT_ADD = 1, STPY_OUT_MUL = 2 };

const char* stpy_version(void);
const char* stpy_last_error_string(void);

/*
 * Gram matrix, replaces KernelFunction.kernel(a, b) for one kernel item (kernels.py:136-159):
 *   out[j*ldo + i] (op)= kappa * phi(|| (b_j - a_i)[cols] * inv_ls ||) (+ offset for LINEAR)
 *                        + diag_add * [i == j]
 * a: n x lda, b: q x ldb, out: q x n  (orientation (|b|,|a|) as kernels.py:393).
 * cols: device int32[d] column subset ("group", kernels.py:387-388) or NULL for 0..d-1.
 * inv_ls: device array of d elements of `dtype`.  lower_only != 0 writes only i <= j blocks
 * (used when a == b feeds the Cholesky).  diag_add carries s^2 of gauss_procc.py:151-163.
 * work: NULL, or stpy_gram_workspace_bytes(dtype, n, q, d) bytes of scratch.  With a workspace the
 * inner products run on the MFMA contraction with the kernel function fused into its epilogue
 * (all kinds except MATERN12, which always uses the direct-difference tile kernel).
 */
int64_t stpy_gram_workspace_bytes(int dtype, int64_t n, int64_t q, int d);
int stpy_gram(int kind, int dtype,
              const void* a, int64_t n, int64_t lda,
              const void* b, int64_t q, int64_t ldb,
              int d, const int32_t* cols, const void* inv_ls,
              double kappa, double offset, double diag_add,
              int lower_only, int combine,
              void* out, int64_t ldo, void* work, int64_t work_bytes, void* stream);

/* k(x_i, x_i) for i < m -- replaces the per-point Python loop of gauss_procc.py:347 */
int stpy_gram_diag(int kind, int dtype, const void* x, int64_t m, int64_t ldx,
                   int d, const int32_t* cols, const void* inv_ls,
                   double kappa, double offset, int combine, void* out, void* stream);

/*
 * Blocked right-looking Cholesky, A = L L^T in place in the lower triangle (the strict upper
 * triangle is scratch).  Replaces torch.linalg.cholesky (estimator.py:35) and stands in for
 * lstsq / lu_factor / slogdet (gauss_procc.py:370-378, :634).
 * winv: ceil(n/128) blocks of 128x128 elements (winv_elems >= stpy_potrf_winv_elems(n), else -21); receives
 *       inverse(L_cc) of every 128x128 diagonal block (reused by the triangular solves below).
 * work: stpy_potrf_workspace_bytes(dtype, n, nb) bytes.   nb: outer panel width, multiple of 128
 *       (0 = library default).   info_dev: device int32.
 *       The workspace holds two panel buffers of n x (widest panel) elements.  With nb = 0 the widest panel follows the size:
 *       256 / 512 / 1024 columns up to 2048 / 16 384 / 32 768 rows and 2048 above (since library 0.3: the K = 2048 trailing updates),
 *       i.e. 0.54 GB at n = 32 768, 2.1 GB at n = 65 536 and 4.3 GB at n = 131 072 in fp64 -- on top of the in-place matrix.  In fp32 the
 *       workspace also holds the three bf16 planes of one panel (6 more bytes per panel element: 14 instead of 8; 1.9 GB at n = 65 536).  ALWAYS
 *       size it by the query for the same (dtype, n, nb): a buffer sized by an older build's formula is refused with -20, not overrun.
 */
int64_t stpy_potrf_workspace_bytes(int dtype, int64_t n, int nb);
int64_t stpy_potrf_winv_elems(int64_t n);
int stpy_potrf(int dtype, int64_t n, void* A, int64_t lda, void* winv, int64_t winv_elems,
               void* work, int64_t work_bytes, int nb, int flags, int32_t* info_dev, void* stream);

/* B <- B L^-T for B: m x n row-major (rows = right-hand sides).  With B = K* (M x N) this is
 * V^T = (L^-1 K*^T)^T of the variance term, gauss_procc.py:378,392.  From 2048 rows on: recursive
 * halving of the column range (one long product per split, no workspace).  Fewer rows: panels of nb
 * columns (nb = 0: library default), right-looking, or left-looking when n >= 32768 and `work`
 * (stpy_trsm_workspace_bytes, may be NULL; 0 bytes when not needed) is given, which lets the long panel
 * products run as several K passes so that the latency-bound diagonal blocks of the next panel overlap them. */
int64_t stpy_trsm_workspace_bytes(int dtype, int64_t m, int64_t n, int nb);
int stpy_trsm_right_lt(int dtype, int64_t m, int64_t n, const void* L, int64_t ldl,
                       const void* winv, int64_t winv_elems, void* B, int64_t ldb, int nb, int flags,
                       void* work, int64_t work_bytes, void* stream);

/*
 * Gradient of the evidence (estimator.py:156-190 drives it through
 * autograd in the reference):  d/dtheta [1/2 y^T K^-1 y + w/2 log det K] = 1/2 tr((w K^-1 - alpha alpha^T) dK/dtheta).
 *
 * stpy_potri: Kinv (n x n, lower triangle written) <- (L L^T)^-1 from the factor; work: n x n
 *   elements of scratch (receives L^-T).  2 n^3/3 flop on the MFMA GEMM.
 * stpy_lml_weight: H <- (weight * Kinv - alpha alpha^T) o F, Kinv the full symmetric K^-1 (n x n; Kinv == NULL or
 *   Kinv == H: in place over H; otherwise Kinv is only read, so several kernel terms share one inverse without a
 *   copy); F_ij is the factor of d k(x_i,x_j) / d lengthscale_m = F_ij u_m^2 / lengthscale_m
 *   (u = scaled coordinate difference) for the kernel family `kind` (SE, MATERN12/32/52).
 *   work: stpy_gram_workspace_bytes(dtype, n, n, d).  The per-coordinate sums sum_ij H_ij u_m^2 then
 *   follow from H [Xs | 1] (one stpy_gemm_nt) -- see stpy_amd/continuous_processes/gauss_procc.py.
 */
int stpy_potri(int dtype, int64_t n, const void* L, int64_t ldl, const void* winv, int64_t winv_elems,
               void* Kinv, int64_t ldk, void* work, int64_t work_bytes, void* stream);
int stpy_lml_weight(int kind, int dtype, const void* x, int64_t n, int64_t ldx, int d,
                    const int32_t* cols, const void* inv_ls, double kappa, double weight,
                    const void* alpha, const void* Kinv, int64_t ldk, void* H, int64_t ldh,
                    void* work, int64_t work_bytes, void* stream);

/* out = L^-1 y (trans = 0) or out = L^-T y (trans = 1); the two together are cholesky_solve,
 * estimator.py:37.  y is used as scratch (destroyed); out must not alias y. */
int stpy_trsv(int dtype, int64_t n, const void* L, int64_t ldl, const void* winv, int64_t winv_elems, void* y,
              void* out, int trans, void* stream);

/* mu[i] = <X_i, z>,  sigma[i] = sqrt(kdiag[i] - <X_i, X_i>)   (X = K* L^-T, z = L^-1 y)
 * gauss_procc.py:381, :391-395.  clamp != 0 clamps the variance at 0 before the sqrt (the
 * reference does not clamp).  clamp == 2: sigma[i] receives the raw <X_i, X_i> instead (partial
 * sums of a column-sharded X, reduced across ranks by the caller).  mu or sigma may be NULL. */
int stpy_predict(int dtype, int64_t m, int64_t n, const void* X, int64_t ldx, const void* z,
                 const void* kdiag, void* mu, void* sigma, int clamp, void* stream);

/* The same epilogue when X is sharded by columns across ranks (multi-GPU): stpy_predict(clamp = 2) yields the local
 * partial sums, the caller all-reduces them, and this finishes in place:  mu[i] *= scale,
 * sigma[i] = sqrt(kdiag[i] - scale * sumsq[i])  (scale = 1 / replicas that took part in the sum).  mu / sigma may be NULL. */
int stpy_predict_finish(int dtype, int64_t m, void* mu, const void* sumsq, const void* kdiag, double scale,
                        void* sigma, int clamp, void* stream);

/* out (op)= src elementwise on an m x n window, then + diag_add on the diagonal: the + / * algebra of kernels.py:146-157
 * for an item that was first summed into scratch (combine: STPY_OUT_SET / _ADD / _MUL).  src == out is allowed (with STPY_OUT_SET:
 * "add diag_add to the diagonal in place"). */
int stpy_combine(int dtype, int64_t m, int64_t n, void* out, int64_t ldo, const void* src, int64_t lds,
                 int combine, double diag_add, void* stream);

/* out2[0] = sum_i log L_ii,  out2[1] = z^T z   (estimator.py:36-38, gauss_procc.py:634-636) */
int stpy_logdet_quad(int dtype, int64_t n, const void* L, int64_t ldl, const void* z,
                     void* out2, void* stream);

/* C (op) A B^T with A: m x k, B: n x k, C: m x n.  mode 0: C = A B^T, 1: C -= A B^T, 2: C += A B^T (slab-wise accumulation
 * of Phi^T Phi in the feature-space normal equations, kernelized_features.py:236-240).
 * lower_only: skip 128x128 tiles strictly above the diagonal (m == n).  ldc must be below 2^25 elements (-10).  This is the MFMA
 * contraction under potrf / trsm; exported for the roofline bench and the full-covariance
 * branch gauss_procc.py:396-399. */
int stpy_gemm_nt(int dtype, int64_t m, int64_t n, int64_t k,
                 const void* A, int64_t lda, const void* B, int64_t ldb,
                 void* C, int64_t ldc, int mode, int lower_only, void* stream);

/* C (op) A A^T on the lower 128x128 tiles (A: n x k; mode as above): the feature-space normal equations V (+)= Phi_slab^T Phi_slab
 * (kernelized_features.py:236-240, torch.mm(Phi.T, Phi)).  Same result as stpy_gemm_nt(A, A, lower_only = 1).  With a
 * workspace of stpy_syrk_workspace_bytes(dtype, n, k) bytes (0 = the shape has no such route: fp64, n < 2048, n % 128, k % 32) an fp32
 * operand is split ONCE into three bf16 planes (6 bytes per element of A) and every output tile reads those, instead of every tile
 * re-splitting its rows (gemm_bf3p.hip); work may be NULL.  Few output tiles with a long K (modes 0 and 2) are also cut along K into chunks that
 * run as one grid and are summed in a fixed order (the query then includes the chunk buffers): those results agree with stpy_gemm_nt to fp32
 * rounding, not bit for bit, and are reproducible from run to run. */
int64_t stpy_syrk_workspace_bytes(int dtype, int64_t n, int64_t k);
int stpy_syrk(int dtype, int64_t n, int64_t k, const void* A, int64_t lda, void* C, int64_t ldc, int mode,
              void* work, int64_t work_bytes, void* stream);

/*
 * The same product when C has few 128x128 tiles but k is long (the left-looking partial sums of the
 * distributed solve, gauss_procc.py:368-378 on a sharded factor): the K range is cut into `passes`
 * pieces that run as separate workgroups into `work` (passes*m*n elements, caller-owned) and are
 * then summed into C in a fixed order.  stpy_gemm_nt_splitk_passes recommends the number of passes
 * (1 = use stpy_gemm_nt).  m <= 8 never needs this: stpy_gemm_nt takes a bandwidth-bound row
 * kernel for such products.
 */
int stpy_gemm_nt_splitk_passes(int64_t m, int64_t n, int64_t k);
int stpy_gemm_nt_splitk(int dtype, int64_t m, int64_t n, int64_t k,
                        const void* A, int64_t lda, const void* B, int64_t ldb,
                        void* C, int64_t ldc, int mode, int passes, void* work, int64_t work_bytes, void* stream);

/*
 * The same contraction on a window of a rank's LOCAL matrix under a 2-D block-cyclic distribution
 * (multi-GPU trailing update): distribution block nb_dist (multiple of 128), process grid pr x pc,
 * this rank (myr, myc); the window starts at local block (i0, j0).  A 128x128 tile in local block
 * (bi, bj) belongs to global block (I, J) = (bi*pr + myr, bj*pc + myc) and is skipped when I < J; inside a diagonal
 * block (I == J) only the tiles on and below that block's own diagonal are touched (the symmetric update needs no more).
 */
int stpy_gemm_nt_bc(int dtype, int64_t m, int64_t n, int64_t k,
                    const void* A, int64_t lda, const void* B, int64_t ldb,
                    void* C, int64_t ldc, int mode,
                    int nb_dist, int pr, int pc, int myr, int myc, int i0, int j0, void* stream);

/* mirror the lower triangle into the upper one (n x n) -- materialises .K after a lower-only Gram */
int stpy_symmetrize_lower(int dtype, int64_t n, void* A, int64_t lda, void* stream);

/* zero the strict upper triangle of A (n x n): the in-place factor of stpy_potrf as a proper lower-triangular operand of
 * stpy_gemm_nt -- L r of the samplers (gauss_procc.py:472-474, kernelized_features.py:328-330) */
int stpy_tril(int dtype, int64_t n, void* A, int64_t lda, void* stream);

/* out2[0] = tr(A) (A: n x n; NULL: 0),  out2[1] = <u, v> (n elements each; u NULL: 0) in a fixed summation order -- the scalar
 * tr(w K^-1) - alpha^T alpha of the noise gradient of the evidence (dK/ds = 2 s I), alpha^T y of GaussianProcess.norm
 * (gauss_procc.py:179-184), tr(V^-1) of KernelizedFeatures.effective_dim (kernelized_features.py:103-106) */
int stpy_trace_dot(int dtype, int64_t n, const void* A, int64_t lda, const void* u, const void* v, void* out2, void* stream);

/* out[k*ldo + i] = x[i*ldx + cols[k]] * inv_ls[k] for k < d (cols NULL: k), and out[d*ldo + i] = 1 when ones_row != 0:
 * [Xs | 1]^T, the "row x K" operand of the evidence gradient's H [Xs | 1] product (see stpy_lml_weight). out: (d + ones_row) x n. */
int stpy_scaled_points_t(int dtype, const void* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                         void* out, int64_t ldo, int ones_row, void* stream);

/* The last step of the evidence gradient w.r.t. the lengthscales of one kernel term: with P = H [Xs | 1] (n x (d+1), ldp >= d+1),
 *   acc[pidx[k]] += inv_ls[k] * sum_i ( xs_ik^2 P_id - xs_ik P_ik ),   xs_ik = x[i*ldx + cols[k]] * inv_ls[k],   k < d
 * ( = inv_ls[k]/2 * sum_ij H_ij (xs_ik - xs_jk)^2 ).  pidx: device int32[d], the parameter slot of coordinate k (all zero for an
 * isotropic 'gamma'; NULL: k); acc is accumulated in coordinate order by one workgroup, so the result is reproducible. */
int stpy_lml_grad_reduce(int dtype, const void* x, int64_t n, int64_t ldx, int d, const int32_t* cols, const void* inv_ls,
                         const void* P, int64_t ldp, const int32_t* pidx, void* acc, void* stream);

/* The same for a full-covariance item (kernels.py:464-549: z = x[:, cols] cov (n x p), then SE / Matern of |z_i - z_j|; the reference differentiates
 * it by autograd).  With H formed by stpy_lml_weight on the mapped points z (unit lengthscales) and P = H [Z | 1] (n x (p+1)):
 *   out[a * p + m] -= sum_i x[i*ldx + cols[a]] * (P[i][p] * z[i*ldz + m] - P[i][m]),   a < dg, m < p      (= d evidence / d cov[a][m])
 * One workgroup, fixed order. */
int stpy_lml_grad_cov_reduce(int dtype, const void* x, int64_t n, int64_t ldx, int dg, const int32_t* cols,
                             const void* z, int64_t ldz, int p, const void* P, int64_t ldp, void* out, void* stream);

/*
 * Random Fourier features, replaces RFFEmbedding.embed (embedding.py:225-241):
 *   bias == NULL: out[i*ldo + j] = scale * cos(<W_j, x_i>)  for j <  m/2
 *                                  scale * sin(<W_j, x_i>)  for j >= m/2
 *   bias != NULL: out[i*ldo + j] = scale * cos(<W_j, x_i> + bias[j])
 * x: n x ldx (d columns used), W: m x ldw, out: n x m;  scale = sqrt(2/m) * sqrt(kappa).
 * feat_scale != NULL: feature j is additionally multiplied by feat_scale[j] (m elements of `dtype`) -- the sqrt of the
 *   quadrature weights of QuadratureEmbedding.embed / HermiteEmbedding (embedding.py:450-466, :573-602), which pairs
 *   cos and sin of the SAME node: pass W stacked twice and scale = sqrt(kappa).  An odd m is accepted with a bias only
 *   (cosine-only grids pass a zero bias).
 * transposed != 0 writes Phi^T instead (out: m x n, out[j*ldo + i]) -- the "row x K" operand the
 * feature-space normal equations Phi^T Phi need (kernelized_features.py:236-240).
 * work (optional, stpy_rff_workspace_bytes; 0 for most shapes): with it the large fp32 d = 64 shapes take the contraction to
 *   the bf16 matrix cores from an EXACT three-way bf16 split of both fp32 operands (six products, fp32 accumulation: the
 *   dropped terms are below one fp32 rounding) -- the fp32 MFMA shares the SIMD's ALUs with the trig work, the bf16 pipe does
 *   not.  The workspace receives the split W (m * 64 * 6 bytes); NULL keeps the fp32-MFMA kernel.  Undersized: -20.
 */
int64_t stpy_rff_workspace_bytes(int dtype, int64_t n, int d, int64_t m);
int stpy_rff_embed(int dtype, const void* x, int64_t n, int64_t ldx, int d,
                   const void* W, int64_t ldw, int64_t m, const void* bias, const void* feat_scale, double scale,
                   void* out, int64_t ldo, int transposed, void* work, int64_t work_bytes, void* stream);

/*
 * Launch profiler (bench.py's live roofline numbers).  While enabled, HIP events are recorded on
 * the launch stream around every MFMA GEMM / diagonal-block launch.  tag: 0 = trailing SYRK
 * update of potrf, 1 = panel GEMMs of potrf, 2 = GEMMs of stpy_trsm_right_lt, 3 = 128x128
 * diagonal-block kernel, 4 = direct stpy_gemm_nt calls.  read() returns the summed event time
 * (ms), the summed algorithmic flops and the number of launches of that tag.
 */
void stpy_profile_enable(int enable);
/* Errors only the device can detect after a call has returned: waits for `stream`, returns 0, or 1 when a hand-off wait of the
 * one-launch vector solve (stpy_trsv) gave up -- that solve's output is then NaN from the affected block on, so every quantity
 * derived from it is NaN as well -- and clears the word; < 0: the query itself failed. */
int stpy_async_status(void* stream);
/* Route switches (process-wide; see "state kept by the library" at the top -- the shipped host code never writes them).
 * Unknown keys are ignored; stpy_tune_get returns -1 for them.
 *   5  block-solve algorithm: 0 auto (recursive halving with strip leaves), 1 right-looking sweep, 2 left-looking with K passes,
 *      3.. recursive with leaves of 128 << (value - 3) columns
 *   8  K = 128 products with at most this many 64 x 64 tiles take the one-volley kernel (768; 0 never)
 *   9  fp32 RFF route: 1 streaming kernel for large d = 64 shapes (bf16-split form when a workspace is passed) + tile kernel for the
 *      other d = 32 / 64 shapes; 5 the same but always the fp32-MFMA streaming kernel; 2 tile kernel only; 0 GEMM epilogue only
 *   16 vector solves as one dataflow launch for n a multiple of 128 (1; 0: the chain of per-block launches)
 *   17 leaf width of the recursive block solve that runs as one strip launch (1 = the default, 512; 128 / 256 / 512 / 1024; 0 = off)
 *   26 fp32 products: aligned plain / lower-triangular products of at least this many 128 x 128 tiles run on the bf16 matrix cores
 *      from an exact three-way split of both operands (64; 0 = always the fp32 MFMA kernels)
 *   28 fp64 Gram fill: 1 = the dedicated fill kernel for aligned overwriting fills (three small workgroups per CU), 0 = always the
 *      fused epilogue of the MFMA GEMM
 *   30 plain / lower-only products (both types) of at most this many 128 x 128 tiles (and K >= 64) run as 32 x 128 slivers, four
 *      times the workgroups of the tile kernels (3200; 0 = never) -- the small trailing updates at the end of every factorisation
 *   32 fp32 factorisation: 1 = each finished panel is split ONCE into three bf16 planes in the workspace and its trailing updates of
 *      2048 rows and more run from those planes (gemm_bf3p.hip); 0 = every tile of an update splits its operands on the fly (key 26's
 *      kernel).  Both give bit-identical factors.
 * The lab build (libstpy_hip_lab.so) adds the experiment knobs listed in csrc/common.h (STPY_KNOB_LIST). */
void stpy_tune(int key, int value);
/* current value of a switch (-1: unknown key), so a caller can restore what it changed */
int stpy_tune_get(int key);
int stpy_profile_read(int tag, double* total_ms, double* total_flops, int64_t* launches);
/* union of the launch intervals of all tags in tagmask (bit t = tag t): overlapping launches counted once */
int stpy_profile_read_union(int tagmask, double* busy_ms, double* total_flops, int64_t* launches);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
