// GP-prior KL on the device (reference elbo_functions.py:196-285, kernels of
// GP_model.py:27-116), fp64 throughout: the structured / fused pieces are hand-written here, the plain batched
// [L,M,M] x [L,M,M] and [L,B,M] x [L,M,M] products in between stay library GEMMs on the host side.
//
//   k_gp_kernel_matrix   additive product kernels for all latent dimensions: K[l][i][j] (+ jitter on the diagonal)
//   k_gp_chol_inv        batched Cholesky + inverse + log-determinant of SPD matrices up to 128 x 128, in LDS
//   k_gp_subject_fwd     one workgroup per (subject, latent): K0_st, B_st = K1_st + sigma^2 I (padded subjects get an
//                        identity block), Cholesky, inverse, the partial sums A, B, C, D1 of the bound, iB*Ks, iB*a,
//                        and the gradients of the bound w.r.t. the encoder outputs (mu, log_var) as fp32 [B, L]
//   k_gp_subject_bwd     gradients of the bound w.r.t. B_st and K0_st, chained into the kernel hyper-parameters
//   k_gp_param_grad      chain rule from a gradient matrix dL/dK[l][i][j] into scales, lengthscales and the second
//                        argument's (inducing) points
#include "common.h"

#define GP_MIN_LOG (-16.0)

__device__ __forceinline__ double gp_softplus(double t) { return t > 30.0 ? t : log1p(exp(t)); }
__device__ __forceinline__ double gp_sigmoid(double t) { return 1.0 / (1.0 + exp(-t)); }
__device__ __forceinline__ double gp_positive(double raw) { return exp(GP_MIN_LOG + gp_softplus(raw - GP_MIN_LOG)); }   // GP_model.py:57,85

// value of one term (without / with its scale) between covariate rows xa and xb, for latent dimension l
__device__ __forceinline__ double gp_term_value(const hlvae_gp_kernel& k, int t, const double* __restrict__ prm, int L, int l,
                                                const double* xa, const double* xb, bool with_scale) {
    double v = 1.0;
    for (int f = 0; f < k.n_factors[t]; ++f) {
        const int dim = k.dim[t][f];
        const double a = xa[dim], b = xb[dim];
        if (k.kind[t][f] == HLVAE_GP_CAT) v *= (a == b) ? 1.0 : 0.0;                  // GP_model.py:40-41
        else if (k.kind[t][f] == HLVAE_GP_BIN) v *= (a + b == 2.0) ? 1.0 : 0.0;       // :32-33
        else {
            const double ls = gp_positive(prm[(size_t)k.ls_slot[t][f] * L + l]);
            const double d = a - b;
            v *= exp(-d * d / (2.0 * ls * ls));                                       // :64-69
        }
    }
    return with_scale ? v * gp_positive(prm[(size_t)k.scale_slot[t] * L + l]) : v;
}
__device__ __forceinline__ double gp_kernel_value(const hlvae_gp_kernel& k, const double* __restrict__ prm, int L, int l,
                                                  const double* xa, const double* xb) {
    double s = 0.0;
    for (int t = 0; t < k.n_terms; ++t) s += gp_term_value(k, t, prm, L, l, xa, xb, true);
    return s;
}

__global__ void k_gp_kernel_matrix(hlvae_gp_kernel k, const double* __restrict__ prm, int L, int Q,
                                   const double* __restrict__ x1, int n1, int per_latent1, const double* __restrict__ x2,
                                   int n2, int per_latent2, double jitter, double* __restrict__ out) {
    const long total = (long)L * n1 * n2;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int j = (int)(e % n2), i = (int)((e / n2) % n1), l = (int)(e / ((long)n1 * n2));
        const double* xa = x1 + ((size_t)(per_latent1 ? l : 0) * n1 + i) * Q;
        const double* xb = x2 + ((size_t)(per_latent2 ? l : 0) * n2 + j) * Q;
        double v = gp_kernel_value(k, prm, L, l, xa, xb);
        if (i == j) v += jitter;
        out[e] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// batched Cholesky / inverse / logdet, one workgroup per matrix, matrix resident in LDS (row stride N + 1)
// ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gp_chol_inv(const double* __restrict__ A, int N, double* __restrict__ inv,
                                                     double* __restrict__ logdet, int* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* a = reinterpret_cast<double*>(smem);          // [N][N+1]: lower = L, strict upper (transposed) = L^-1
    double* dinv = a + (size_t)N * (N + 1);               // [N] diagonal of L^-1
    const int ld = N + 1, tid = threadIdx.x;
    const double* src = A + (size_t)blockIdx.x * N * N;
    for (int e = tid; e < N * N; e += 256) a[(e / N) * ld + e % N] = src[e];
    __syncthreads();
    // right-looking Cholesky on the lower triangle
    for (int k = 0; k < N; ++k) {
        if (tid == 0) {
            const double d = a[k * ld + k];
            if (!(d > 0.0) && fail != nullptr) atomicExch(fail, 1);
            a[k * ld + k] = sqrt(d);
        }
        __syncthreads();
        const double dk = a[k * ld + k];
        for (int i = k + 1 + tid; i < N; i += 256) a[i * ld + k] /= dk;
        __syncthreads();
        const int n = N - k - 1;
        for (int e = tid; e < n * n; e += 256) {
            const int i = k + 1 + e / n, j = k + 1 + e % n;
            if (j <= i) a[i * ld + j] -= a[i * ld + k] * a[j * ld + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < N; ++k) s += log(a[k * ld + k]);
        logdet[blockIdx.x] = 2.0 * s;
    }
    // L^-1 by forward substitution, column j by thread j: x_i stored at a[j][i] (i > j), x_j in dinv[j]
    for (int j = tid; j < N; j += 256) {
        const double xj = 1.0 / a[j * ld + j];
        dinv[j] = xj;
        for (int i = j + 1; i < N; ++i) {
            double s = a[i * ld + j] * xj;
            for (int k = j + 1; k < i; ++k) s += a[i * ld + k] * a[j * ld + k];
            a[j * ld + i] = -s / a[i * ld + i];
        }
    }
    __syncthreads();
    // A^-1 = L^-T L^-1:  inv[i][j] = sum_{k >= max(i,j)} Linv[k][i] Linv[k][j]
    double* dst = inv + (size_t)blockIdx.x * N * N;
    for (int e = tid; e < N * N; e += 256) {
        const int i = e / N, j = e % N;
        if (j > i) continue;
        double s = 0.0;
        for (int k = i; k < N; ++k) {
            const double li = (k == i) ? dinv[i] : a[i * ld + k];
            const double lj = (k == j) ? dinv[j] : a[j * ld + k];
            s += li * lj;
        }
        dst[i * N + j] = s;
        dst[j * N + i] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// per (subject, latent) block.  T <= 32 rows per subject (padded), 256 threads.
// LDS: xs [T][Q], B [T][T+1], iB [T][T+1], K0 [T][T+1], tmp [T][T+1]
// ------------------------------------------------------------------------------------------------------------
#define GP_TMAX 32

__device__ __forceinline__ void gp_chol_inv_small(double* b, double* ib, int T, int ld, int tid, double* logdet_out) {
    // Cholesky of b (lower, in place) then ib = b^-1 (full, symmetric); T <= 32, block of 256 threads
    for (int k = 0; k < T; ++k) {
        if (tid == 0) b[k * ld + k] = sqrt(b[k * ld + k]);
        __syncthreads();
        const double dk = b[k * ld + k];
        if (tid > k && tid < T) b[tid * ld + k] /= dk;
        __syncthreads();
        const int n = T - k - 1;
        for (int e = tid; e < n * n; e += 256) {
            const int i = k + 1 + e / n, j = k + 1 + e % n;
            if (j <= i) b[i * ld + j] -= b[i * ld + k] * b[j * ld + k];
        }
        __syncthreads();
    }
    if (tid == 0) {
        double s = 0.0;
        for (int k = 0; k < T; ++k) s += log(b[k * ld + k]);
        *logdet_out = 2.0 * s;
    }
    // L^-1 column j by thread j into the strict upper triangle of b (transposed), diagonal kept in ib's diagonal for now
    if (tid < T) {
        const int j = tid;
        const double xj = 1.0 / b[j * ld + j];
        ib[j * ld + j] = xj;
        for (int i = j + 1; i < T; ++i) {
            double s = b[i * ld + j] * xj;
            for (int k = j + 1; k < i; ++k) s += b[i * ld + k] * b[j * ld + k];
            b[j * ld + i] = -s / b[i * ld + i];
        }
    }
    __syncthreads();
    double r[4];
    int cnt = 0;
    for (int e = tid; e < T * T; e += 256, ++cnt) {           // T*T <= 1024 -> at most 4 per thread
        const int i = e / T, j = e % T;
        const int hi = i > j ? i : j;
        double s = 0.0;
        for (int k = hi; k < T; ++k) {
            const double li = (k == i) ? ib[i * ld + i] : b[i * ld + k];
            const double lj = (k == j) ? ib[j * ld + j] : b[j * ld + k];
            s += li * lj;
        }
        r[cnt] = s;
    }
    __syncthreads();
    cnt = 0;
    for (int e = tid; e < T * T; e += 256, ++cnt) ib[(e / T) * ld + e % T] = r[cnt];
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gp_subject_fwd(
    hlvae_gp_kernel k0, hlvae_gp_kernel k1, const double* __restrict__ prm, int L, int Q, const double* __restrict__ x,
    const double* __restrict__ noise, const int32_t* __restrict__ idx, int T, const double* __restrict__ Kxz, int Bn, int M,
    const double* __restrict__ resid, const float* __restrict__ lv, double c,
    double* __restrict__ iB_out, double* __restrict__ K0_out, double* __restrict__ V_out, double* __restrict__ v_out,
    double* __restrict__ part, float* __restrict__ g_mu, float* __restrict__ g_lv) {
    __shared__ double xs[GP_TMAX * 8];
    __shared__ double bm[GP_TMAX * (GP_TMAX + 1)], ib[GP_TMAX * (GP_TMAX + 1)], k0m[GP_TMAX * (GP_TMAX + 1)];
    __shared__ int rows[GP_TMAX];
    __shared__ double red[4][4];
    __shared__ double ldet;
    const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, ld = T + 1;
    if (tid < T) rows[tid] = idx[(size_t)s * T + tid];
    __syncthreads();
    for (int e = tid; e < T * Q; e += 256) {
        const int t = e / Q, r = rows[t];
        xs[e] = r >= 0 ? x[(size_t)r * Q + e % Q] : 0.0;
    }
    __syncthreads();
    const double nz = noise[l];
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        const bool ok = rows[i] >= 0 && rows[j] >= 0;
        double kb = 0.0, k0v = 0.0;
        if (ok) {
            kb = gp_kernel_value(k1, prm, L, l, xs + i * Q, xs + j * Q) + (i == j ? nz : 0.0);     // elbo_functions.py:249-250
            k0v = gp_kernel_value(k0, prm, L, l, xs + i * Q, xs + j * Q);                           // :248
        } else if (i == j) {
            kb = 1.0;                                             // padded row: identity block
        }
        bm[i * ld + j] = kb;
        k0m[i * ld + j] = k0v;
    }
    __syncthreads();
    gp_chol_inv_small(bm, ib, T, ld, tid, &ldet);
    // mask the inverse to the valid block, write iB and K0_st
    double d1 = 0.0;
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        const bool ok = rows[i] >= 0 && rows[j] >= 0;
        const double v = ok ? ib[i * ld + j] : 0.0;
        ib[i * ld + j] = v;
        const size_t o = (((size_t)s * L + l) * T + i) * T + j;
        iB_out[o] = v;
        K0_out[o] = k0m[i * ld + j];
        d1 += v * k0m[i * ld + j];                                // sum(iB * K0_st)  (:259)
    }
    __syncthreads();
    // v = iB a, A = a.v, Bt = sum diag(iB) e^lv, g_mu, g_lv
    double pa = 0.0, pb = 0.0;
    if (tid < T && rows[tid] >= 0) {
        const int i = tid, r = rows[i];
        double acc = 0.0;
        for (int j = 0; j < T; ++j)
            if (rows[j] >= 0) acc += ib[i * ld + j] * resid[(size_t)l * Bn + rows[j]];
        const double ai = resid[(size_t)l * Bn + r];
        const double e = exp((double)lv[(size_t)r * L + l]);
        v_out[(size_t)l * Bn + r] = acc;
        pa = ai * acc;                                            // (:256)
        pb = ib[i * ld + i] * e;                                  // (:257)
        g_mu[(size_t)r * L + l] = (float)(-c * acc);             // d/dmu  of  c/2 a^T iB a  with a = pred - mu
        g_lv[(size_t)r * L + l] = (float)(c * 0.5 * (ib[i * ld + i] * e - 1.0));
    }
    // V = iB Ks  [T][M] -> V_out[l][row][:]
    for (int e = tid; e < T * M; e += 256) {
        const int i = e / M, mcol = e % M;
        if (rows[i] < 0) continue;
        double acc = 0.0;
        for (int j = 0; j < T; ++j)
            if (rows[j] >= 0) acc += ib[i * ld + j] * Kxz[((size_t)l * Bn + rows[j]) * M + mcol];
        V_out[((size_t)l * Bn + rows[i]) * M + mcol] = acc;
    }
    // block reduction of the three partial sums
    pa = wave_sum_d(pa); pb = wave_sum_d(pb); d1 = wave_sum_d(d1);
    if ((tid & 63) == 0) { red[0][tid >> 6] = pa; red[1][tid >> 6] = pb; red[2][tid >> 6] = d1; }
    __syncthreads();
    if (tid == 0) {
        double* p = part + ((size_t)s * L + l) * 4;
        p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        p[2] = ldet;                                              // C contribution: log det B_st  (:258)
        p[3] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

// gradient of the bound w.r.t. B_st and K0_st of one (subject, latent), chained into the hyper-parameters:
//   G_K0 = c/2 iB ;   G_B = c/2 [ iB - v v^T - iB diag(e^lv) iB - iB K0 iB + Y V^T ],  Y = V (iK - Q)
__global__ __launch_bounds__(256) void k_gp_subject_bwd(
    hlvae_gp_kernel k0, hlvae_gp_kernel k1, const double* __restrict__ prm, int L, int Q, const double* __restrict__ x,
    const int32_t* __restrict__ idx, int T, int Bn, int M, const double* __restrict__ iB_in, const double* __restrict__ K0_in,
    const double* __restrict__ V, const double* __restrict__ v, const double* __restrict__ Y, const float* __restrict__ lv,
    double c, int n_slots, double* __restrict__ gprm) {
    __shared__ double xs[GP_TMAX * 8];
    __shared__ double ib[GP_TMAX * (GP_TMAX + 1)], k0m[GP_TMAX * (GP_TMAX + 1)], w[GP_TMAX * (GP_TMAX + 1)],
        gb[GP_TMAX * (GP_TMAX + 1)];
    __shared__ int rows[GP_TMAX];
    __shared__ double vv[GP_TMAX], ee[GP_TMAX];
    __shared__ double gacc[32];                                   // per-slot gradient accumulators of this block
    const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, ld = T + 1;
    if (tid < T) rows[tid] = idx[(size_t)s * T + tid];
    if (tid < 32) gacc[tid] = 0.0;
    __syncthreads();
    for (int e = tid; e < T * Q; e += 256) {
        const int t = e / Q, r = rows[t];
        xs[e] = r >= 0 ? x[(size_t)r * Q + e % Q] : 0.0;
    }
    if (tid < T) {
        const int r = rows[tid];
        vv[tid] = r >= 0 ? v[(size_t)l * Bn + r] : 0.0;
        ee[tid] = r >= 0 ? exp((double)lv[(size_t)r * L + l]) : 0.0;
    }
    for (int e = tid; e < T * T; e += 256) {
        const size_t o = (((size_t)s * L + l) * T + e / T) * T + e % T;
        ib[(e / T) * ld + e % T] = iB_in[o];
        k0m[(e / T) * ld + e % T] = K0_in[o];
    }
    __syncthreads();
    // w = diag(e) + K0   then  gb = iB w iB
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        w[i * ld + j] = k0m[i * ld + j] + (i == j ? ee[i] : 0.0);
    }
    __syncthreads();
    double tmp[4];
    int cnt = 0;
    for (int e = tid; e < T * T; e += 256, ++cnt) {               // tmp = iB w
        const int i = e / T, j = e % T;
        double a = 0.0;
        for (int k = 0; k < T; ++k) a += ib[i * ld + k] * w[k * ld + j];
        tmp[cnt] = a;
    }
    __syncthreads();
    cnt = 0;
    for (int e = tid; e < T * T; e += 256, ++cnt) w[(e / T) * ld + e % T] = tmp[cnt];
    __syncthreads();
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        double a = 0.0;
        for (int k = 0; k < T; ++k) a += w[i * ld + k] * ib[k * ld + j];          // (iB w iB)[i][j]
        double yv = 0.0;                                                          // (Y V^T)[i][j]
        if (rows[i] >= 0 && rows[j] >= 0) {
            const double* yi = Y + ((size_t)l * Bn + rows[i]) * M;
            const double* vj = V + ((size_t)l * Bn + rows[j]) * M;
            for (int m = 0; m < M; ++m) yv += yi[m] * vj[m];
        }
        gb[i * ld + j] = 0.5 * c * (ib[i * ld + j] - vv[i] * vv[j] - a + yv);
    }
    __syncthreads();
    // chain rule into scales / lengthscales: G_B with the k1 terms, G_K0 = c/2 iB with the k0 terms
    for (int e = tid; e < T * T; e += 256) {
        const int i = e / T, j = e % T;
        if (rows[i] < 0 || rows[j] < 0) continue;
        for (int pass = 0; pass < 2; ++pass) {
            const hlvae_gp_kernel& kk = pass == 0 ? k1 : k0;
            const double g = pass == 0 ? gb[i * ld + j] : 0.5 * c * ib[i * ld + j];
            if (g == 0.0) continue;
            for (int t = 0; t < kk.n_terms; ++t) {
                const double tv = gp_term_value(kk, t, prm, L, l, xs + i * Q, xs + j * Q, true);
                if (tv == 0.0) continue;
                const int ss = kk.scale_slot[t];
                atomicAdd(&gacc[ss], g * tv * gp_sigmoid(prm[(size_t)ss * L + l] - GP_MIN_LOG));
                for (int f = 0; f < kk.n_factors[t]; ++f)
                    if (kk.kind[t][f] == HLVAE_GP_RBF) {
                        const int sl = kk.ls_slot[t][f];
                        const double raw = prm[(size_t)sl * L + l], ls = gp_positive(raw);
                        const double d = xs[i * Q + kk.dim[t][f]] - xs[j * Q + kk.dim[t][f]];
                        atomicAdd(&gacc[sl], g * tv * d * d / (ls * ls) * gp_sigmoid(raw - GP_MIN_LOG));
                    }
            }
        }
    }
    __syncthreads();
    if (tid < n_slots && gacc[tid] != 0.0) atomicAdd(gprm + (size_t)tid * L + l, gacc[tid]);
}

// chain rule from G[l][i][j] = dL/dK(x1_i, x2_j) into the hyper-parameters and the points of the SECOND argument.
// both_args != 0: x1 and x2 are the same per-latent point set (K0zz): G is used as given for the hyper-parameters and
// as G + G^T for the points.  One workgroup per (latent, column j).
__global__ __launch_bounds__(256) void k_gp_param_grad(hlvae_gp_kernel k, const double* __restrict__ prm, int L, int Q,
                                                       const double* __restrict__ x1, int n1, int per_latent1,
                                                       const double* __restrict__ x2, int n2, int both_args,
                                                       const double* __restrict__ G, int n_slots,
                                                       double* __restrict__ gprm, double* __restrict__ gx2) {
    __shared__ double gacc[32];
    __shared__ double gz[8];
    const int j = blockIdx.x, l = blockIdx.y, tid = threadIdx.x;
    if (tid < 32) gacc[tid] = 0.0;
    if (tid < 8) gz[tid] = 0.0;
    __syncthreads();
    const double* xb = x2 + ((size_t)l * n2 + j) * Q;
    double lacc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) lacc[q] = 0.0;
    double lz[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n1; i += 256) {
        const double* xa = x1 + ((size_t)(per_latent1 ? l : 0) * n1 + i) * Q;
        const double g = G[((size_t)l * n1 + i) * n2 + j];
        const double gsym = both_args ? g + G[((size_t)l * n1 + j) * n2 + i] : g;
        for (int t = 0; t < k.n_terms; ++t) {
            const double tv = gp_term_value(k, t, prm, L, l, xa, xb, true);
            if (tv == 0.0) continue;
            const int ss = k.scale_slot[t];
            const double sg = gp_sigmoid(prm[(size_t)ss * L + l] - GP_MIN_LOG);
#pragma unroll
            for (int q = 0; q < 32; ++q)
                if (q == ss) lacc[q] += g * tv * sg;
            for (int f = 0; f < k.n_factors[t]; ++f)
                if (k.kind[t][f] == HLVAE_GP_RBF) {
                    const int sl = k.ls_slot[t][f], dim = k.dim[t][f];
                    const double raw = prm[(size_t)sl * L + l], ls = gp_positive(raw);
                    const double d = xa[dim] - xb[dim];
                    const double gl = g * tv * d * d / (ls * ls) * gp_sigmoid(raw - GP_MIN_LOG);
#pragma unroll
                    for (int q = 0; q < 32; ++q)
                        if (q == sl) lacc[q] += gl;
                    const double gzq = gsym * tv * d / (ls * ls);                      // d k / d x2[dim]
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q == dim) lz[q] += gzq;
                }
        }
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) {
        if (q >= n_slots) break;
        const double sum = wave_sum_d(lacc[q]);
        if ((tid & 63) == 0 && sum != 0.0) atomicAdd(&gacc[q], sum);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        if (q >= Q) break;
        const double sum = wave_sum_d(lz[q]);
        if ((tid & 63) == 0 && sum != 0.0) atomicAdd(&gz[q], sum);
    }
    __syncthreads();
    if (tid < n_slots && gacc[tid] != 0.0) atomicAdd(gprm + (size_t)tid * L + l, gacc[tid]);
    if (tid < Q && gx2 != nullptr && gz[tid] != 0.0) atomicAdd(gx2 + ((size_t)l * n2 + j) * Q + tid, gz[tid]);
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
static int gp_check_kernel(const hlvae_gp_kernel* k, int n_slots, int Q) {
    HL_REQUIRE(k && k->n_terms >= 0 && k->n_terms <= HLVAE_GP_MAX_TERMS, HLVAE_EINVAL, "gp kernel: n_terms");
    HL_REQUIRE(n_slots <= 32 && Q <= 8, HLVAE_EINVAL, "gp kernel: at most 32 hyper-parameter rows and 8 covariates");
    for (int t = 0; t < k->n_terms; ++t) {
        HL_REQUIRE(k->n_factors[t] >= 1 && k->n_factors[t] <= HLVAE_GP_MAX_FACTORS && k->scale_slot[t] >= 0 &&
                       k->scale_slot[t] < n_slots, HLVAE_EINVAL, "gp kernel: term %d", t);
        for (int f = 0; f < k->n_factors[t]; ++f) {
            HL_REQUIRE(k->dim[t][f] >= 0 && k->dim[t][f] < Q, HLVAE_EINVAL, "gp kernel: covariate index");
            if (k->kind[t][f] == HLVAE_GP_RBF)
                HL_REQUIRE(k->ls_slot[t][f] >= 0 && k->ls_slot[t][f] < n_slots, HLVAE_EINVAL, "gp kernel: lengthscale row");
        }
    }
    return 0;
}

extern "C" {

int hlvae_gp_kernel_matrix(const hlvae_gp_kernel* k, const double* prm, int n_slots, int L, int Q, const double* x1, int n1,
                           int per_latent1, const double* x2, int n2, int per_latent2, double jitter, double* out,
                           hlvae_stream s) {
    if (int rc = gp_check_kernel(k, n_slots, Q)) return rc;
    HL_REQUIRE(prm && x1 && x2 && out && L > 0 && n1 > 0 && n2 > 0, HLVAE_EINVAL, "gp_kernel_matrix: bad arguments");
    const long total = (long)L * n1 * n2;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    HL_PROF("gp_kernel_matrix", (hipStream_t)s);
    k_gp_kernel_matrix<<<blocks, 256, 0, (hipStream_t)s>>>(*k, prm, L, Q, x1, n1, per_latent1, x2, n2, per_latent2, jitter, out);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_chol_inv(const double* A, int n, int N, double* inv, double* logdet, int* fail, hlvae_stream s) {
    HL_REQUIRE(A && inv && logdet && n > 0 && N > 0 && N <= 128, HLVAE_EINVAL, "gp_chol_inv: N=%d (max 128)", N);
    const size_t smem = ((size_t)N * (N + 1) + N) * sizeof(double);
    static size_t attr_max = 48 * 1024;
    if (smem > attr_max) {
        HL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gp_chol_inv), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)smem));
        attr_max = smem;
    }
    HL_PROF("gp_chol_inv", (hipStream_t)s);
    k_gp_chol_inv<<<n, 256, smem, (hipStream_t)s>>>(A, N, inv, logdet, fail);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_subject_fwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* prm, int n_slots, int L, int Q,
                         const double* x, const double* noise, const int32_t* idx, int S, int T, const double* Kxz, int B,
                         int M, const double* resid, const float* lv, double c, double* iB, double* K0s, double* V,
                         double* v, double* part, float* g_mu, float* g_lv, hlvae_stream s) {
    if (int rc = gp_check_kernel(k0, n_slots, Q)) return rc;
    if (int rc = gp_check_kernel(k1, n_slots, Q)) return rc;
    HL_REQUIRE(T >= 1 && T <= GP_TMAX && S >= 1 && Q <= 8, HLVAE_ESHAPE, "gp_subject_fwd: T=%d (max %d)", T, GP_TMAX);
    HL_PROF("gp_subject_fwd", (hipStream_t)s);
    k_gp_subject_fwd<<<dim3(S, L), 256, 0, (hipStream_t)s>>>(*k0, *k1, prm, L, Q, x, noise, idx, T, Kxz, B, M, resid, lv, c, iB,
                                                           K0s, V, v, part, g_mu, g_lv);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_subject_bwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* prm, int n_slots, int L, int Q,
                         const double* x, const int32_t* idx, int S, int T, int B, int M, const double* iB, const double* K0s,
                         const double* V, const double* v, const double* Y, const float* lv, double c, double* gprm,
                         hlvae_stream s) {
    if (int rc = gp_check_kernel(k0, n_slots, Q)) return rc;
    if (int rc = gp_check_kernel(k1, n_slots, Q)) return rc;
    HL_REQUIRE(T >= 1 && T <= GP_TMAX && S >= 1, HLVAE_ESHAPE, "gp_subject_bwd: T=%d", T);
    HL_PROF("gp_subject_bwd", (hipStream_t)s);
    k_gp_subject_bwd<<<dim3(S, L), 256, 0, (hipStream_t)s>>>(*k0, *k1, prm, L, Q, x, idx, T, B, M, iB, K0s, V, v, Y, lv, c,
                                                           n_slots, gprm);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_param_grad(const hlvae_gp_kernel* k, const double* prm, int n_slots, int L, int Q, const double* x1, int n1,
                        int per_latent1, const double* x2, int n2, int both_args, const double* G, double* gprm, double* gx2,
                        hlvae_stream s) {
    if (int rc = gp_check_kernel(k, n_slots, Q)) return rc;
    HL_REQUIRE(!both_args || n1 == n2, HLVAE_ESHAPE, "gp_param_grad: both_args needs a square matrix");
    HL_PROF("gp_param_grad", (hipStream_t)s);
    k_gp_param_grad<<<dim3(n2, L), 256, 0, (hipStream_t)s>>>(*k, prm, L, Q, x1, n1, per_latent1, x2, n2, both_args, G, n_slots,
                                                            gprm, gx2);
    HL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
