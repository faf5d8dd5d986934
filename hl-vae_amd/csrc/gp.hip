// GP-prior KL on the device (reference elbo_functions.py:196-285, kernels of
// GP_model.py:27-116), fp64 throughout: the structured / fused pieces are hand-written here, the plain batched
// [L,M,M] x [L,M,M] and [L,B,M] x [L,M,M] products in between stay library GEMMs on the host side.
//
//   k_gp_transform       hyper-parameter planes from the raw values, once per step: pos = exp(-16 + softplus(raw + 16))
//                        (GP_model.py:57,85), dpos = sigmoid(raw + 16) (d pos / d raw = pos * dpos), il2 = 1 / pos^2
//   k_gp_kernel_matrix   additive product kernels for all latent dimensions: K[l][i][j] (+ jitter on the diagonal)
//   k_gp_spd_inv         batched SPD inverse + log-determinant up to 128 x 128: in-place Gauss-Jordan with the whole
//                        matrix in REGISTERS (8 x 8 elements per lane of a 16 x 16 thread grid); only the pivot row and
//                        column travel through (double-buffered) LDS, one barrier per pivot
//   k_gp_subject_fwd     one workgroup per (subject, latent): K0_st, B_st = K1_st + sigma^2 I (padded subjects get an
//                        identity block), its inverse by the same register Gauss-Jordan, the partial sums A, B, C, D1 of
//                        the bound, iB*Ks, iB*a, and the gradients of the bound w.r.t. the encoder outputs (mu, log_var)
//   k_gp_subject_bwd     gradients of the bound w.r.t. B_st and K0_st, chained into the kernel hyper-parameters
//   k_gp_param_grad      chain rule from a gradient matrix dL/dK[l][i][j] into scales, lengthscales and the second
//                        argument's (inducing) points
//   k_gp_bound           every scalar reduction of the bound (elbo_functions.py:268-285) in one launch
//   k_gp_adam            torch.optim.Adam on the flat fp64 arena [hyper-parameters | inducing points]
#include <stdlib.h>
#include "common.h"

#pragma clang diagnostic ignored "-Wpass-failed"       // (unroll requests on run-time trip counts, as in mid.hip)

typedef __attribute__((ext_vector_type(4))) double f64x4_t;     // one v_mfma_f64_16x16x4_f64 accumulator fragment
#define GP_TMAX 32
#define GP_MMAX 128
#define GP_MAX_RBF 2                                        // RBF factors per term (validated by the launchers)
#define GP_XS 9                                             // padded covariate row in LDS

// clock64() phase profile of the per-subject kernels (tools/gp_phases.py): thread 0 of a workgroup stamps the phases into a
// buffer when one is installed (hlvae_debug_gp_clk); one scalar load per workgroup otherwise
#define GP_CLK_PH 10
#define GP_CLK_WG 2048
__device__ long long* g_gpclk = nullptr;
#define GP_CLK(kernel, ph)                                                                                                  \
    do {                                                                                                                     \
        if (clkp != nullptr && threadIdx.x == 0) {                                                                          \
            const int wg_ = blockIdx.x + gridDim.x * blockIdx.y;                                                             \
            if (wg_ < GP_CLK_WG) clkp[((kernel) * GP_CLK_WG + wg_) * GP_CLK_PH + (ph)] = clock64();                       \
        }                                                                                                                    \
    } while (0)

// ------------------------------------------------------------------------------------------------------------
// covariance terms.  Hyper-parameters of one latent dimension are hoisted into registers once per thread.
// ------------------------------------------------------------------------------------------------------------
// (templates on the kernel's shape, round 3: NT = terms, NR = RBF factors per term.  The generic 8 x 2 form costs a lane 24
//  hyper-parameter and 24 accumulator doubles per kernel -- k_gp_subject_bwd, with two kernels, sat at 204 VGPRs; BASELINE
//  configs[4]'s kernels have <= 3 terms of <= 1 RBF factor and run the <4, 1> instances.)
template <int NT = HLVAE_GP_MAX_TERMS, int NR = GP_MAX_RBF>
struct GpHypT {
    double sc[NT];
    double il2[NT][NR];
};
template <int NT = HLVAE_GP_MAX_TERMS, int NR = GP_MAX_RBF>
struct GpAccT {                                             // per-lane gradient accumulators of one additive kernel
    double ts[NT];                                          // d / d scale      (x scale)
    double tl[NT][NR];                                      // d / d lengthscale (x lengthscale)
};
typedef GpHypT<> GpHyp;
typedef GpAccT<> GpAcc;

// does the <4, 1> instance cover this kernel?
static bool gp_kernel_small(const hlvae_gp_kernel* k) {
    if (k->n_terms > 4) return false;
    for (int t = 0; t < k->n_terms; ++t) {
        int r = 0;
        for (int f = 0; f < k->n_factors[t]; ++f) r += k->kind[t][f] == HLVAE_GP_RBF;
        if (r > 1) return false;
    }
    return true;
}

template <int NT, int NR>
__device__ __forceinline__ void gp_hoist(const hlvae_gp_kernel& k, const double* __restrict__ hyp, int n_slots, int L, int l,
                                         GpHypT<NT, NR>& h) {
    const double* pos = hyp;
    const double* il2 = hyp + (size_t)2 * n_slots * L;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        h.sc[t] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) h.il2[t][r] = 0.0;
        if (t < k.n_terms) {
            h.sc[t] = pos[(size_t)k.scale_slot[t] * L + l];
            int r = 0;
            for (int f = 0; f < k.n_factors[t]; ++f)
                if (k.kind[t][f] == HLVAE_GP_RBF) {
                    const double v = il2[(size_t)k.ls_slot[t][f] * L + l];
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
                        if (rr == r) h.il2[t][rr] = v;
                    ++r;
                }
        }
    }
}
template <int NT, int NR>
__device__ __forceinline__ void gp_acc_zero(GpAccT<NT, NR>& a) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        a.ts[t] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) a.tl[t][r] = 0.0;
    }
}

// term t (compile-time index after unrolling) between covariate rows xa and xb: value with its scale, 0 when an
// indicator factor is off; d[r] = differences of the (up to NR) RBF factors, e[r] = d^2 / ls^2
template <int NR>
__device__ __forceinline__ double gp_term(const hlvae_gp_kernel& k, int t, double sc, const double (&il2)[NR], const double* xa,
                                          const double* xb, double (&d)[NR], double (&e)[NR]) {
#pragma unroll
    for (int r = 0; r < NR; ++r) d[r] = 0.0;
    int r = 0;
    bool on = true;
    for (int f = 0; f < k.n_factors[t]; ++f) {
        const int dim = k.dim[t][f], kind = k.kind[t][f];
        const double a = xa[dim], b = xb[dim];
        if (kind == HLVAE_GP_CAT) on = on && (a == b);                    // GP_model.py:40-41
        else if (kind == HLVAE_GP_BIN) on = on && (a + b == 2.0);         // :32-33
        else {
#pragma unroll
            for (int rr = 0; rr < NR; ++rr)
                if (rr == r) d[rr] = a - b;
            ++r;
        }
    }
    double es = 0.0;
#pragma unroll
    for (int rr = 0; rr < NR; ++rr) {
        e[rr] = d[rr] * d[rr] * il2[rr];
        es += e[rr];
    }
    if (!on) return 0.0;
    return r == 0 ? sc : sc * exp(-0.5 * es);                             // :64-69 (product of RBFs = exp of the sum)
}
template <int NT, int NR>
__device__ __forceinline__ double gp_value(const hlvae_gp_kernel& k, const GpHypT<NT, NR>& h, const double* xa, const double* xb) {
    double s = 0.0, d[NR], e[NR];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= k.n_terms) break;
        s += gp_term<NR>(k, t, h.sc[t], h.il2[t], xa, xb, d, e);
    }
    return s;
}
// accumulate g * d k / d (hyper-parameters) of one pair
template <int NT, int NR>
__device__ __forceinline__ void gp_pair_grad(const hlvae_gp_kernel& k, const GpHypT<NT, NR>& h, const double* xa, const double* xb,
                                             double g, GpAccT<NT, NR>& a) {
    double d[NR], e[NR];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= k.n_terms) break;
        const double tv = gp_term<NR>(k, t, h.sc[t], h.il2[t], xa, xb, d, e);
        const double gt = g * tv;
        a.ts[t] += gt;
#pragma unroll
        for (int r = 0; r < NR; ++r) a.tl[t][r] += gt * e[r];
    }
}
// wave reduction of the accumulators, lane 0 adds (x d pos / d raw / pos) into the block's LDS rows gacc[slot]
template <int NT, int NR>
__device__ __forceinline__ void gp_flush(const hlvae_gp_kernel& k, const GpAccT<NT, NR>& a, const double* __restrict__ dpos_l, int L,
                                         double* gacc, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= k.n_terms) break;
        const double s = wave_sum_d(a.ts[t]);
        if (lane == 0 && s != 0.0) atomicAdd(&gacc[k.scale_slot[t]], s * dpos_l[(size_t)k.scale_slot[t] * L]);
        int r = 0;
        for (int f = 0; f < k.n_factors[t]; ++f)
            if (k.kind[t][f] == HLVAE_GP_RBF) {
                double v = 0.0;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) v = rr == r ? a.tl[t][rr] : v;
                v = wave_sum_d(v);
                if (lane == 0 && v != 0.0) atomicAdd(&gacc[k.ls_slot[t][f]], v * dpos_l[(size_t)k.ls_slot[t][f] * L]);
                ++r;
            }
    }
}

__global__ void k_gp_transform(const double* __restrict__ raw, int n, double* __restrict__ hyp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = raw[i] + 16.0;
    const double sp = x > 20.0 ? x : log1p(exp(x));          // torch softplus (beta 1, threshold 20)
    const double p = exp(sp - 16.0);
    hyp[i] = p;
    hyp[(size_t)n + i] = 1.0 / (1.0 + exp(-x));
    hyp[(size_t)2 * n + i] = 1.0 / (p * p);
}

// One workgroup = GP_KM_ROWS rows of x1 against ALL rows of x2 (n2 <= 128) for one latent: both covariate blocks are staged
// in LDS once (round 2 read every covariate of every element from global memory behind an integer division: 43 us alone for the
// 31 MB of K0xz at configs[4]); thread (column j = tid & 127, row phase tid >> 7) keeps x2[j] in registers and walks its rows, so
// a wave's stores are 64 consecutive doubles of one output row.
#define GP_KM_ROWS 32
template <int NT, int NR>
__global__ __launch_bounds__(256) void k_gp_kernel_matrix(hlvae_gp_kernel k, const double* __restrict__ hyp, int n_slots, int L,
                                                          int Q, const double* __restrict__ x1, int n1, int per_latent1,
                                                          const double* __restrict__ x2, int n2, int per_latent2,
                                                          double jitter, double* __restrict__ out, int rows_wg) {
    // rows_wg <= GP_KM_ROWS rows per workgroup: 32 for the batch's K0xz, 8 for K0zz (120 rows: 4 x L workgroups walked 16 rows
    // per thread one after the other -- 26 us on the state update's chain for 0.46 M entries; 15 x L workgroups of 4 rows)
    __shared__ double xs[GP_KM_ROWS * GP_XS], zs[GP_MMAX * GP_XS];
    const int l = blockIdx.y, row0 = blockIdx.x * rows_wg, tid = threadIdx.x;
    const int nrow = min(rows_wg, n1 - row0);
    const double* x1l = x1 + (size_t)(per_latent1 ? l : 0) * n1 * Q;
    const double* x2l = x2 + (size_t)(per_latent2 ? l : 0) * n2 * Q;
    for (int e = tid; e < nrow * Q; e += 256) xs[(e / Q) * GP_XS + e % Q] = x1l[(size_t)row0 * Q + e];
    for (int e = tid; e < n2 * Q; e += 256) zs[(e / Q) * GP_XS + e % Q] = x2l[e];
    GpHypT<NT, NR> h;
    gp_hoist(k, hyp, n_slots, L, l, h);
    __syncthreads();
    const int j = tid & 127, ph = tid >> 7;
    if (j >= n2) return;
    double xb[GP_XS];
#pragma unroll
    for (int q = 0; q < GP_XS - 1; ++q) xb[q] = q < Q ? zs[j * GP_XS + q] : 0.0;
    double* o = out + ((size_t)l * n1 + row0) * n2 + j;
    for (int i = ph; i < nrow; i += 2) {
        double v = gp_value(k, h, xs + i * GP_XS, xb);
        if (row0 + i == j) v += jitter;
        o[(size_t)i * n2] = v;
    }
}

// ------------------------------------------------------------------------------------------------------------
// SPD inverse + logdet: Gauss-Jordan without pivoting, matrix in registers.
// lane (ti, tj) of the 16 x 16 grid owns elements (ti + 16 ii, tj + 16 jj), ii, jj < NB; the matrix is padded with an
// identity block, which the elimination never touches.  One call handles 16 pivots with a compile-time register-block
// index KB (no dynamic register indexing); pivot row / column are published through double-buffered LDS vectors, so
// one barrier per pivot is enough.
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double gp_rcp(double p) {
    double r = __builtin_amdgcn_rcp(p);                      // v_rcp_f64 + two Newton steps (full fp64 precision)
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}
template <int NB, int KB>
__device__ __forceinline__ void gj_pivots(double (&a)[NB][NB], double* row, double* col, double* pv, int N, int ti, int tj) {
    constexpr int W = 16 * NB;
    for (int kr = 0; kr < 16; ++kr) {
        const int k = KB * 16 + kr;
        if (k >= N) return;
        double* rw = row + (k & 1) * W;
        double* cl = col + (k & 1) * W;
        const bool ik = ti == kr, jk = tj == kr;
        if (ik) {
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) rw[tj + 16 * jj] = a[KB][jj];
        }
        if (jk) {
#pragma unroll
            for (int ii = 0; ii < NB; ++ii) cl[ti + 16 * ii] = a[ii][KB];
        }
        __syncthreads();
        const double piv = rw[k];
        const double pk = gp_rcp(piv);
        if (ik && jk) pv[k] = piv;
        double cr[NB], rc[NB], cz[NB], rz[NB];
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) cz[ii] = cr[ii] = cl[ti + 16 * ii];
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) rz[jj] = rc[jj] = rw[tj + 16 * jj] * pk;
        if (ik) cz[KB] = 0.0;                              // the generic update leaves pivot row and column alone
        if (jk) rz[KB] = 0.0;
#pragma unroll
        for (int ii = 0; ii < NB; ++ii)
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) a[ii][jj] = fma(-cz[ii], rz[jj], a[ii][jj]);
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) a[KB][jj] = ik ? rc[jj] : a[KB][jj];            // pivot row:    a[k][j] / p
#pragma unroll
        for (int ii = 0; ii < NB; ++ii) a[ii][KB] = jk ? -cr[ii] * pk : a[ii][KB];      // pivot column: -a[i][k] / p
        if (ik && jk) a[KB][KB] = pk;                                                  // pivot:        1 / p
    }
}

__global__ __launch_bounds__(256) void k_gp_spd_inv(const double* __restrict__ A, int N, double* __restrict__ inv,
                                                    double* __restrict__ logdet, int* __restrict__ fail, int n_neg,
                                                    double* __restrict__ logdet_neg) {
    __shared__ double row[2 * GP_MMAX], col[2 * GP_MMAX];
    __shared__ double pv[GP_MMAX];
    const int tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    const double* src = A + (size_t)blockIdx.x * N * N;
    double a[8][8];
#pragma unroll
    for (int ii = 0; ii < 8; ++ii)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            a[ii][jj] = (i < N && j < N) ? src[(size_t)i * N + j] : (i == j ? 1.0 : 0.0);
        }
    gj_pivots<8, 0>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 1>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 2>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 3>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 4>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 5>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 6>(a, row, col, pv, N, ti, tj);
    gj_pivots<8, 7>(a, row, col, pv, N, ti, tj);
    double* dst = inv + (size_t)blockIdx.x * N * N;
#pragma unroll
    for (int ii = 0; ii < 8; ++ii)
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            if (i < N && j < N) dst[(size_t)i * N + j] = a[ii][jj];
        }
    __syncthreads();
    if (tid < 64) {                                        // log-determinant = sum of log pivots
        double ld = 0.0;
        bool bad = false;
        for (int k = tid; k < N; k += 64) {
            const double p = pv[k];
            bad |= !(p > 0.0);
            ld += log(p);
        }
        ld = wave_sum_d(ld);
        if (bad && fail != nullptr) atomicExch(fail, 1);
        if (tid == 0) {
            logdet[blockIdx.x] = ld;
            if ((int)blockIdx.x < n_neg) logdet_neg[blockIdx.x] = -ld;       // log det of the INVERSE (H_new from iH_new)
        }
    }
}

// (Round 3 built a blocked form of this sweep -- four pivots per phase: P^-1 of the 4 x 4 pivot block in every lane, the scaled
//  pivot rows published by their owners, a rank-4 update of the 8 x 8 register block, 30 phases of two barriers instead of 120
//  pivots -- and dropped it: 105 us against 84 us for this kernel on [iH_new | K0zz] (64 matrices of 120 x 120), and 2.6e-5
//  instead of 5e-7 on the natural-gradient terms of the ill-conditioned (1e8) config-5 matrices.  The redundant 4 x 4 inversion
//  is four DEPENDENT reciprocal chains per phase, and the rank-4 update reads 64 LDS doubles per lane where the rank-1 form reads
//  16: the phase costs 3500 clocks against 4 x 1400.  fp64 MFMA would not help either: v_mfma_f64_16x16x4 runs at the vector
//  FMA rate on MI355X.  A 512-thread form -- 4 x 8 elements per lane, two waves per SIMD, the NEXT pivot's reciprocal computed by
//  its owner beside the block update and published with the row -- measured 86.6 us against 84.5: the sweep's ~200 instructions
//  per pivot are per WAVE (row scaling, selects, LDS traffic), halving the block only halves the 64 FMAs among them.)
// ------------------------------------------------------------------------------------------------------------
// per (subject, latent) block.  T <= 32 rows per subject (padded), M <= 128 inducing points, 256 threads as a
// 16 x 16 grid: lane (ti, tj) owns the 2 x 2 elements (ti + 16 ii, tj + 16 jj) of the T x T blocks.
// ------------------------------------------------------------------------------------------------------------
#define GP_TS (GP_TMAX + 1)

// (Round 3 tried the T x T inverse inside ONE wave -- lane j holds column j in registers, the pivot column arrives by v_readlane as a
//  scalar operand of the FMAs, no LDS, no barrier: k_gp_subject_fwd 74 -> 92 us alone.  Three waves idle while one issues ~3 k dependent
//  readlane / FMA pairs; the 256-thread form with its 20 barriers is the faster one for 20 x 20.)
template <int NT, int NR>
__global__ __launch_bounds__(256) void k_gp_subject_fwd(
    hlvae_gp_kernel k0, hlvae_gp_kernel k1, const double* __restrict__ hyp, int n_slots, int L, int Q,
    const double* __restrict__ x, const double* __restrict__ noise, const int32_t* __restrict__ idx, int T,
    const double* __restrict__ Kxz, int Bn, int M, const double* __restrict__ resid, const float* __restrict__ lv, double c,
    double* __restrict__ iB_out, double* __restrict__ K0_out, double* __restrict__ V_out, double* __restrict__ v_out,
    double* __restrict__ part, float* __restrict__ g_mu, float* __restrict__ g_lv, const double* __restrict__ iKm,
    const float* __restrict__ mu, double* __restrict__ u_acc, double* __restrict__ p1_acc) {
    // iKm != nullptr (training step, round 3): the residual a = K0xz (iK m) - mu of the subject's rows is computed HERE from the
    // staged rows of K0xz (`resid` is then unused: one launch and one 31 MB pass less), and the two matrix^T-vector sums over
    // the batch that followed as launches of their own -- u = K0xz^T v and P1 = V^T mu = Ks^T (iB mu), one streaming pass over
    // K0xz / V each, 76 + 32 us on the natural-gradient chain -- are accumulated per subject into u_acc / p1_acc [L][M] (fp64
    // atomics, zeroed by the caller) from the same LDS tile.
    __shared__ double xs[GP_TMAX * GP_XS];
    __shared__ double vsh[GP_TMAX], wsh[GP_TMAX], mus[GP_TMAX];
    __shared__ double ib[GP_TMAX * GP_TS], kzs[GP_TMAX * GP_TS];
    extern __shared__ __attribute__((aligned(16))) char dsm_fwd[];
    double* ks = reinterpret_cast<double*>(dsm_fwd);      // the subject's rows of K0xz, [T][M] (sized for the actual T, M:
                                                          // 19 KB at T = 20, M = 120 -> five workgroups per CU instead of three)
    __shared__ double gjrow[2 * GP_TMAX], gjcol[2 * GP_TMAX], pv[GP_TMAX], rs[GP_TMAX];
    __shared__ int rows[GP_TMAX];
    __shared__ double red[3][4];
    const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    long long* clkp = g_gpclk;
    GP_CLK(0, 0);
    if (tid < GP_TMAX) rows[tid] = tid < T ? idx[(size_t)s * T + tid] : -1;
    __syncthreads();
    for (int e = tid; e < T * Q; e += 256) {
        const int t = e / Q, r = rows[t];
        xs[t * GP_XS + e % Q] = r >= 0 ? x[(size_t)r * Q + e % Q] : 0.0;
    }
    for (int e0 = tid; e0 < T * M; e0 += 4 * 256) {       // stage Ks (coalesced along M), zero rows for padding;
        double tk[4];                                     // four loads in flight per lane
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = min(e0 + 256 * u, T * M - 1), t = e / M, r = rows[t];
            tk[u] = Kxz[((size_t)l * Bn + (r >= 0 ? r : 0)) * M + e % M];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < T * M) ks[e] = rows[e / M] >= 0 ? tk[u] : 0.0;
        }
    }
    if (iKm == nullptr && tid < T) rs[tid] = rows[tid] >= 0 ? resid[(size_t)l * Bn + rows[tid]] : 0.0;
    if (tid < GP_TMAX) {
        vsh[tid] = wsh[tid] = 0.0;
        mus[tid] = (mu != nullptr && tid < T && rows[tid] >= 0) ? (double)mu[(size_t)rows[tid] * L + l] : 0.0;
    }
    GP_CLK(0, 1);      // staging issued
    GpHypT<NT, NR> h0, h1;
    gp_hoist(k0, hyp, n_slots, L, l, h0);
    gp_hoist(k1, hyp, n_slots, L, l, h1);
    __syncthreads();
    if (iKm != nullptr) {                                         // a[t] = sum_m Ks[t][m] (iK m)[m] - mu[t]: 8 lanes per row
        const int t = tid >> 3, sub = tid & 7;                    // (256 threads = 32 rows x 8)
        double sa = 0.0;
        if (t < T)
            for (int mm = sub; mm < M; mm += 8) sa += ks[t * M + mm] * iKm[(size_t)l * M + mm];
        sa += __shfl_xor(sa, 4, 64);
        sa += __shfl_xor(sa, 2, 64);
        sa += __shfl_xor(sa, 1, 64);
        if (sub == 0 && t < T) rs[t] = rows[t] >= 0 ? sa - mus[t] : 0.0;
    }
    GP_CLK(0, 2);      // staged + residual
    const double nz = noise[l];
    // covariance entries: the kernels are symmetric, so the T (T + 1) / 2 = 210 pairs i <= j are evaluated once, one per
    // thread, into LDS (in the 2 x 2 register blocking of the Gauss-Jordan below 16 threads would evaluate 4 pairs each)
    for (int p = tid; p < T * (T + 1) / 2; p += 256) {
        int i = 0, rem = p;
        while (rem >= T - i) { rem -= T - i; ++i; }
        const int j = i + rem;
        double kb = i == j ? 1.0 : 0.0, kz = 0.0;                 // padded rows: identity block
        if (rows[i] >= 0 && rows[j] >= 0) {
            kb = gp_value(k1, h1, xs + i * GP_XS, xs + j * GP_XS) + (i == j ? nz : 0.0);       // elbo_functions.py:249-250
            kz = gp_value(k0, h0, xs + i * GP_XS, xs + j * GP_XS);                              // :248
        }
        ib[i * GP_TS + j] = kb; ib[j * GP_TS + i] = kb;
        kzs[i * GP_TS + j] = kz; kzs[j * GP_TS + i] = kz;
    }
    __syncthreads();
    GP_CLK(0, 3);      // covariance pairs
    double a[2][2], k0v[2][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            const bool in = i < T && j < T;
            a[ii][jj] = in ? ib[i * GP_TS + j] : (i == j ? 1.0 : 0.0);
            k0v[ii][jj] = in ? kzs[i * GP_TS + j] : 0.0;
        }
    __syncthreads();                                              // ib is reused for the inverse below
    gj_pivots<2, 0>(a, gjrow, gjcol, pv, T, ti, tj);
    gj_pivots<2, 1>(a, gjrow, gjcol, pv, T, ti, tj);
    GP_CLK(0, 4);      // Gauss-Jordan
    // mask the inverse to the valid block, write iB and K0_st
    double d1 = 0.0;
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            if (i < T && j < T) {
                const bool ok = rows[i] >= 0 && rows[j] >= 0;
                const double v = ok ? a[ii][jj] : 0.0;
                ib[i * GP_TS + j] = v;
                const size_t o = (((size_t)s * L + l) * T + i) * T + j;
                iB_out[o] = v;
                K0_out[o] = k0v[ii][jj];
                d1 += v * k0v[ii][jj];                            // sum(iB * K0_st)  (:259)
            }
        }
    __syncthreads();
    GP_CLK(0, 5);      // iB, K0 written
    // v = iB a, A = a.v, Bt = sum diag(iB) e^lv, g_mu, g_lv
    double pa = 0.0, pb = 0.0, pc = 0.0;
    if (tid < T) {
        if (rows[tid] >= 0) {
            const int i = tid, r = rows[i];
            double acc = 0.0, accw = 0.0;
            for (int j = 0; j < T; ++j) { acc += ib[i * GP_TS + j] * rs[j]; accw += ib[i * GP_TS + j] * mus[j]; }
            vsh[i] = acc;                                         // v = iB a
            wsh[i] = accw;                                        // iB mu
            const double e = exp((double)lv[(size_t)r * L + l]);
            v_out[(size_t)l * Bn + r] = acc;
            pa = rs[i] * acc;                                     // (:256)
            pb = ib[i * GP_TS + i] * e;                           // (:257)
            g_mu[(size_t)r * L + l] = (float)(-c * acc);         // d/dmu  of  c/2 a^T iB a  with a = pred - mu
            g_lv[(size_t)r * L + l] = (float)(c * 0.5 * (ib[i * GP_TS + i] * e - 1.0));
        }
        pc = log(pv[tid]);                                        // log det B_st = sum of log pivots (:258)
    }
    GP_CLK(0, 6);      // v, g_mu, g_lv
    // V = iB Ks  [T][M] -> V_out[l][row][:] on the fp64 matrix cores (round 3: as 20 scalar FMAs per output with two LDS reads each
    // this loop was 18 k of a workgroup's 76 k clocks -- LDS bandwidth shared by the four co-resident workgroups): wave w owns the
    // 16-column fragments w, w + 4 of both 16-row halves; A = iB [i][j], B = Ks [j][m], k = j in steps of 4
    {
        const int wave = tid >> 6, lane = tid & 63, g = lane >> 4, q = lane & 15;
        const int nfc = (M + 15) >> 4;
        for (int fc = wave; fc < nfc; fc += 4) {
            f64x4_t acc2[2] = {f64x4_t{0.0, 0.0, 0.0, 0.0}, f64x4_t{0.0, 0.0, 0.0, 0.0}};
            const int mcol = 16 * fc + q;
            for (int k4 = 0; k4 < T; k4 += 4) {
                const int kk = k4 + g;
                const double b = (kk < T && mcol < M) ? ks[kk * M + mcol] : 0.0;
#pragma unroll
                for (int fi = 0; fi < 2; ++fi) {
                    const int i = 16 * fi + q;
                    const double a = (i < T && kk < T) ? ib[i * GP_TS + kk] : 0.0;
                    acc2[fi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2[fi], 0, 0, 0);
                }
            }
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * fi + g + 4 * r;
                    if (i < T && mcol < M && rows[i] >= 0) V_out[((size_t)l * Bn + rows[i]) * M + mcol] = acc2[fi][r];
                }
        }
    }
    GP_CLK(0, 7);      // V = iB Ks
    // block reduction of the partial sums
    pa = wave_sum_d(pa); pb = wave_sum_d(pb); d1 = wave_sum_d(d1); pc = wave_sum_d(pc);
    if ((tid & 63) == 0) { red[0][tid >> 6] = pa; red[1][tid >> 6] = pb; red[2][tid >> 6] = d1; }
    __syncthreads();
    if (u_acc != nullptr && tid < M) {                            // this subject's share of K0xz^T v and of V^T mu = Ks^T (iB mu)
        double su = 0.0, sp = 0.0;
        for (int i = 0; i < T; ++i) {
            const double kv = ks[i * M + tid];
            su += kv * vsh[i];
            sp += kv * wsh[i];
        }
        atomicAdd(u_acc + (size_t)l * M + tid, su);
        atomicAdd(p1_acc + (size_t)l * M + tid, sp);
    }
    GP_CLK(0, 8);      // u / P1 atomics
    if (tid == 0) {
        double* p = part + ((size_t)s * L + l) * 4;
        p[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        p[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        p[2] = pc;                                                // wave 0 holds all T <= 32 pivots
        p[3] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    }
}

// gradient of the bound w.r.t. B_st and K0_st of one (subject, latent), chained into the hyper-parameters:
//   G_K0 = c/2 iB ;   G_B = c/2 [ iB - v v^T - iB diag(e^lv) iB - iB K0 iB + Y V^T ],  Y = V (iK - Q)
template <int NT, int NR>
__global__ __launch_bounds__(256) void k_gp_subject_bwd(
    hlvae_gp_kernel k0, hlvae_gp_kernel k1, const double* __restrict__ hyp, int n_slots, int L, int Q,
    const double* __restrict__ x, const int32_t* __restrict__ idx, int T, int Bn, int M, const double* __restrict__ iB_in,
    const double* __restrict__ K0_in, const double* __restrict__ V, const double* __restrict__ v, const double* __restrict__ Y,
    const float* __restrict__ lv, double c, double* __restrict__ gprm) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int MS = M + 1;                                         // padded: lanes walk different rows at the same m
    double* vs = reinterpret_cast<double*>(dsm);                  // V_s [T][MS]
    double* ys = vs + (size_t)T * MS;                             // Y_s [T][MS]
    const int TS = T + 1;                                         // iB, w [T][TS]: sized for the actual T (with V_s / Y_s 49 KB at
    double* ib = ys + (size_t)T * MS;                             // T = 20, M = 120 -> three workgroups per CU instead of two)
    double* w = ib + (size_t)T * TS;
    double* yvt = w + (size_t)T * TS;                             // (Y V^T) [T][TS], from the matrix cores
    __shared__ double xs[GP_TMAX * GP_XS];
    __shared__ int rows[GP_TMAX];
    __shared__ double vv[GP_TMAX], ee[GP_TMAX];
    __shared__ double gacc[32];                                   // per-row gradient accumulators of this block
    const int s = blockIdx.x, l = blockIdx.y, tid = threadIdx.x, ti = tid >> 4, tj = tid & 15;
    long long* clkp = g_gpclk;
    GP_CLK(1, 0);
    if (tid < GP_TMAX) rows[tid] = tid < T ? idx[(size_t)s * T + tid] : -1;
    if (tid < 32) gacc[tid] = 0.0;
    __syncthreads();
    for (int e = tid; e < T * Q; e += 256) {
        const int t = e / Q, r = rows[t];
        xs[t * GP_XS + e % Q] = r >= 0 ? x[(size_t)r * Q + e % Q] : 0.0;
    }
    for (int e0 = tid; e0 < T * M; e0 += 4 * 256) {         // four elements per pass: 8 global loads in flight per lane
        double tv[4], ty[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            const int t = min(e, T * M - 1) / M, m = min(e, T * M - 1) - t * M, r = rows[t];
            const size_t o = ((size_t)l * Bn + (r >= 0 ? r : 0)) * M + m;
            tv[u] = V[o];
            ty[u] = Y[o];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 256 * u;
            if (e < T * M) {
                const int t = e / M, m = e - t * M;
                const bool ok = rows[t] >= 0;
                vs[t * MS + m] = ok ? tv[u] : 0.0;
                ys[t * MS + m] = ok ? ty[u] : 0.0;
            }
        }
    }
    if (tid < T) {
        const int r = rows[tid];
        vv[tid] = r >= 0 ? v[(size_t)l * Bn + r] : 0.0;
        ee[tid] = r >= 0 ? exp((double)lv[(size_t)r * L + l]) : 0.0;
    }
    __syncthreads();
    GP_CLK(1, 1);      // V_s, Y_s staged
    // w = diag(e) + K0
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            if (i < T && j < T) {
                const size_t o = (((size_t)s * L + l) * T + i) * T + j;
                ib[i * TS + j] = iB_in[o];
                w[i * TS + j] = K0_in[o] + (i == j ? ee[i] : 0.0);
            }
        }
    __syncthreads();
    double tmp[2][2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {                           // tmp = iB w
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            double acc = 0.0;
            if (i < T && j < T)
                for (int k = 0; k < T; ++k) acc += ib[i * TS + k] * w[k * TS + j];
            tmp[ii][jj] = acc;
        }
    __syncthreads();
#pragma unroll
    for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int i = ti + 16 * ii, j = tj + 16 * jj;
            if (i < T && j < T) w[i * TS + j] = tmp[ii][jj];
        }
    __syncthreads();
    // (Y V^T)[i][j] = sum_m Y_s[i][m] V_s[j][m] on the fp64 matrix cores: one 16 x 16 fragment per wave, k = m in steps of 4.  (As two
    // scalar chains of 60 LDS-read pairs per (i, j) thread the dot products were 19 k of a workgroup's 62 k clocks: rows of V_s
    // 121 doubles apart collide in the banks.)
    {
        const int wave = tid >> 6, lane = tid & 63, g4 = lane >> 4, q = lane & 15;
        const int i = 16 * (wave & 1) + q, j = 16 * (wave >> 1) + q;
        const double* yp = ys + (size_t)min(i, T - 1) * MS;
        const double* vp = vs + (size_t)min(j, T - 1) * MS;
        // four independent accumulation chains, 16 k per pass: the LDS reads of a pass are issued together and a dependent MFMA
        // is four products away (one chain of 30 dependent MFMAs behind two LDS reads each was 14 k clocks)
        f64x4_t ac[4];
#pragma unroll
        for (int c_ = 0; c_ < 4; ++c_) ac[c_] = f64x4_t{0.0, 0.0, 0.0, 0.0};
        for (int m16 = 0; m16 < M; m16 += 16) {
            double a_[4], b_[4];
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) {
                const int m = m16 + 4 * c_ + g4;
                a_[c_] = (i < T && m < M) ? yp[min(m, M - 1)] : 0.0;
                b_[c_] = (j < T && m < M) ? vp[min(m, M - 1)] : 0.0;
            }
#pragma unroll
            for (int c_ = 0; c_ < 4; ++c_) ac[c_] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_[c_], b_[c_], ac[c_], 0, 0, 0);
        }
        f64x4_t acc4;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc4[r] = (ac[0][r] + ac[1][r]) + (ac[2][r] + ac[3][r]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * (wave & 1) + g4 + 4 * r, col = 16 * (wave >> 1) + q;
            if (row < T && col < T) yvt[row * TS + col] = acc4[r];
        }
    }
    __syncthreads();
    GP_CLK(1, 2);      // iB w
    GpHypT<NT, NR> h0, h1;
    gp_hoist(k0, hyp, n_slots, L, l, h0);
    gp_hoist(k1, hyp, n_slots, L, l, h1);
    GpAccT<NT, NR> a0, a1;
    gp_acc_zero(a0);
    gp_acc_zero(a1);
    // the T x T pairs dealt round-robin to the 256 threads (T = 20: at most 2 pairs per thread; the 16 x 16 + 2 x 2 blocking
    // of the stages above would give 16 threads 4 pairs each and this stage was 70 % of the kernel: 73 k clocks)
    // Both gradient matrices are symmetric (iB, iB (E + K0) iB and Y V^T = V (iK - Q) V^T are) and so are the kernels: only
    // the pairs i <= j are evaluated, off-diagonal ones with weight 2 -- T (T + 1) / 2 = 210 pairs, one per thread.
    for (int p = tid; p < T * (T + 1) / 2; p += 256) {
        int i = 0, rem = p;
        while (rem >= T - i) { rem -= T - i; ++i; }
        const int j = i + rem;
        const double sym = i == j ? 1.0 : 2.0;
        if (rows[i] >= 0 && rows[j] >= 0) {
            double acc = 0.0;                                                 // (iB w iB)[i][j]
            for (int k = 0; k < T; ++k) acc += w[i * TS + k] * ib[k * TS + j];
            const double yv = yvt[i * TS + j];
            if (p == 0) GP_CLK(1, 5);  // (thread 0: its dots done)
            const double ibv = ib[i * TS + j];
            const double g1 = sym * 0.5 * c * (ibv - vv[i] * vv[j] - acc + yv);     // dL / dB_st
            const double g0 = sym * 0.5 * c * ibv;                                  // dL / dK0_st
            gp_pair_grad(k1, h1, xs + i * GP_XS, xs + j * GP_XS, g1, a1);
            gp_pair_grad(k0, h0, xs + i * GP_XS, xs + j * GP_XS, g0, a0);
        }
    }
    GP_CLK(1, 3);      // pair loop
    const double* dpos_l = hyp + (size_t)n_slots * L + l;
    constexpr int NVK = NT + NT * NR;                             // accumulators per kernel
    if (NVK <= 8 && (size_t)T * MS >= (size_t)NVK * 256) {       // (uniform)
        // every thread's 2 NVK accumulators -> LDS image [value][thread] over V_s / Y_s (dead now), 16 threads sum one value's 256
        // entries, one shuffle tree per value instead of one per value AND wave (gp_flush: 16 x 12 ds_bpermute per wave, 12 k clocks)
        __shared__ int slot_tab[16];                              // hyper-parameter row of every accumulator (uniform code, -1: unused)
        auto fill = [&](const hlvae_gp_kernel& kk, int base) {
            for (int t = 0; t < NT; ++t) {
                int sl = t < kk.n_terms ? kk.scale_slot[t] : -1, lsl[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) lsl[r] = -1;
                if (t < kk.n_terms) {
                    int rcount = 0;
                    for (int f = 0; f < kk.n_factors[t]; ++f)
                        if (kk.kind[t][f] == HLVAE_GP_RBF) {
#pragma unroll
                            for (int r = 0; r < NR; ++r)
                                if (r == rcount) lsl[r] = kk.ls_slot[t][f];
                            ++rcount;
                        }
                }
                if (tid == 0) {
                    slot_tab[base + t] = sl;
#pragma unroll
                    for (int r = 0; r < NR; ++r) slot_tab[base + NT + t * NR + r] = lsl[r];
                }
            }
        };
        fill(k1, 0);
        fill(k0, NVK);
        __syncthreads();
        double* red = vs;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            red[(size_t)t * 256 + tid] = a1.ts[t];
            red[(size_t)(NVK + t) * 256 + tid] = a0.ts[t];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                red[(size_t)(NT + t * NR + r) * 256 + tid] = a1.tl[t][r];
                red[(size_t)(NVK + NT + t * NR + r) * 256 + tid] = a0.tl[t][r];
            }
        }
        __syncthreads();
        for (int v0 = 0; v0 < 2 * NVK; v0 += 16) {
            const int vi = v0 + (tid >> 4), part = tid & 15;
            double sum = 0.0;
            if (vi < 2 * NVK) {
#pragma unroll
                for (int qq = 0; qq < 16; ++qq) sum += red[(size_t)vi * 256 + part + 16 * qq];
            }
            sum += __shfl_xor(sum, 8, 64);
            sum += __shfl_xor(sum, 4, 64);
            sum += __shfl_xor(sum, 2, 64);
            sum += __shfl_xor(sum, 1, 64);
            if (part == 0 && vi < 2 * NVK && sum != 0.0) {
                const int slot = slot_tab[vi];
                if (slot >= 0) atomicAdd(&gacc[slot], sum * dpos_l[(size_t)slot * L]);
            }
        }
    } else {
        gp_flush(k1, a1, dpos_l, L, gacc, tid & 63);
        gp_flush(k0, a0, dpos_l, L, gacc, tid & 63);
    }
    GP_CLK(1, 4);      // flush
    __syncthreads();
    if (tid < n_slots && gacc[tid] != 0.0) atomicAdd(gprm + (size_t)tid * L + l, gacc[tid]);
}

// chain rule from G[l][i][j] = dL/dK(x1_i, x2_j) into the hyper-parameters and the points of the SECOND argument.
// both_args != 0: x1 and x2 are the same per-latent point set (K0zz) and G arrives symmetrised (G + G^T): half of it
// drives the hyper-parameters, all of it the points (which sit in both argument positions).
// grid (ceil(n2 / 64), row chunks of 64, L), block 256 = 4 waves x 64 columns: a lane owns one column j and walks 16
// rows; covariates of the row chunk and of the 64 columns are staged in LDS.
// Round 3: specialised on the kernel's shape -- NT = terms (<= 4 or <= 8), NR = RBF factors per term (1 or 2) -- and on the rows a
// lane walks (ROWS / 4).  The generic form carried 40 fp64 accumulators + 24 hyper-parameters per lane for 8 terms x 2 RBF
// factors (232 VGPRs: two waves per SIMD, and the K0zz launch -- 128 workgroups of 16 dependent pairs per lane -- ran at one
// wave per SIMD for 50 us); configs[4]'s K0 has 3 terms of one RBF factor each: 92 VGPRs.  The lane's own column covariates sit
// in registers per (term, factor) instead of being re-read from LDS for every pair.
template <int NT, int NR, int ROWS>
__global__ __launch_bounds__(256) void k_gp_param_grad(hlvae_gp_kernel k, const double* __restrict__ hyp, int n_slots, int L,
                                                       int Q, const double* __restrict__ x1, int n1, int per_latent1,
                                                       const double* __restrict__ x2, int n2, int both_args,
                                                       const double* __restrict__ G, double* __restrict__ gprm,
                                                       double* __restrict__ gx2, const double* __restrict__ vrow,
                                                       const double* __restrict__ wcol, double cg) {
    // vrow != nullptr: G holds Y = V (iK - Q) and the gradient w.r.t. K0xz is formed here, g = cg (v_i w_j - Y_ij) (what
    // k_gp_gkxz wrote out as a 31 MB matrix for this kernel to read back: one launch and 62 MB less on chain C)
    constexpr int RPL = ROWS / 4;                                 // rows per lane
    __shared__ double gacc[32];
    __shared__ double xs[ROWS * GP_XS], xbs[64 * GP_XS];
    __shared__ double zred[4][64][GP_XS];
    const int tid = threadIdx.x, lane = tid & 63, sub = tid >> 6;
    const int j = blockIdx.x * 64 + lane, l = blockIdx.z;
    const int row0 = blockIdx.y * ROWS, nrow = min(ROWS, n1 - row0);
    const int jc = min(j, n2 - 1);
    const int i_lo = sub * RPL;
    double gv[RPL];
#pragma unroll
    for (int r = 0; r < RPL; ++r) {                               // all gradient loads of this lane in flight before anything else
        const int i = min(i_lo + r, nrow - 1);
        gv[r] = G[((size_t)l * n1 + row0 + i) * n2 + jc];
    }
    if (vrow != nullptr) {
        const double wj = wcol[(size_t)l * n2 + jc];
#pragma unroll
        for (int r = 0; r < RPL; ++r) gv[r] = cg * (vrow[(size_t)l * n1 + row0 + min(i_lo + r, nrow - 1)] * wj - gv[r]);
    }
    if (tid < 32) gacc[tid] = 0.0;
    for (int e = tid; e < nrow * Q; e += 256)
        xs[(e / Q) * GP_XS + e % Q] = x1[((size_t)(per_latent1 ? l : 0) * n1 + row0) * Q + e];
    for (int e = tid; e < 64 * Q; e += 256) {
        const int col = blockIdx.x * 64 + e / Q;
        xbs[(e / Q) * GP_XS + e % Q] = col < n2 ? x2[((size_t)l * n2 + col) * Q + e % Q] : 0.0;
    }
    for (int q = 0; q < GP_XS; ++q) zred[sub][lane][q] = 0.0;
    // hyper-parameters and the (uniform) shape of every term: RBF dims rd[t][.] (-1: none), indicator dims through k
    double sc[NT], il2[NT][NR];
    int rd[NT][NR];
    const double* pos = hyp;
    const double* il2p = hyp + (size_t)2 * n_slots * L;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        sc[t] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) { il2[t][r] = 0.0; rd[t][r] = -1; }
        if (t < k.n_terms) {
            sc[t] = pos[(size_t)k.scale_slot[t] * L + l];
            int r = 0;
            for (int f = 0; f < k.n_factors[t]; ++f)
                if (k.kind[t][f] == HLVAE_GP_RBF) {
#pragma unroll
                    for (int rr = 0; rr < NR; ++rr)
                        if (rr == r) { il2[t][rr] = il2p[(size_t)k.ls_slot[t][f] * L + l]; rd[t][rr] = k.dim[t][f]; }
                    ++r;
                }
        }
    }
    double ts[NT], tl[NT][NR], tz[NT][NR];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ts[t] = 0.0;
#pragma unroll
        for (int r = 0; r < NR; ++r) tl[t][r] = tz[t][r] = 0.0;
    }
    __syncthreads();
    // this lane's column: RBF coordinates per (term, factor) and the whole covariate row for the indicator factors, in registers
    double zb[NT][NR], xb[GP_XS - 1];
#pragma unroll
    for (int q = 0; q < GP_XS - 1; ++q) xb[q] = xbs[lane * GP_XS + q];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < NR; ++r) zb[t][r] = rd[t][r] >= 0 ? xbs[lane * GP_XS + rd[t][r]] : 0.0;
    const double gs = both_args ? 0.5 : 1.0;
#pragma unroll
    for (int r = 0; r < RPL; ++r) {
        const int i = i_lo + r;
        if (i >= nrow || j >= n2) continue;
        const double* xa = xs + i * GP_XS;
        const double gsym = gv[r], g = gs * gsym;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t >= k.n_terms) break;
            bool on = true;
            for (int f = 0; f < k.n_factors[t]; ++f) {            // indicator factors (uniform loop; the lane's side from registers)
                const int kind = k.kind[t][f];
                if (kind == HLVAE_GP_RBF) continue;
                const int dim = k.dim[t][f];
                double b = xb[0];
#pragma unroll
                for (int q = 1; q < GP_XS - 1; ++q) b = dim == q ? xb[q] : b;
                const double a = xa[dim];
                on = on && (kind == HLVAE_GP_CAT ? (a == b) : (a + b == 2.0));       // GP_model.py:40-41, :32-33
            }
            double d[NR], e[NR], es = 0.0;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                d[rr] = rd[t][rr] >= 0 ? xa[rd[t][rr]] - zb[t][rr] : 0.0;
                e[rr] = d[rr] * d[rr] * il2[t][rr];
                es += e[rr];
            }
            const double tv = !on ? 0.0 : (rd[t][0] < 0 ? sc[t] : sc[t] * exp(-0.5 * es));   // :64-69
            const double gt = g * tv, gz = gsym * tv;
            ts[t] += gt;
#pragma unroll
            for (int rr = 0; rr < NR; ++rr) {
                tl[t][rr] += gt * e[rr];
                tz[t][rr] += gz * d[rr] * il2[t][rr];             // d k / d x2[dim] = k (xa - xb) / ls^2
            }
        }
    }
    // hyper-parameters: wave sums, lane 0 adds (x d pos / d raw) into the block's LDS rows
    const double* dpos_l = hyp + (size_t)n_slots * L + l;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t >= k.n_terms) break;
        const double s0 = wave_sum_d(ts[t]);
        if (lane == 0 && s0 != 0.0) atomicAdd(&gacc[k.scale_slot[t]], s0 * dpos_l[(size_t)k.scale_slot[t] * L]);
        int r = 0;
        for (int f = 0; f < k.n_factors[t]; ++f)
            if (k.kind[t][f] == HLVAE_GP_RBF) {
                double v = 0.0;
#pragma unroll
                for (int rr = 0; rr < NR; ++rr) v = rr == r ? tl[t][rr] : v;
                v = wave_sum_d(v);
                if (lane == 0 && v != 0.0) atomicAdd(&gacc[k.ls_slot[t][f]], v * dpos_l[(size_t)k.ls_slot[t][f] * L]);
                ++r;
            }
    }
    // inducing points: per-lane LDS row indexed by covariate, summed over the 4 row sub-chunks
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int rr = 0; rr < NR; ++rr)
            if (rd[t][rr] >= 0) zred[sub][lane][rd[t][rr]] += tz[t][rr];
    __syncthreads();
    if (tid < n_slots && gacc[tid] != 0.0) atomicAdd(gprm + (size_t)tid * L + l, gacc[tid]);
    if (gx2 != nullptr && sub == 0 && j < n2) {
        for (int q = 0; q < Q; ++q) {
            const double sum = zred[0][lane][q] + zred[1][lane][q] + zred[2][lane][q] + zred[3][lane][q];
            if (sum != 0.0) atomicAdd(gx2 + ((size_t)l * n2 + j) * Q + q, sum);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// batched N x N fp64 products of the natural-gradient algebra (N = inducing points <= 128):
//     C[l] = alpha A[l] B[l] + beta D[l]                 (row-major, dense; D may be null or alias C)
// on the fp64 matrix cores (v_mfma_f64_16x16x4_f64: A/B one double per lane, A[row l&15][k l>>4], B[k l>>4][col l&15];
// C/D col = lane&15, row = (lane>>4) + 4 reg).  One workgroup = one 32 x 32 tile of one matrix (64 x 64: 15.3 us per
// product, 32 x 64: 14.6, 32 x 32: 13.7 -- 512 workgroups for 32 latents); its 32 x K and K x 32 operand panels (K = N <= 128)
// sit whole in LDS, each wave owns one 16 x 16 fragment.  The library's batched GEMM takes
// 11-32 us for these 120 x 120 x 120 x 32 products (3.5 TFLOP/s); seven of them per step were 18 % of the GP step.
// ------------------------------------------------------------------------------------------------------------
#define GP_BMM_T 32                                         // tile columns
#define GP_BMM_R 32                                         // tile rows
__global__ __launch_bounds__(256) void k_gp_bmm(const double* __restrict__ A, const double* __restrict__ B, const double* D,
                                                double* C, int N, double alpha, double beta) {
    extern __shared__ __attribute__((aligned(16))) char dsm_bmm[];
    const int Kp = (N + 3) & ~3;                                  // k padded to the MFMA step with zeros
    const int lda = Kp + 1, ldb = GP_BMM_T + 1;
    double* As = reinterpret_cast<double*>(dsm_bmm);              // [64][Kp + 1]
    double* Bs = As + GP_BMM_R * lda;                             // [Kp][65]
    const int l = blockIdx.z, m0 = blockIdx.y * GP_BMM_R, n0 = blockIdx.x * GP_BMM_T;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Al = A + (size_t)l * N * N;
    const double* Bl = B + (size_t)l * N * N;
    // panels -> LDS, zero outside the matrix.  N even (rows 16-byte aligned): 16-byte loads, a whole panel (<= 16 per lane)
    // requested before the first LDS store; otherwise 8-byte loads in batches of 8
    if ((N & 1) == 0) {
        typedef __attribute__((ext_vector_type(2))) double f64x2_t;
        const int K2 = Kp >> 1, T2 = GP_BMM_T >> 1;
        constexpr int NB = (GP_BMM_T * (GP_MMAX / 2) + 255) / 256;              // 16
        f64x2_t t[NB], tb[NB];                                                  // both panels in flight: one memory latency
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + 256 * u, r = e / K2, k = 2 * (e - r * K2);
            t[u] = (e < GP_BMM_R * K2 && m0 + r < N && k < N) ? *reinterpret_cast<const f64x2_t*>(Al + (size_t)(m0 + r) * N + k)
                                                              : f64x2_t{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + 256 * u, k = e / T2, c = 2 * (e - k * T2);
            tb[u] = (e < Kp * T2 && k < N && n0 + c < N) ? *reinterpret_cast<const f64x2_t*>(Bl + (size_t)k * N + n0 + c)
                                                         : f64x2_t{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + 256 * u, r = e / K2, k = 2 * (e - r * K2);
            if (e < GP_BMM_R * K2) { As[r * lda + k] = t[u][0]; As[r * lda + k + 1] = t[u][1]; }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int e = tid + 256 * u, k = e / T2, c = 2 * (e - k * T2);
            if (e < Kp * T2) { Bs[k * ldb + c] = tb[u][0]; Bs[k * ldb + c + 1] = tb[u][1]; }
        }
    } else {
    for (int e0 = tid; e0 < GP_BMM_R * Kp; e0 += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, r = e / Kp, k = e - r * Kp;
            t[u] = (e < GP_BMM_R * Kp && m0 + r < N && k < N) ? Al[(size_t)(m0 + r) * N + k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, r = e / Kp, k = e - r * Kp;
            if (e < GP_BMM_R * Kp) As[r * lda + k] = t[u];
        }
    }
    for (int e0 = tid; e0 < Kp * GP_BMM_T; e0 += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, k = e / GP_BMM_T, c = e - k * GP_BMM_T;
            t[u] = (e < Kp * GP_BMM_T && k < N && n0 + c < N) ? Bl[(size_t)k * N + n0 + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, k = e / GP_BMM_T, c = e - k * GP_BMM_T;
            if (e < Kp * GP_BMM_T) Bs[k * ldb + c] = t[u];
        }
    }
    }
    __syncthreads();
    // 4 waves = 2 (16-row halves of the tile) x 2 (32-column halves): two 16 x 16 output fragments per wave
    const int wr = wave & 1, wc = wave >> 1;
    constexpr int NJ = GP_BMM_T / 32;
    f64x4_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const double* ap = As + (wr * 16 + (lane & 15)) * lda + (lane >> 4);
    const double* bp = Bs + (lane >> 4) * ldb + wc * (GP_BMM_T / 2) + (lane & 15);
#pragma unroll 4
    for (int k = 0; k < Kp; k += 4) {
        const double a = ap[k];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[k * ldb + 16 * j], acc[j], 0, 0, 0);
    }
    const double* Dl = D != nullptr ? D + (size_t)l * N * N : nullptr;
    double* Cl = C + (size_t)l * N * N;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wr * 16 + (lane >> 4) + 4 * r, col = n0 + wc * (GP_BMM_T / 2) + 16 * j + (lane & 15);
            if (row < N && col < N) {
                double v = alpha * acc[j][r];
                if (Dl != nullptr) v += beta * Dl[(size_t)row * N + col];
                Cl[(size_t)row * N + col] = v;
            }
        }
}


// ------------------------------------------------------------------------------------------------------------
// general batched fp64 product on the matrix cores (the rectangular products of the bound that round 1 left to the library):
//     C[l] (M x N) = alpha op(A[l]) B[l] + beta D[l],      op(A) = A ([M][K] rows, lda) or A^T (A stored [K][M], lda)
//     W = Kxz^T V   (TA, M = N = inducing points, K = batch rows)        elbo_functions.py:256-261 summed over subjects
//     Y = V (iK - iK H iK)   (M = batch rows, K = N = inducing points)
// One workgroup = one TM x 32 tile; K walks through LDS in chunks of 32 with the next chunk's global loads in flight in
// registers while the MFMAs (v_mfma_f64_16x16x4_f64) run on the current one.  split-K (gridDim.x > tiles): the slices add
// their partial tiles with fp64 atomics into a C the launcher has cleared (beta / D then belong to slice 0).
// ------------------------------------------------------------------------------------------------------------
#define GP_GK 32
template <int TM, int TA>
__global__ __launch_bounds__(256) void k_gp_gemm(const double* __restrict__ A, int lda, long sA, const double* __restrict__ B, int ldb,
                                                 long sB, const double* D, int ldd, long sD, double* C, int ldc, long sC, int M,
                                                 int N, int K, int tiles_n, int ksplit, int atomic, double alpha, double beta) {
    constexpr int LA = GP_GK + 1, LB = 32 + 1;
    __shared__ double As[TM * LA], Bs[GP_GK * LB];
    const int l = blockIdx.z, ks = blockIdx.y;
    const int m0 = (blockIdx.x / tiles_n) * TM, n0 = (blockIdx.x % tiles_n) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* Al = A + (size_t)l * sA;
    const double* Bl = B + (size_t)l * sB;
    const int kper = ((K + ksplit - 1) / ksplit + GP_GK - 1) / GP_GK * GP_GK;
    const int kb = ks * kper, ke = min(K, kb + kper);
    constexpr int NA = TM * GP_GK / 256, NBv = GP_GK * 32 / 256;       // elements per thread and chunk
    double ra[NA], rb[NBv];
    auto gload = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = tid + 256 * u;
            int r, k;
            if (TA) { k = e / TM; r = e - k * TM; } else { r = e / GP_GK; k = e - r * GP_GK; }     // contiguous axis fastest
            const bool in = m0 + r < M && k0 + k < ke;
            ra[u] = in ? (TA ? Al[(size_t)(k0 + k) * lda + m0 + r] : Al[(size_t)(m0 + r) * lda + k0 + k]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < NBv; ++u) {
            const int e = tid + 256 * u, k = e >> 5, c = e & 31;
            rb[u] = (k0 + k < ke && n0 + c < N) ? Bl[(size_t)(k0 + k) * ldb + n0 + c] : 0.0;
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int e = tid + 256 * u;
            int r, k;
            if (TA) { k = e / TM; r = e - k * TM; } else { r = e / GP_GK; k = e - r * GP_GK; }
            As[r * LA + k] = ra[u];
        }
#pragma unroll
        for (int u = 0; u < NBv; ++u) {
            const int e = tid + 256 * u;
            Bs[(e >> 5) * LB + (e & 31)] = rb[u];
        }
    };
    // waves: TM = 32 -> 2 x 2 fragments of 16 x 16; TM = 64 -> 4 row blocks x 2 column fragments each
    constexpr int NJ = TM == 32 ? 1 : 2;
    const int wr = TM == 32 ? (wave & 1) : wave, wc = TM == 32 ? (wave >> 1) : 0;
    f64x4_t acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    if (kb < ke) {
        gload(kb);
        for (int k0 = kb; k0 < ke; k0 += GP_GK) {
            __syncthreads();                               // the previous chunk's MFMAs have read their operands
            lstore();
            __syncthreads();
            if (k0 + GP_GK < ke) gload(k0 + GP_GK);
            const double* ap = As + (wr * 16 + (lane & 15)) * LA + (lane >> 4);
            const double* bp = Bs + (lane >> 4) * LB + wc * 16 + (lane & 15);
#pragma unroll
            for (int k = 0; k < GP_GK; k += 4) {
                const double a = ap[k];
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bp[k * LB + 16 * j], acc[j], 0, 0, 0);
            }
        }
    }
    const double* Dl = D != nullptr ? D + (size_t)l * sD : nullptr;
    double* Cl = C + (size_t)l * sC;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wr * 16 + (lane >> 4) + 4 * r, col = n0 + wc * 16 + 16 * j + (lane & 15);
            if (row < M && col < N) {
                double v = alpha * acc[j][r];
                if (atomic) {       // split-K slices, or the caller's C accumulates (hlvae_gp_gemm_acc)
                    if (ks == 0 && Dl != nullptr) v += beta * Dl[(size_t)row * ldd + col];
                    atomicAdd(Cl + (size_t)row * ldc + col, v);
                } else {
                    if (Dl != nullptr) v += beta * Dl[(size_t)row * ldd + col];
                    Cl[(size_t)row * ldc + col] = v;
                }
            }
        }
}

// (Round 3 also built k_gp_gemm's job in the fragment-from-L2 form of k_gp_chain_rb -- 32 rows x 128 columns per workgroup, no LDS, no
//  barrier: 40 us per product alone, the same as this kernel; both sit at a third of the fp64 MFMA peak.  Not kept.)
// out[l] = alpha A[l] x[l] + beta y[l]      (A: [batch][N][N] row-major, x, y, out: [batch][N]; y may be null or alias out)
// eight lanes per row, one workgroup per matrix: the matrix-vector products of the natural gradient (iK m, Bm m, iK P1, H tmp)
__global__ __launch_bounds__(1024) void k_gp_bmv(const double* __restrict__ A, const double* __restrict__ x, const double* y,
                                                 double* out, int N, double alpha, double beta) {
    const int l = blockIdx.x, sub = threadIdx.x & 7;
    const double* Al = A + (size_t)l * N * N;
    const double* xl = x + (size_t)l * N;
    for (int i = threadIdx.x >> 3; i < N; i += 128) {
        double s = 0.0;
        for (int j = sub; j < N; j += 8) s += Al[(size_t)i * N + j] * xl[j];
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (sub == 0) out[(size_t)l * N + i] = alpha * s + (y != nullptr ? beta * y[(size_t)l * N + i] : 0.0);
    }
}

// resid[l][b] = sum_m Kxz[l][b][m] w[l][m] - mu[b][l]      (A_part of elbo_functions.py:230; mu = the VAE's fp32 encoder means)
__global__ __launch_bounds__(256) void k_gp_resid(const double* __restrict__ Kxz, const double* __restrict__ w,
                                                  const float* __restrict__ mu, int L, int Bn, int M, double* __restrict__ out) {
    const int sub = threadIdx.x & 7;
    const long row = (long)blockIdx.x * 32 + (threadIdx.x >> 3);       // (l, b) flattened
    if (row >= (long)L * Bn) return;
    const int l = (int)(row / Bn), b = (int)(row - (long)l * Bn);
    const double* kr = Kxz + (size_t)row * M;
    const double* wl = w + (size_t)l * M;
    double s = 0.0;
    for (int m = sub; m < M; m += 8) s += kr[m] * wl[m];
    s += __shfl_xor(s, 4, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 1, 64);
    if (sub == 0) out[row] = s - (double)mu[(size_t)b * L + l];
}

// Natural-gradient terms and the right-hand side of the (m, H) update in one launch per latent (elbo_functions.py:279-283,
// training.py:130-137):
//     grad_m = -(iK P1) + Bm m,   grad_H = (Bm - iH) / 2,
//     tmp    = iH m - lr (grad_m - 2 grad_H m)      (H_new tmp = the updated m, once iH has been updated and inverted)
// Bm = iK W iK + iK.  One workgroup per latent, eight lanes per matrix row.
__global__ __launch_bounds__(1024) void k_gp_natgrad(const double* __restrict__ Bm, const double* __restrict__ iK,
                                                     const double* __restrict__ iH, const double* __restrict__ m,
                                                     const double* __restrict__ P1, double lr, int N, double* __restrict__ grad_m,
                                                     double* __restrict__ tmp) {
    __shared__ double ms[GP_MMAX], ps[GP_MMAX];
    const int l = blockIdx.x, sub = threadIdx.x & 7;
    const size_t o = (size_t)l * N * N;
    if (threadIdx.x < N) { ms[threadIdx.x] = m[(size_t)l * N + threadIdx.x]; ps[threadIdx.x] = P1[(size_t)l * N + threadIdx.x]; }
    __syncthreads();
    for (int i = threadIdx.x >> 3; i < N; i += 128) {
        double bm = 0.0, kp = 0.0, hm = 0.0;
        for (int j = sub; j < N; j += 8) {
            bm += Bm[o + (size_t)i * N + j] * ms[j];
            kp += iK[o + (size_t)i * N + j] * ps[j];
            hm += iH[o + (size_t)i * N + j] * ms[j];
        }
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) {
            bm += __shfl_xor(bm, off, 64);
            kp += __shfl_xor(kp, off, 64);
            hm += __shfl_xor(hm, off, 64);
        }
        if (sub == 0) {
            const double gm = bm - kp;                           // grad_m
            const double ghm = 0.5 * (bm - hm);                  // grad_H m
            grad_m[(size_t)l * N + i] = gm;
            tmp[(size_t)l * N + i] = hm - lr * (gm - 2.0 * ghm);
        }
    }
}

// grad_H = (Bm - iH) / 2, element-wise (elbo_functions.py:283)
__global__ __launch_bounds__(256) void k_gp_natgrad_h(const double* __restrict__ Bm, const double* __restrict__ iH, int n,
                                                      double* __restrict__ grad_H) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < n) grad_H[e] = 0.5 * (Bm[e] - iH[e]);
}
// iH <- iH + lr (grad_H + grad_H^T)   (training.py:131-133), in place: the input of the end-of-step inversion
__global__ __launch_bounds__(256) void k_gp_ih_update(const double* __restrict__ grad_H, double* __restrict__ iH, double lr, int N,
                                                      int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int NN = N * N, l = e / NN, ij = e - l * NN, i = ij / N, j = ij - i * N;
    iH[e] += lr * (grad_H[e] + grad_H[(size_t)l * NN + (size_t)j * N + i]);
}

// R + R^T of the inducing-point covariance gradient in one pass (elbo_functions.py of this package, kl_and_grads):
//     out = c (u m^T + m u^T - W + X + X^T) + H + m m^T           per latent, N x N, u and m are N-vectors
__global__ __launch_bounds__(256) void k_gp_rsym(const double* __restrict__ u, const double* __restrict__ m, const double* __restrict__ W,
                                                 const double* __restrict__ X, const double* __restrict__ H, double c, int N, int n_total,
                                                 double* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_total) return;
    const int l = e / (N * N), ij = e - l * N * N, i = ij / N, j = ij - i * N;
    const double ui = u[l * N + i], uj = u[l * N + j], mi = m[l * N + i], mj = m[l * N + j];
    const size_t o = (size_t)l * N * N;
    out[e] = c * (ui * mj + mi * uj - W[e] + X[e] + X[o + (size_t)j * N + i]) + H[e] + mi * mj;
}

// ------------------------------------------------------------------------------------------------------------
// The M x M algebra of a step behind W = sum_s Ks^T iB Ks as ONE launch (round 3).  Per latent it is two independent chains of
// dependent products (elbo_functions.py:279-283 and the K0zz gradient of kl_and_grads):
//     X:  T1 = iK W  ->  Bm = T1 iK + iK  ->  grad_m, grad_H, tmp                      (k_gp_natgrad, k_gp_natgrad_h)
//     Z:  HiKW = (H iK) W  ->  Rs = R + R^T  ->  T1b = iK Rs  ->  G = a (T1b iK) + b iK    (k_gp_rsym)
// As 5 k_gp_bmm + 3 element-wise launches on one stream they were the tail of the GP step's longest chain: every launch is 512
// workgroups x 61 KB of LDS that cannot share a CU with the VAE's optimiser launches running beside them (4 x 40 KB), and took
// 60-85 us instead of 12.  Here one workgroup of 512 threads per (latent, chain) walks its chain: 64 workgroups, no LDS for the
// products -- a wave owns a 32 x 64 block of the output and reads its fragments of both operands straight from L2 (the whole
// working set of a latent is < 1 MB) -- so nothing it needs can be taken away by a co-running launch.
// Fragments of v_mfma_f64_16x16x4_f64 over a block of 16 k: step s of the block uses k = kb + 4 (lane >> 4) + s on BOTH sides,
// so a lane's four A values are 32 contiguous bytes of its row (two 16-byte loads, every 128-byte row segment used whole) and
// its B values four loads of 128-byte row segments.  N % 4 == 0.
template <int FJ>   // 16-column fragments per wave: 4 (8 waves, 32 x 64 blocks) or 2 (16 waves, 32 x 32 blocks)
__device__ __forceinline__ void gp_mm_wg(const double* __restrict__ A, const double* __restrict__ B, const double* __restrict__ D,
                                         double* __restrict__ C, int N, double alpha, double beta, int wave, int lane) {
    // wave (wr, wc) owns rows 32 wr .. + 31, columns 16 FJ wc .. of the 128 x 128 padded output: 2 x FJ fragments
    constexpr int NWC = 8 / FJ;
    const int wr = wave / NWC, wc = wave % NWC, g = lane >> 4, q = lane & 15;
    f64x4_t acc[2][FJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j) acc[i][j] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int row0 = 32 * wr + q, col0 = 16 * FJ * wc + q;
    bool rok[2], cok[FJ];
    // 32-bit element offsets from the (uniform) operand pointers: A[offa[i] + kb], B[offb[j] + (kb + s) N] -- no 64-bit address
    // arithmetic per load (the first form's v_mad_u64_u32 chains made hipcc drain vmcnt(0) at the top of every block)
    unsigned offa[2], offb[FJ];
#pragma unroll
    for (int i = 0; i < 2; ++i) { rok[i] = row0 + 16 * i < N; offa[i] = (unsigned)(min(row0 + 16 * i, N - 1) * N + 4 * g); }
#pragma unroll
    for (int j = 0; j < FJ; ++j) { cok[j] = col0 + 16 * j < N; offb[j] = (unsigned)(4 * g * N + min(col0 + 16 * j, N - 1)); }
    // operands of block kb + 16 are requested before the 32 MFMAs of block kb (two register sets): a wave's loads ride under
    // its own products as well as under the other wave of its SIMD
    auto load = [&](int kb, f64x4_t (&a)[2], double (&b)[FJ][4]) {
        const bool kok = kb + 4 * g < N;                           // (N % 4 == 0: the lane's four k are in or out together)
        const double* Ak = A + kb;                                 // (uniform)
        const double* Bk = B + (size_t)kb * N;
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const f64x4_t*>(Ak + (kok ? offa[i] : 0u));
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) b[j][s_] = Bk[(kok ? offb[j] : 0u) + (unsigned)(s_ * N)];
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (!(kok && rok[i])) a[i] = f64x4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < FJ; ++j)
            if (!(kok && cok[j])) { b[j][0] = b[j][1] = b[j][2] = b[j][3] = 0.0; }
    };
    auto mma = [&](const f64x4_t (&a)[2], const double (&b)[FJ][4]) {
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < FJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][s_], b[j][s_], acc[i][j], 0, 0, 0);
    };
    f64x4_t a0[2], a1[2];
    double b0[FJ][4], b1[FJ][4];
    load(0, a0, b0);
    for (int kb = 0; kb < N; kb += 32) {
        const bool more = kb + 16 < N;
        if (more) load(kb + 16, a1, b1);
        mma(a0, b0);
        if (more) {
            if (kb + 32 < N) load(kb + 32, a0, b0);
            mma(a1, b1);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < FJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 32 * wr + 16 * i + g + 4 * r, col = 16 * FJ * wc + 16 * j + q;
                if (row < N && col < N) {
                    double v = alpha * acc[i][j][r];
                    if (D != nullptr) v += beta * D[(size_t)row * N + col];
                    C[(size_t)row * N + col] = v;
                }
            }
}

struct GpChainArgs {
    const double *iK, *W, *HiK, *H, *iH, *m, *P1, *u;
    double *T1, *Bm, *grad_m, *grad_H, *tmp, *HiKW, *Rs, *T1b, *G;
    double lr, c, g_alpha, g_beta;
    int N;
};
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_gp_chain(GpChainArgs a) {
    constexpr int FJ = THREADS == 512 ? 4 : 2;
    __shared__ double ms[GP_MMAX], ps[GP_MMAX];
    const int l = blockIdx.x, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t o = (size_t)l * N * N, ov = (size_t)l * N;
    if (blockIdx.y == 0) {                                        // ---- chain X
        gp_mm_wg<FJ>(a.iK + o, a.W + o, nullptr, a.T1 + o, N, 1.0, 0.0, wave, lane);
        if (tid < N) { ms[tid] = a.m[ov + tid]; ps[tid] = a.P1[ov + tid]; }
        __syncthreads();                                          // T1 complete (stores acknowledged), visible to the workgroup
        gp_mm_wg<FJ>(a.T1 + o, a.iK + o, a.iK + o, a.Bm + o, N, 1.0, 1.0, wave, lane);
        __syncthreads();
        const int sub = tid & 7;                                  // k_gp_natgrad: eight lanes per matrix row
        for (int i = tid >> 3; i < N; i += THREADS / 8) {
            double bm = 0.0, kp = 0.0, hm = 0.0;
            for (int j = sub; j < N; j += 8) {
                bm += a.Bm[o + (size_t)i * N + j] * ms[j];
                kp += a.iK[o + (size_t)i * N + j] * ps[j];
                hm += a.iH[o + (size_t)i * N + j] * ms[j];
            }
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {
                bm += __shfl_xor(bm, off, 64);
                kp += __shfl_xor(kp, off, 64);
                hm += __shfl_xor(hm, off, 64);
            }
            if (sub == 0) {
                const double gm = bm - kp, ghm = 0.5 * (bm - hm);
                a.grad_m[ov + i] = gm;
                a.tmp[ov + i] = hm - a.lr * (gm - 2.0 * ghm);
            }
        }
        for (int e = tid; e < N * N; e += THREADS) a.grad_H[o + e] = 0.5 * (a.Bm[o + e] - a.iH[o + e]);       // k_gp_natgrad_h
    } else {                                                      // ---- chain Z
        gp_mm_wg<FJ>(a.HiK + o, a.W + o, nullptr, a.HiKW + o, N, 1.0, 0.0, wave, lane);
        __syncthreads();
        for (int e = tid; e < N * N; e += THREADS) {                 // k_gp_rsym
            const int i = e / N, j = e - i * N;
            const double ui = a.u[ov + i], uj = a.u[ov + j], mi = a.m[ov + i], mj = a.m[ov + j];
            a.Rs[o + e] = a.c * (ui * mj + mi * uj - a.W[o + e] + a.HiKW[o + e] + a.HiKW[o + (size_t)j * N + i]) + a.H[o + e] + mi * mj;
        }
        __syncthreads();
        gp_mm_wg<FJ>(a.iK + o, a.Rs + o, nullptr, a.T1b + o, N, 1.0, 0.0, wave, lane);
        __syncthreads();
        gp_mm_wg<FJ>(a.T1b + o, a.iK + o, a.iK + o, a.G + o, N, a.g_alpha, a.g_beta, wave, lane);
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same algebra by ROW BLOCKS (round 3, second form).  A 32-row block of T1 = iK W, Bm = T1 iK + iK, X = (H iK) W and
// X^T = W (H iK)^T needs no other block's results, so launch 1 gives one workgroup per (latent, row block) the natural-gradient
// outputs and its rows of Rs = c (u m^T + m u^T - W + X + X^T) + H + m m^T; launch 2 does the same for T1b = iK Rs and
// G = g_alpha T1b iK + g_beta iK once Rs is whole.  2 x 128 workgroups that never wait for each other, six 32 x N x N product
// passes in all, A and intermediate row blocks in LDS -- 52 us alone where k_gp_chain takes 97 (64 workgroups, up to three dependent
// N x N x N products each) and the eight separate launches 79.
// (An expanded form -- iK Rs iK = c [..] - c T1 iK + c (Q W iK + T1 Q) + Q + .. with Q = iK H iK, five passes in ONE launch -- was
//  built first and is algebraically equal, but it cancels AFTER the multiplication by iK instead of inside Rs: on the
//  config-5 matrices (condition 1e8) the inducing-point gradient lost its sign in 15 % of the large entries.)
// grid (batch, ceil(N / 32)), 512 threads: wave w owns output columns 16 w .. 16 w + 15 of both 16-row halves.
#define GP_RB_LD 130                                          // LDS row stride of the intermediate row block (>= GP_MMAX, 16-byte aligned rows)
static_assert(GP_RB_LD >= GP_MMAX && GP_RB_LD % 2 == 0, "row blocks of up to GP_MMAX columns, 16-byte aligned rows");
struct GpChainRbArgs {
    const double *iK, *W, *HiK, *H, *iH, *m, *P1, *u;
    double *grad_m, *grad_H, *tmp, *Rs, *G;
    double lr, c, g_alpha, g_beta;
    int N;
};
// both A operands from LDS row blocks: acc0 += As0 B (B k-major in global memory), acc1 += As1 (Bt given ? Bt^T : B)
// (the A fragments of a row block are the same for all eight waves: read from global memory by each of them they were eight
//  times the L2 traffic of the B fragments and the latency every k-block waited for)
template <bool USE0, bool USE1, bool TRANS_B1>
__device__ __forceinline__ void gp_rb_sweep_lds2(const double* As0, const double* As1, const double* __restrict__ B,
                                                 const double* __restrict__ Bt, int N, int col, bool cok, int g, int q,
                                                 f64x4_t (&acc0)[2], f64x4_t (&acc1)[2]) {
    typedef __attribute__((ext_vector_type(2))) double f64x2_t;
    const unsigned offb = (unsigned)(4 * g * N + min(col, N - 1)), offbt = (unsigned)(min(col, N - 1) * N + 4 * g);
    auto loadb = [&](int kb, double (&b)[4], f64x4_t& bt) {
        const bool kok = kb + 4 * g < N;
        if (USE0 || !TRANS_B1)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) b[s_] = (B + (size_t)kb * N)[(kok ? offb : 0u) + (unsigned)(s_ * N)];
        if (TRANS_B1) bt = *reinterpret_cast<const f64x4_t*>(Bt + kb + (kok ? offbt : 0u));
        if (!(kok && cok)) { if (USE0 || !TRANS_B1) { b[0] = b[1] = b[2] = b[3] = 0.0; } if (TRANS_B1) bt = f64x4_t{0.0, 0.0, 0.0, 0.0}; }
    };
    auto mma = [&](int kb, const double (&b)[4], const f64x4_t& bt) {
        const bool kok = kb + 4 * g < N;
        const int kc = kok ? kb + 4 * g : 0;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {
            double t0[4], t1[4];
            if (USE0) {
                const double* tp = As0 + (16 * fi + q) * GP_RB_LD + kc;
                const f64x2_t x01 = *reinterpret_cast<const f64x2_t*>(tp), x23 = *reinterpret_cast<const f64x2_t*>(tp + 2);
                t0[0] = x01[0]; t0[1] = x01[1]; t0[2] = x23[0]; t0[3] = x23[1];
                if (!kok) { t0[0] = t0[1] = t0[2] = t0[3] = 0.0; }
            }
            if (USE1) {
                const double* tp = As1 + (16 * fi + q) * GP_RB_LD + kc;
                const f64x2_t x01 = *reinterpret_cast<const f64x2_t*>(tp), x23 = *reinterpret_cast<const f64x2_t*>(tp + 2);
                t1[0] = x01[0]; t1[1] = x01[1]; t1[2] = x23[0]; t1[3] = x23[1];
                if (!kok) { t1[0] = t1[1] = t1[2] = t1[3] = 0.0; }
            }
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                if (USE0) acc0[fi] = __builtin_amdgcn_mfma_f64_16x16x4f64(t0[s_], b[s_], acc0[fi], 0, 0, 0);
                if (USE1) acc1[fi] = __builtin_amdgcn_mfma_f64_16x16x4f64(t1[s_], TRANS_B1 ? bt[s_] : b[s_], acc1[fi], 0, 0, 0);
            }
        }
    };
    double b0[4], b1[4];
    f64x4_t bt0, bt1;
    loadb(0, b0, bt0);
    for (int kb = 0; kb < N; kb += 32) {
        const bool more = kb + 16 < N;
        if (more) loadb(kb + 16, b1, bt1);
        mma(kb, b0, bt0);
        if (more) {
            if (kb + 32 < N) loadb(kb + 32, b0, bt0);
            mma(kb + 16, b1, bt1);
        }
    }
}
// rows [R0, R0 + 32) of a row-major N x N matrix -> LDS [32][GP_RB_LD] (16-byte pieces, zero rows past the matrix)
__device__ __forceinline__ void gp_rb_stage(const double* __restrict__ A, int N, int R0, double* As, int tid) {
    typedef __attribute__((ext_vector_type(2))) double f64x2_t;
    const int n2 = N >> 1;                                    // (N even)
    for (int e = tid; e < 32 * n2; e += 512) {
        const int r = e / n2, c2 = e - r * n2;
        const f64x2_t v = R0 + r < N ? *reinterpret_cast<const f64x2_t*>(A + (size_t)(R0 + r) * N + 2 * c2) : f64x2_t{0.0, 0.0};
        *reinterpret_cast<f64x2_t*>(As + r * GP_RB_LD + 2 * c2) = v;
    }
}
template <int PHASE>
__global__ __launch_bounds__(512) void k_gp_chain_rb(GpChainRbArgs a) {
    extern __shared__ __attribute__((aligned(16))) double rb_sm[];
    double* t1s = rb_sm;                                      // T1 (phase 1) / T1b (phase 2) rows   [32][GP_RB_LD]
    double* ar0 = t1s + 32 * GP_RB_LD;                        // staged A row blocks: iK[R] | (phase 1) (H iK)[R] | W[R]
    double* ar1 = ar0 + 32 * GP_RB_LD;
    double* ar2 = ar1 + 32 * GP_RB_LD;
    double* ms = ar2 + 32 * GP_RB_LD;                         // m, P1, u   [GP_MMAX] each
    double* ps = ms + GP_MMAX;
    double* us = ps + GP_MMAX;
    double* rowsum = us + GP_MMAX;                            // sum_j Bm[i][j] m[j] of the block's rows  [32]
    const int l = blockIdx.x, R0 = 32 * blockIdx.y, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, q = lane & 15, col = 16 * wave + q;
    const bool cok = col < N;
    const size_t o = (size_t)l * N * N, ov = (size_t)l * N;
    const double* iK = a.iK + o;
    f64x4_t z2[2] = {f64x4_t{0.0, 0.0, 0.0, 0.0}, f64x4_t{0.0, 0.0, 0.0, 0.0}};
    if (PHASE == 1) {
        const double* W = a.W + o;
        const double* HiK = a.HiK + o;
        if (tid < N) { ms[tid] = a.m[ov + tid]; ps[tid] = a.P1[ov + tid]; us[tid] = a.u[ov + tid]; }
        if (tid < 32) rowsum[tid] = 0.0;
        gp_rb_stage(iK, N, R0, ar0, tid);
        gp_rb_stage(HiK, N, R0, ar1, tid);
        gp_rb_stage(W, N, R0, ar2, tid);
        __syncthreads();
        // sweep over W: T1 rows = iK[R] W, X rows = HiK[R] W
        f64x4_t at[2] = {z2[0], z2[1]}, ax[2] = {z2[0], z2[1]};
        gp_rb_sweep_lds2<true, true, false>(ar0, ar1, W, nullptr, N, col, cok, g, q, at, ax);
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (cok) t1s[(16 * fi + g + 4 * r) * GP_RB_LD + col] = at[fi][r];
        __syncthreads();
        // P = T1 iK (A from LDS);  X^T rows = W[R] (H iK)^T (B transposed)
        f64x4_t ap[2] = {z2[0], z2[1]}, axt[2] = {z2[0], z2[1]};
        gp_rb_sweep_lds2<true, true, true>(t1s, ar2, iK, HiK, N, col, cok, g, q, ap, axt);
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr_ = 16 * fi + g + 4 * r, i = R0 + lr_;
                double part = 0.0;
                if (i < N && cok) {
                    const size_t e = (size_t)i * N + col;
                    const double Bm = ap[fi][r] + iK[e];
                    a.grad_H[o + e] = 0.5 * (Bm - a.iH[o + e]);
                    a.Rs[o + e] = a.c * (us[i] * ms[col] + ms[i] * us[col] - W[e] + ax[fi][r] + axt[fi][r]) + a.H[o + e] + ms[i] * ms[col];
                    part = Bm * ms[col];
                }
                part += __shfl_xor(part, 8, 64);
                part += __shfl_xor(part, 4, 64);
                part += __shfl_xor(part, 2, 64);
                part += __shfl_xor(part, 1, 64);
                if (q == 0 && i < N) atomicAdd(&rowsum[lr_], part);
            }
        __syncthreads();
        if (tid < 256) {                                      // natural-gradient vectors of the block's rows (k_gp_natgrad)
            const int lr_ = tid >> 3, i = R0 + lr_, sub = tid & 7;
            if (i < N) {
                double kp = 0.0, hm = 0.0;
                for (int j = sub; j < N; j += 8) {
                    kp += iK[(size_t)i * N + j] * ps[j];
                    hm += a.iH[o + (size_t)i * N + j] * ms[j];
                }
                kp += __shfl_xor(kp, 4, 64); kp += __shfl_xor(kp, 2, 64); kp += __shfl_xor(kp, 1, 64);
                hm += __shfl_xor(hm, 4, 64); hm += __shfl_xor(hm, 2, 64); hm += __shfl_xor(hm, 1, 64);
                if (sub == 0) {
                    const double bm = rowsum[lr_], gm = bm - kp, ghm = 0.5 * (bm - hm);
                    a.grad_m[ov + i] = gm;
                    a.tmp[ov + i] = hm - a.lr * (gm - 2.0 * ghm);
                }
            }
        }
    } else {
        const double* Rs = a.Rs + o;
        gp_rb_stage(iK, N, R0, ar0, tid);
        __syncthreads();
        f64x4_t at[2] = {z2[0], z2[1]}, dum[2] = {z2[0], z2[1]};
        gp_rb_sweep_lds2<true, false, false>(ar0, nullptr, Rs, nullptr, N, col, cok, g, q, at, dum);      // T1b rows = iK[R] Rs
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (cok) t1s[(16 * fi + g + 4 * r) * GP_RB_LD + col] = at[fi][r];
        __syncthreads();
        f64x4_t ag[2] = {z2[0], z2[1]};
        gp_rb_sweep_lds2<true, false, false>(t1s, nullptr, iK, nullptr, N, col, cok, g, q, ag, dum);
#pragma unroll
        for (int fi = 0; fi < 2; ++fi)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = R0 + 16 * fi + g + 4 * r;
                if (i < N && cok) {
                    const size_t e = (size_t)i * N + col;
                    a.G[o + e] = a.g_alpha * ag[fi][r] + a.g_beta * iK[e];
                }
            }
    }
}

// out[l][m] = sum_b A[l][b][m] x[l][b]: the two matrix^T-vector products of the bound (Kxz^T v, V^T mu).  As batched GEMMs with
// one column the library reads the 15.7 MB operand at 0.5 TB/s (32 us each).  One workgroup per (latent, row chunk) streams its
// part of the slab: thread = (column m, one of 1024 / 128 row groups), partials folded through LDS.  With ONE workgroup per
// latent (round 1) 32 CUs pulled the whole operand: 65 us for 31 MB at 1024 rows, on the longest chain of the GP step.  Several
// chunks per latent add their rows with fp64 atomics into an output the launcher has cleared.  (A last-chunk-reduces ticket
// needs a device-scope fence per workgroup: the L2 write-backs made the kernel 226 us and slowed everything beside it.)
#define GP_GEMV_CHUNKS_MAX 16
template <typename XT>
__global__ __launch_bounds__(1024) void k_gp_gemv_t(const double* __restrict__ A, const XT* __restrict__ x, long xs_l, long xs_b,
                                                    double* __restrict__ out, int Bn, int M) {
    __shared__ double red[8][GP_MMAX];
    const int l = blockIdx.x, nc = gridDim.y, c = blockIdx.y, m = threadIdx.x & 127, g = threadIdx.x >> 7;
    const int per = ((Bn + nc - 1) / nc + 7) & ~7, b_lo = c * per, b_hi = min(Bn, b_lo + per);
    const double* Al = A + (size_t)l * Bn * M;
    const XT* xl = x + (size_t)l * xs_l;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if (m < M) {
        int b = b_lo + g;
        for (; b + 8 * 15 < b_hi; b += 8 * 16) {                     // 16 loads of the slab in flight per lane (32 KB per wave)
            double t[16], xv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                t[k] = Al[(size_t)(b + 8 * k) * M + m];
                xv[k] = (double)xl[(size_t)(b + 8 * k) * xs_b];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] += t[k] * xv[k];
        }
        for (; b < b_hi; b += 8) acc[0] += Al[(size_t)b * M + m] * (double)xl[(size_t)b * xs_b];
    }
    red[g][m] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    __syncthreads();
    if (g == 0 && m < M) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][m];
        if (nc == 1) out[(size_t)l * M + m] = s;
        else atomicAdd(out + (size_t)l * M + m, s);
    }
}

static int gemv_chunks(int L, int B) {      // about one workgroup per CU, at least 128 rows each
    int c = 256 / (L > 0 ? L : 1);
    if (c > GP_GEMV_CHUNKS_MAX) c = GP_GEMV_CHUNKS_MAX;
    while (c > 1 && B / c < 128) --c;
    return c < 1 ? 1 : c;
}

// G[l][b][m] = c (v[l][b] w[l][m] - Y[l][b][m]): the gradient w.r.t. K0xz from its two pieces in one pass (as baddbmm: a
// 15.7 MB copy plus a rank-1 batched GEMM)
__global__ __launch_bounds__(256) void k_gp_gkxz(const double* __restrict__ Y, const double* __restrict__ v, const double* __restrict__ w,
                                                 double c, int Bn, int M, long n_total, double* __restrict__ G) {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n_total; e += (long)gridDim.x * 256) {
        const long lb = e / M;
        const int m = (int)(e - lb * M);
        const long l = lb / Bn;
        G[e] = c * (v[lb] * w[l * M + m] - Y[e]);
    }
}

// every scalar of the bound in one launch (elbo_functions.py:268-285):
//   out += c/2 [ sum(part) - sum(W o iK) + sum(Qm o W) - sum(log_var) ]
//        + rep { 1/2 [ sum(iK o H) + sum(m o iKm) + sum(ldK) - sum(ldH) ] + konst }
// The first bracket is linear in per-subject sums: under data parallelism every rank passes its LOCAL part / W / lv and
// rep = 1 / world for the replicated terms; the sum of the ranks' results is the bound of the global batch.
__global__ __launch_bounds__(256) void k_gp_bound(const double* __restrict__ part, int n_part, const double* __restrict__ W,
                                                  const double* __restrict__ iK, const double* __restrict__ Qm,
                                                  const double* __restrict__ H, int LMM, const double* __restrict__ m,
                                                  const double* __restrict__ iKm, int LM, const double* __restrict__ ldK,
                                                  const double* __restrict__ ldH, int L, const float* __restrict__ lv, int BL,
                                                  double c, double rep, double konst, double* __restrict__ out) {
    __shared__ double red[2][4];
    const int g = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    double ac = 0.0, au = 0.0;
    for (int e = g; e < LMM; e += stride) {
        const double w = W[e], ik = iK[e];
        ac -= Qm[e] * w;                                   // Qm holds N1 = iK - iK H iK:  sum((Q - iK) o W)
        au += ik * H[e];
    }
    for (int e = g; e < n_part; e += stride) ac += part[e];
    for (int e = g; e < BL; e += stride) ac -= (double)lv[e];
    for (int e = g; e < LM; e += stride) au += m[e] * iKm[e];
    for (int e = g; e < L; e += stride) au += ldK[e] - ldH[e];
    ac = wave_sum_d(ac);
    au = wave_sum_d(au);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ac; red[1][threadIdx.x >> 6] = au; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sc = red[0][0] + red[0][1] + red[0][2] + red[0][3], su = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        atomicAdd(out, 0.5 * c * sc + rep * (0.5 * su + (blockIdx.x == 0 ? konst : 0.0)));
    }
}

// torch.optim.Adam (HLVAE_main.py:277-278) on the flat fp64 arena: the step counter lives on the device (HIP-graph safe)
// and the consumed gradients are zeroed for the next step's atomics.  step[0] = completed steps, step[1] = ticket: the
// last workgroup to finish (every workgroup has read step[0] by then) advances the counter and resets the ticket.
__global__ __launch_bounds__(256) void k_gp_adam(double* __restrict__ p, double* __restrict__ g, double* __restrict__ m1,
                                                 double* __restrict__ m2, int n, int64_t* __restrict__ step, double lr,
                                                 double b1, double b2, double eps) {
    const double t = (double)(step[0] + 1);
    const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
    const double step_size = lr / bc1, rs = 1.0 / sqrt(bc2);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const double gi = g[i];
        const double a = b1 * m1[i] + (1.0 - b1) * gi, v = b2 * m2[i] + (1.0 - b2) * gi * gi;
        m1[i] = a;
        m2[i] = v;
        p[i] -= step_size * a / (sqrt(v) * rs + eps);
        g[i] = 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(step + 1), 1ull);
        if (done == gridDim.x - 1) {
            step[1] = 0;
            step[0] += 1;
        }
    }
}

// The head of the prior's state update as ONE launch (round 3): Adam on [raw hyper-parameters | inducing points] with the
// transform of the hyper-parameters it has just produced, and iH <- iH + lr (grad_H + grad_H^T) in the remaining workgroups --
// k_gp_adam -> k_gp_ih_update -> k_gp_transform were three dependent launches of 5-8 us each at the start of the step's serial tail.
__global__ __launch_bounds__(256) void k_gp_state_head(double* __restrict__ p, double* __restrict__ g, double* __restrict__ m1,
                                                       double* __restrict__ m2, int n, int64_t* __restrict__ step, double lr,
                                                       double b1, double b2, double eps, int n_hyp, double* __restrict__ hyp,
                                                       const double* __restrict__ grad_H, double* __restrict__ iH, double ng_lr,
                                                       int N, int n_ih, int blocks_adam) {
    if ((int)blockIdx.x < blocks_adam) {
        const double t = (double)(step[0] + 1);
        const double bc1 = 1.0 - pow(b1, t), bc2 = 1.0 - pow(b2, t);
        const double step_size = lr / bc1, rs = 1.0 / sqrt(bc2);
        const int i = blockIdx.x * 256 + threadIdx.x;
        if (i < n) {
            const double gi = g[i];
            const double a = b1 * m1[i] + (1.0 - b1) * gi, v = b2 * m2[i] + (1.0 - b2) * gi * gi;
            m1[i] = a;
            m2[i] = v;
            const double pn = p[i] - step_size * a / (sqrt(v) * rs + eps);
            p[i] = pn;
            g[i] = 0.0;
            if (i < n_hyp) {                                      // k_gp_transform
                const double x = pn + 16.0;
                const double sp = x > 20.0 ? x : log1p(exp(x));
                const double pos = exp(sp - 16.0);
                hyp[i] = pos;
                hyp[(size_t)n_hyp + i] = 1.0 / (1.0 + exp(-x));
                hyp[(size_t)2 * n_hyp + i] = 1.0 / (pos * pos);
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(step + 1), 1ull);
            if (done == (unsigned long long)blocks_adam - 1) {
                step[1] = 0;
                step[0] += 1;
            }
        }
    } else {                                                      // k_gp_ih_update
        const int e = ((int)blockIdx.x - blocks_adam) * 256 + threadIdx.x;
        if (e >= n_ih) return;
        const int NN = N * N, l = e / NN, ij = e - l * NN, i = ij / N, j = ij - i * N;
        iH[e] += ng_lr * (grad_H[e] + grad_H[(size_t)l * NN + (size_t)j * N + i]);
    }
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
        int n_rbf = 0;
        for (int f = 0; f < k->n_factors[t]; ++f) {
            HL_REQUIRE(k->dim[t][f] >= 0 && k->dim[t][f] < Q, HLVAE_EINVAL, "gp kernel: covariate index");
            if (k->kind[t][f] == HLVAE_GP_RBF) {
                HL_REQUIRE(k->ls_slot[t][f] >= 0 && k->ls_slot[t][f] < n_slots, HLVAE_EINVAL, "gp kernel: lengthscale row");
                ++n_rbf;
            }
        }
        HL_REQUIRE(n_rbf <= GP_MAX_RBF, HLVAE_EINVAL, "gp kernel: at most 2 RBF factors per term");
    }
    return 0;
}

extern "C" {

int hlvae_gp_transform(const double* raw, int n_slots, int L, double* hyp, hlvae_stream s) {
    HL_REQUIRE(raw && hyp && n_slots > 0 && L > 0, HLVAE_EINVAL, "gp_transform: bad arguments");
    const int n = n_slots * L;
    HL_PROF("gp_transform", (hipStream_t)s);
    k_gp_transform<<<(n + 255) / 256, 256, 0, (hipStream_t)s>>>(raw, n, hyp);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_kernel_matrix(const hlvae_gp_kernel* k, const double* hyp, int n_slots, int L, int Q, const double* x1, int n1,
                           int per_latent1, const double* x2, int n2, int per_latent2, double jitter, double* out,
                           hlvae_stream s) {
    if (int rc = gp_check_kernel(k, n_slots, Q)) return rc;
    HL_REQUIRE(hyp && x1 && x2 && out && L > 0 && n1 > 0 && n2 > 0, HLVAE_EINVAL, "gp_kernel_matrix: bad arguments");
    HL_REQUIRE(n2 <= GP_MMAX && Q <= 8, HLVAE_ESHAPE, "gp_kernel_matrix: n2=%d (max %d columns), Q=%d (max 8)", n2, GP_MMAX, Q);
    HL_PROF("gp_kernel_matrix", (hipStream_t)s);
    const int rows_wg = (long)((n1 + GP_KM_ROWS - 1) / GP_KM_ROWS) * L >= 512 ? GP_KM_ROWS : 8;
    const dim3 grid((n1 + rows_wg - 1) / rows_wg, L);
    if (gp_kernel_small(k))
        k_gp_kernel_matrix<4, 1><<<grid, 256, 0, (hipStream_t)s>>>(*k, hyp, n_slots, L, Q, x1, n1, per_latent1, x2, n2, per_latent2, jitter, out,
                                                                  rows_wg);
    else
        k_gp_kernel_matrix<HLVAE_GP_MAX_TERMS, GP_MAX_RBF><<<grid, 256, 0, (hipStream_t)s>>>(*k, hyp, n_slots, L, Q, x1, n1, per_latent1, x2, n2,
                                                                                           per_latent2, jitter, out, rows_wg);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_chol_inv(const double* A, int n, int N, double* inv, double* logdet, int* fail, hlvae_stream s) {
    HL_REQUIRE(A && inv && logdet && n > 0 && N > 0 && N <= GP_MMAX, HLVAE_EINVAL, "gp_chol_inv: N=%d (max %d)", N, GP_MMAX);
    HL_PROF("gp_spd_inv", (hipStream_t)s);
    k_gp_spd_inv<<<n, 256, 0, (hipStream_t)s>>>(A, N, inv, logdet, fail, 0, nullptr);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_subject_fwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* hyp, int n_slots, int L, int Q,
                         const double* x, const double* noise, const int32_t* idx, int S, int T, const double* Kxz, int B,
                         int M, const double* resid, const float* lv, double c, double* iB, double* K0s, double* V,
                         double* v, double* part, float* g_mu, float* g_lv, const double* iKm, const float* mu, double* u_acc,
                         double* p1_acc, hlvae_stream s) {
    if (int rc = gp_check_kernel(k0, n_slots, Q)) return rc;
    if (int rc = gp_check_kernel(k1, n_slots, Q)) return rc;
    HL_REQUIRE(T >= 1 && T <= GP_TMAX && S >= 1 && Q <= 8 && M <= GP_MMAX, HLVAE_ESHAPE,
               "gp_subject_fwd: T=%d (max %d), M=%d (max %d)", T, GP_TMAX, M, GP_MMAX);
    HL_REQUIRE((iKm == nullptr || mu != nullptr) && (resid != nullptr || iKm != nullptr) && ((u_acc == nullptr) == (p1_acc == nullptr)) &&
                   (u_acc == nullptr || mu != nullptr), HLVAE_EINVAL,
               "gp_subject_fwd: resid or (iKm, mu); u_acc / p1_acc both or none, with mu");
    HL_PROF("gp_subject_fwd", (hipStream_t)s);
#define GP_SF(NTv, NRv)                                                                                                         \
    k_gp_subject_fwd<NTv, NRv><<<dim3(S, L), 256, (size_t)T * M * sizeof(double), (hipStream_t)s>>>(                               \
        *k0, *k1, hyp, n_slots, L, Q, x, noise, idx, T, Kxz, B, M, resid, lv, c, iB, K0s, V, v, part, g_mu, g_lv, iKm, mu, u_acc, p1_acc)
    if (gp_kernel_small(k0) && gp_kernel_small(k1)) GP_SF(4, 1); else GP_SF(HLVAE_GP_MAX_TERMS, GP_MAX_RBF);
#undef GP_SF
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_subject_bwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* hyp, int n_slots, int L, int Q,
                         const double* x, const int32_t* idx, int S, int T, int B, int M, const double* iB, const double* K0s,
                         const double* V, const double* v, const double* Y, const float* lv, double c, double* gprm,
                         hlvae_stream s) {
    if (int rc = gp_check_kernel(k0, n_slots, Q)) return rc;
    if (int rc = gp_check_kernel(k1, n_slots, Q)) return rc;
    HL_REQUIRE(T >= 1 && T <= GP_TMAX && S >= 1 && M <= GP_MMAX, HLVAE_ESHAPE, "gp_subject_bwd: T=%d M=%d", T, M);
    const size_t smem = ((size_t)2 * T * (M + 1) + (size_t)3 * T * (T + 1)) * sizeof(double);
    const bool small = gp_kernel_small(k0) && gp_kernel_small(k1);
    static size_t attr_max[2] = {32 * 1024, 32 * 1024};
    if (smem > attr_max[small]) {
        HL_CHECK(hipFuncSetAttribute(small ? reinterpret_cast<const void*>(&k_gp_subject_bwd<4, 1>)
                                           : reinterpret_cast<const void*>(&k_gp_subject_bwd<HLVAE_GP_MAX_TERMS, GP_MAX_RBF>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_max[small] = smem;
    }
    HL_PROF("gp_subject_bwd", (hipStream_t)s);
#define GP_SB(NTv, NRv)                                                                                                         \
    k_gp_subject_bwd<NTv, NRv><<<dim3(S, L), 256, smem, (hipStream_t)s>>>(*k0, *k1, hyp, n_slots, L, Q, x, idx, T, B, M, iB, K0s, V, v, Y, \
                                                                        lv, c, gprm)
    if (small) GP_SB(4, 1); else GP_SB(HLVAE_GP_MAX_TERMS, GP_MAX_RBF);
#undef GP_SB
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_param_grad(const hlvae_gp_kernel* k, const double* hyp, int n_slots, int L, int Q, const double* x1, int n1,
                        int per_latent1, const double* x2, int n2, int both_args, const double* G, double* gprm, double* gx2,
                        const double* v, const double* w, double c, hlvae_stream s) {
    if (int rc = gp_check_kernel(k, n_slots, Q)) return rc;
    HL_REQUIRE(!both_args || n1 == n2, HLVAE_ESHAPE, "gp_param_grad: both_args needs a square matrix");
    HL_REQUIRE((v == nullptr) == (w == nullptr), HLVAE_EINVAL, "gp_param_grad: v and w both or none");
    HL_PROF("gp_param_grad", (hipStream_t)s);
    const bool small = gp_kernel_small(k);
    // rows per workgroup: 64 (a lane walks 16) when that still fills the machine, else 16 (K0zz: 128 -> 512 workgroups)
    const bool tall = (long)((n2 + 63) / 64) * ((n1 + 63) / 64) * L >= 512;
#define GP_PG(NTv, NRv, ROWSv)                                                                                             \
    k_gp_param_grad<NTv, NRv, ROWSv><<<dim3((n2 + 63) / 64, (n1 + ROWSv - 1) / ROWSv, L), 256, 0, (hipStream_t)s>>>(        \
        *k, hyp, n_slots, L, Q, x1, n1, per_latent1, x2, n2, both_args, G, gprm, gx2, v, w, c)
    if (small) { if (tall) GP_PG(4, 1, 64); else GP_PG(4, 1, 16); }
    else { if (tall) GP_PG(HLVAE_GP_MAX_TERMS, GP_MAX_RBF, 32); else GP_PG(HLVAE_GP_MAX_TERMS, GP_MAX_RBF, 16); }   // (64 rows: 256 VGPRs)
#undef GP_PG
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_bmm(const double* A, const double* B, const double* D, double* C, int N, int batch, double alpha, double beta,
                 hlvae_stream s) {
    HL_REQUIRE(A && B && C && N >= 1 && N <= GP_MMAX && batch >= 1, HLVAE_EINVAL, "gp_bmm: N=%d batch=%d", N, batch);
    const int Kp = (N + 3) & ~3;
    const size_t smem = ((size_t)GP_BMM_R * (Kp + 1) + (size_t)Kp * (GP_BMM_T + 1)) * sizeof(double);
    static size_t attr_max = 0;
    if (smem > attr_max) {
        HL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gp_bmm), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_max = smem;
    }
    const int tn = (N + GP_BMM_T - 1) / GP_BMM_T, tm = (N + GP_BMM_R - 1) / GP_BMM_R;
    HL_PROF("gp_bmm", (hipStream_t)s);
    k_gp_bmm<<<dim3(tn, tm, batch), 256, smem, (hipStream_t)s>>>(A, B, D, C, N, alpha, beta);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_gemv_t(const double* A, const double* x, long x_stride_l, long x_stride_b, double* out, int L, int B, int M,
                    hlvae_stream s) {
    HL_REQUIRE(A && x && out && L >= 1 && B >= 1 && M >= 1 && M <= GP_MMAX, HLVAE_EINVAL, "gp_gemv_t: L=%d B=%d M=%d", L, B, M);
    const int nc = gemv_chunks(L, B);
    if (nc > 1) HL_CHECK(hipMemsetAsync(out, 0, sizeof(double) * (size_t)L * M, (hipStream_t)s));
    HL_PROF("gp_gemv_t", (hipStream_t)s);
    k_gp_gemv_t<double><<<dim3(L, nc), 1024, 0, (hipStream_t)s>>>(A, x, x_stride_l, x_stride_b, out, B, M);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_gkxz(const double* Y, const double* v, const double* w, double c, int L, int B, int M, double* G, hlvae_stream s) {
    HL_REQUIRE(Y && v && w && G && L >= 1 && B >= 1 && M >= 1, HLVAE_EINVAL, "gp_gkxz: null argument");
    const long n = (long)L * B * M;
    HL_PROF("gp_gkxz", (hipStream_t)s);
    k_gp_gkxz<<<(int)((n + 4 * 256 - 1) / (4 * 256)), 256, 0, (hipStream_t)s>>>(Y, v, w, c, B, M, n, G);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_rsym(const double* u, const double* m, const double* W, const double* X, const double* H, double c, int N, int batch,
                  double* out, hlvae_stream s) {
    HL_REQUIRE(u && m && W && X && H && out && N >= 1 && batch >= 1, HLVAE_EINVAL, "gp_rsym: null argument");
    const int n = batch * N * N;
    HL_PROF("gp_rsym", (hipStream_t)s);
    k_gp_rsym<<<(n + 255) / 256, 256, 0, (hipStream_t)s>>>(u, m, W, X, H, c, N, n, out);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_debug_gp_clk(long long* host, int install) {
    // install != 0: allocate + install the stamp buffer; else copy it to host [2][GP_CLK_WG][GP_CLK_PH] and uninstall
    static long long* buf = nullptr;
    const size_t n = (size_t)2 * GP_CLK_WG * GP_CLK_PH;
    if (install) {
        if (!buf) HL_CHECK(hipMalloc(&buf, n * sizeof(long long)));
        HL_CHECK(hipMemset(buf, 0, n * sizeof(long long)));
        HL_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gpclk), &buf, sizeof(buf)));
        return 0;
    }
    HL_REQUIRE(buf && host, HLVAE_EINVAL, "gp_clk: not installed");
    HL_CHECK(hipDeviceSynchronize());
    HL_CHECK(hipMemcpy(host, buf, n * sizeof(long long), hipMemcpyDeviceToHost));
    long long* nul = nullptr;
    HL_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_gpclk), &nul, sizeof(nul)));
    return 0;
}

int hlvae_gp_chain(const double* iK, const double* W, const double* HiK, const double* H, const double* iH, const double* m,
                   const double* P1, const double* u, double lr, double c, double g_alpha, double g_beta, int N, int batch,
                   double* T1, double* Bm, double* grad_m, double* grad_H, double* tmp, double* HiKW, double* Rs, double* T1b,
                   double* G, hlvae_stream s) {
    HL_REQUIRE(iK && W && HiK && H && iH && m && P1 && u && T1 && Bm && grad_m && grad_H && tmp && HiKW && Rs && T1b && G, HLVAE_EINVAL,
               "gp_chain: null argument");
    HL_REQUIRE(N >= 4 && N <= GP_MMAX && N % 4 == 0 && batch >= 1, HLVAE_ESHAPE, "gp_chain: N=%d (a multiple of 4, at most %d)", N, GP_MMAX);
    HL_REQUIRE(T1 != T1b && T1 != Bm && HiKW != Rs, HLVAE_EINVAL, "gp_chain: the intermediates must be distinct buffers");
    GpChainArgs a{iK, W, HiK, H, iH, m, P1, u, T1, Bm, grad_m, grad_H, tmp, HiKW, Rs, T1b, G, lr, c, g_alpha, g_beta, N};
    HL_PROF("gp_chain", (hipStream_t)s);
    static const bool wide = [] { const char* e = getenv("HL_GP_CHAIN_THREADS"); return e != nullptr && e[0] == '1'; }();   // =1024: A/B
    if (wide) k_gp_chain<1024><<<dim3(batch, 2), 1024, 0, (hipStream_t)s>>>(a);
    else k_gp_chain<512><<<dim3(batch, 2), 512, 0, (hipStream_t)s>>>(a);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_chain_rb(const double* iK, const double* W, const double* HiK, const double* H, const double* iH, const double* m,
                      const double* P1, const double* u, double lr, double c, double g_alpha, double g_beta, int N, int batch,
                      double* grad_m, double* grad_H, double* tmp, double* Rs, double* G, hlvae_stream s) {
    HL_REQUIRE(iK && W && HiK && H && iH && m && P1 && u && grad_m && grad_H && tmp && Rs && G, HLVAE_EINVAL, "gp_chain_rb: null argument");
    HL_REQUIRE(N >= 4 && N <= GP_MMAX && N % 4 == 0 && batch >= 1, HLVAE_ESHAPE, "gp_chain_rb: N=%d (a multiple of 4, at most %d)", N, GP_MMAX);
    const size_t smem = ((size_t)4 * 32 * GP_RB_LD + 3 * GP_MMAX + 32) * sizeof(double);
    static bool attr = false;
    if (!attr) {
        HL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gp_chain_rb<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        HL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gp_chain_rb<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = true;
    }
    GpChainRbArgs a{iK, W, HiK, H, iH, m, P1, u, grad_m, grad_H, tmp, Rs, G, lr, c, g_alpha, g_beta, N};
    const dim3 grid(batch, (N + 31) / 32);
    HL_PROF("gp_chain", (hipStream_t)s);
    k_gp_chain_rb<1><<<grid, 512, smem, (hipStream_t)s>>>(a);
    k_gp_chain_rb<2><<<grid, 512, smem, (hipStream_t)s>>>(a);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_bound(const double* part, int S, const double* W, const double* iK, const double* Qm, const double* H,
                   const double* m, const double* iKm, const double* ldK, const double* ldH, const float* lv, int B, int L,
                   int M, double c, double n_total, double rep, double* out, hlvae_stream s) {
    HL_REQUIRE(part && W && iK && Qm && H && m && iKm && ldK && ldH && lv && out, HLVAE_EINVAL, "gp_bound: null pointer");
    HL_CHECK(hipMemsetAsync(out, 0, sizeof(double), (hipStream_t)s));
    const double konst = -0.5 * (double)L * M - 0.5 * (double)L * n_total;
    HL_PROF("gp_bound", (hipStream_t)s);
    k_gp_bound<<<128, 256, 0, (hipStream_t)s>>>(part, S * L * 4, W, iK, Qm, H, L * M * M, m, iKm, L * M, ldK, ldH, L, lv, B * L, c,
                                              rep, konst, out);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_spd_inv2(const double* A, int n, int N, double* inv, double* logdet, int n_neg, double* logdet_neg, int* fail,
                      hlvae_stream s) {
    HL_REQUIRE(A && inv && logdet && n > 0 && N > 0 && N <= GP_MMAX && n_neg >= 0 && n_neg <= n && (n_neg == 0 || logdet_neg),
               HLVAE_EINVAL, "gp_spd_inv2: N=%d (max %d) n=%d n_neg=%d", N, GP_MMAX, n, n_neg);
    HL_PROF("gp_spd_inv", (hipStream_t)s);
    k_gp_spd_inv<<<n, 256, 0, (hipStream_t)s>>>(A, N, inv, logdet, fail, n_neg, logdet_neg);
    HL_LAUNCH_CHECK();
    return 0;
}

static int hl_gp_gemm_impl(const double* A, int lda, int64_t strideA, int transA, const double* B, int ldb, int64_t strideB, const double* D,
                           int ldd, int64_t strideD, double* C, int ldc, int64_t strideC, int M, int N, int K, int batch, double alpha,
                           double beta, hlvae_stream s, bool acc) {
    HL_REQUIRE(A && B && C && M >= 1 && N >= 1 && K >= 1 && batch >= 1, HLVAE_EINVAL, "gp_gemm: M=%d N=%d K=%d batch=%d", M, N, K, batch);
    HL_REQUIRE(lda >= (transA ? M : K) && ldb >= N && ldc >= N && (D == nullptr || ldd >= N), HLVAE_ESHAPE, "gp_gemm: leading dimensions");
    hipStream_t st = (hipStream_t)s;
    const int TM = M >= 256 ? 64 : 32;
    const int tiles_n = (N + 31) / 32, tiles_m = (M + TM - 1) / TM;
    // few output tiles and a long K (W = Kxz^T V: 16 tiles per latent, K = batch rows): slices of K add into a cleared C
    int ksplit = 1;
    while ((long)tiles_m * tiles_n * batch * ksplit < 1024 && K / (2 * ksplit) >= 4 * GP_GK) ksplit *= 2;
    if (ksplit > 1) {
        HL_REQUIRE(D == nullptr || D != C, HLVAE_EINVAL, "gp_gemm: D aliasing C is not available with split-K");
        HL_REQUIRE(ldc == N && strideC == (int64_t)M * N, HLVAE_ESHAPE, "gp_gemm: split-K needs a dense C");
        if (!acc) HL_CHECK(hipMemsetAsync(C, 0, sizeof(double) * (size_t)batch * M * N, st));
    }
    const int atomic = (ksplit > 1 || acc) ? 1 : 0;
    HL_PROF("gp_gemm", st);
    const dim3 grid(tiles_m * tiles_n, ksplit, batch);
#define HL_GG(TMv, TAv) k_gp_gemm<TMv, TAv><<<grid, 256, 0, st>>>(A, lda, strideA, B, ldb, strideB, D, ldd, strideD, C, ldc, strideC, M, \
                                                                N, K, tiles_n, ksplit, atomic, alpha, beta)
    if (TM == 64) { if (transA) HL_GG(64, 1); else HL_GG(64, 0); }
    else { if (transA) HL_GG(32, 1); else HL_GG(32, 0); }
#undef HL_GG
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_gemm(const double* A, int lda, int64_t strideA, int transA, const double* B, int ldb, int64_t strideB, const double* D,
                  int ldd, int64_t strideD, double* C, int ldc, int64_t strideC, int M, int N, int K, int batch, double alpha,
                  double beta, hlvae_stream s) {
    return hl_gp_gemm_impl(A, lda, strideA, transA, B, ldb, strideB, D, ldd, strideD, C, ldc, strideC, M, N, K, batch, alpha, beta, s, false);
}

// C[l] += alpha op(A[l]) B[l]: the caller cleared (or pre-loaded) C -- no memset node in front of the launch (W = Kxz^T V heads the
// critical chain of the GP step; its 7 us clear now rides with the state-only launches of prepare())
int hlvae_gp_gemm_acc(const double* A, int lda, int64_t strideA, int transA, const double* B, int ldb, int64_t strideB, double* C, int ldc,
                      int64_t strideC, int M, int N, int K, int batch, double alpha, hlvae_stream s) {
    return hl_gp_gemm_impl(A, lda, strideA, transA, B, ldb, strideB, nullptr, 0, 0, C, ldc, strideC, M, N, K, batch, alpha, 0.0, s, true);
}

int hlvae_gp_bmv(const double* A, const double* x, const double* y, double* out, int N, int batch, double alpha, double beta,
                 hlvae_stream s) {
    HL_REQUIRE(A && x && out && N >= 1 && N <= GP_MMAX && batch >= 1, HLVAE_EINVAL, "gp_bmv: N=%d batch=%d", N, batch);
    HL_PROF("gp_bmv", (hipStream_t)s);
    k_gp_bmv<<<batch, 1024, 0, (hipStream_t)s>>>(A, x, y, out, N, alpha, beta);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_resid(const double* Kxz, const double* w, const float* mu, int L, int B, int M, double* out, hlvae_stream s) {
    HL_REQUIRE(Kxz && w && mu && out && L >= 1 && B >= 1 && M >= 1, HLVAE_EINVAL, "gp_resid: bad arguments");
    const long rows = (long)L * B;
    HL_PROF("gp_resid", (hipStream_t)s);
    k_gp_resid<<<(int)((rows + 31) / 32), 256, 0, (hipStream_t)s>>>(Kxz, w, mu, L, B, M, out);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_gemv_t_f32(const double* A, const float* x, long x_stride_l, long x_stride_b, double* out, int L, int B, int M,
                        hlvae_stream s) {
    HL_REQUIRE(A && x && out && L >= 1 && B >= 1 && M >= 1 && M <= GP_MMAX, HLVAE_EINVAL, "gp_gemv_t_f32: L=%d B=%d M=%d", L, B, M);
    const int nc = gemv_chunks(L, B);
    if (nc > 1) HL_CHECK(hipMemsetAsync(out, 0, sizeof(double) * (size_t)L * M, (hipStream_t)s));
    HL_PROF("gp_gemv_t", (hipStream_t)s);
    k_gp_gemv_t<float><<<dim3(L, nc), 1024, 0, (hipStream_t)s>>>(A, x, x_stride_l, x_stride_b, out, B, M);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_natgrad(const double* Bm, const double* iK, const double* iH, const double* m, const double* P1, double lr, int N,
                     int batch, double* grad_m, double* grad_H, double* tmp, hlvae_stream s) {
    HL_REQUIRE(Bm && iK && iH && m && P1 && grad_m && grad_H && tmp && N >= 1 && N <= GP_MMAX && batch >= 1, HLVAE_EINVAL,
               "gp_natgrad: bad arguments");
    {
        HL_PROF("gp_natgrad", (hipStream_t)s);
        k_gp_natgrad<<<batch, 1024, 0, (hipStream_t)s>>>(Bm, iK, iH, m, P1, lr, N, grad_m, tmp);
        HL_LAUNCH_CHECK();
    }
    const int n = batch * N * N;
    HL_PROF("gp_natgrad_h", (hipStream_t)s);
    k_gp_natgrad_h<<<(n + 255) / 256, 256, 0, (hipStream_t)s>>>(Bm, iH, n, grad_H);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_natgrad_apply(const double* grad_H, double* iH, double lr, int N, int batch, hlvae_stream s) {
    HL_REQUIRE(grad_H && iH && N >= 1 && batch >= 1, HLVAE_EINVAL, "gp_natgrad_apply: bad arguments");
    const int n = batch * N * N;
    HL_PROF("gp_ih_update", (hipStream_t)s);
    k_gp_ih_update<<<(n + 255) / 256, 256, 0, (hipStream_t)s>>>(grad_H, iH, lr, N, n);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_adam(double* p, double* g, double* m1, double* m2, int n, int64_t* step, double lr, double b1, double b2,
                  double eps, hlvae_stream s) {
    HL_REQUIRE(p && g && m1 && m2 && step && n > 0, HLVAE_EINVAL, "gp_adam: bad arguments");
    HL_PROF("gp_adam", (hipStream_t)s);
    k_gp_adam<<<(n + 255) / 256, 256, 0, (hipStream_t)s>>>(p, g, m1, m2, n, step, lr, b1, b2, eps);
    HL_LAUNCH_CHECK();
    return 0;
}

int hlvae_gp_state_head(double* p, double* g, double* m1, double* m2, int n, int64_t* step, double lr, double b1, double b2,
                        double eps, int n_slots, int L, double* hyp, const double* grad_H, double* iH, double ng_lr, int N, int batch,
                        hlvae_stream s) {
    HL_REQUIRE(p && g && m1 && m2 && step && hyp && grad_H && iH && n > 0 && n_slots > 0 && L > 0 && n_slots * L <= n && N >= 1 && batch >= 1,
               HLVAE_EINVAL, "gp_state_head: bad arguments");
    const int blocks_adam = (n + 255) / 256, n_ih = batch * N * N;
    HL_PROF("gp_adam", (hipStream_t)s);
    k_gp_state_head<<<blocks_adam + (n_ih + 255) / 256, 256, 0, (hipStream_t)s>>>(p, g, m1, m2, n, step, lr, b1, b2, eps, n_slots * L, hyp, grad_H,
                                                                                   iH, ng_lr, N, n_ih, blocks_adam);
    HL_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
