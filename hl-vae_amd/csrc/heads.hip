// y_layer GEMM fused with the per-type heads, the five log-likelihoods, the ELBO row sums and the
// backward of all of them (SURVEY.md section 8(a) rows D tail, E, F, G, H, I-cat, I-ord, J, M).
//
// Tile = BM batch rows x 16 variables (= 16*YD columns of Y = U * Wy^T + by, column d*YD + k is
// feature k of variable d: the reshape of HLVAE.py:343).  The fp32 accumulator tile is staged in
// LDS; thread (v = tid & 15, rg = tid >> 4) then owns variable v of the tile for the rows HL_ROW(rg, i)
// = 4 rg + (i & 3) + 64 (i >> 2): groups of FOUR CONSECUTIVE rows, so that the transposed copy of dY leaves
// straight from the thread's own values as 8-byte stores (4 bf16 of one column), and with the tile's row
// stride CLD = 16 YD + 4 every LDS access of the epilogue is conflict-free (the two row groups of a
// 32-lane half sit 4 CLD = 16 (mod 32) banks apart, exactly the gap the 16 variables' YD-strided
// columns leave; PMC before: 38 % of the LDS cycles were bank conflicts).
// The type of a variable is a per-thread constant, its head weights live in registers, and
// mixed-type columns cost no gather/scatter (the reference walks boolean masks per type block,
// HLVAE.py:387-412, 422-452).  Y itself is never written to HBM: the tile is overwritten in LDS by
// g * d log_p_x / d Y and leaves as bf16 in both layouts for the two backward GEMMs.
//
// Stop-gradient through missing entries (HLVAE.py:435-452): theta = head(y) everywhere, gradient
// only where the mask is 1 -> dY and all head-parameter gradients are gated by the mask.
#include <stdlib.h>
#include "gemm_nt.h"
#include "gemm_dma.h"

#define HL_LOG2PI 1.8378770664093453f
#define HL_ROW(rg, i) (4 * (rg) + ((i) & 3) + 64 * ((i) >> 2))

// sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), result in every lane: four VALU adds with DPP operands
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror) instead of four ds_bpermute round trips
__device__ __forceinline__ float row16_sum(float s) {
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0xB1, 0xF, 0xF, true));
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x4E, 0xF, 0xF, true));
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x141, 0xF, 0xF, true));
    s += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x140, 0xF, 0xF, true));
    return s;
}

template <int YD, int KM>
struct HeadAcc {
    static constexpr int N = (YD + 1) * (KM - 1) > (YD + KM) ? (YD + 1) * (KM - 1) : (YD + KM);
};

// ---- real / pos:  HL_VAE/loglik.py:27-70 and :73-121 -------------------------------------------
// LVN (logvar_network, HLVAE.py:25-51; a property of the whole model, so a compile-time variant chosen by a uniform branch):
// the head has a second output, the log-variance parameter of every ENTRY (theta[:, n + j], loglik.py:45-47 / :105) instead of
// the free per-variable parameter _log_vy_*; its weights sit behind the mean's in the staged parameter block (w[YD .. 2 YD),
// b[1]); accumulators: mean weights / bias at 0 .. YD, log-variance weights / bias at YD + 1 .. 2 YD + 1.
// (A single function selecting per lane between the two forms compiled into a kernel whose dY cells were occasionally wrong --
// one 16-lane group's store of one column, differing from run to run; tests/test_gpu_configs.py::test_step_is_deterministic.)
template <int YD, int BM, int CLD, int NACC, bool LVN>
__device__ __forceinline__ void proc_realpos(bool is_pos, float* Cs, int v, int rg, int m0, int B, int D, int d,
                                             const hlvae_var& var, const float* __restrict__ P,
                                             const float* __restrict__ norm, int n_stat, const float (&byv)[YD],
                                             const float* __restrict__ xt, const uint8_t* __restrict__ m8,
                                             const float* __restrict__ g_elem, float g_scale,
                                             float* __restrict__ logpx, float* __restrict__ logpx_miss,
                                             float* __restrict__ pfull, int X, float* __restrict__ xhat,
                                             float (&acc)[NACC], float (&lpo)[BM / 16], bool conv_real) {
    // conv_real: real variable under the convolutional decoder -- sigmoid on the mean (HLVAE.py:271-273, 428-430), data
    // scaled by 1/255 (HLVAE.py:393-394), no batch statistics (norm holds mean 0 / var 1; loglik.py:40-41)
    float w[YD], w2[LVN ? YD : 1];
#pragma unroll
    for (int k = 0; k < YD; ++k) w[k] = P[var.w_off + k];
    if (LVN) {
#pragma unroll
        for (int k = 0; k < YD; ++k) w2[k] = P[var.w_off + YD + k];
    }
    const float b = P[var.b_off];
    const float b2 = LVN ? P[var.b_off + 1] : 0.f;
    const float mean_d = norm[var.sidx];
    float vd = norm[n_stat + var.sidx];
    const float xscale = conv_real ? 1.f / 255.f : 1.f;
    vd = fmaxf(vd, is_pos ? 1e-3f : 3e-4f);                          // loglik.py:38 / :80
    float ev = 1.f, dp_fac = 1.f, pos_var = 0.f;
    if (!LVN) {                                                      // variance from the free per-variable parameter
        const float p = P[var.e_off];
        if (!is_pos) {
            ev = vd * __expf(-8.f + softplus_f(p + 8.f));            // :51-52, :56
            dp_fac = sigmoid_f(p + 8.f);
        } else {
            ev = vd * __expf(p);                                     // :100
            pos_var = __expf(p);                                     // read_functions.py:285
        }
    }
    const float sd = sqrtf(vd);
    float inv_ev = 1.f / ev;
    float c0 = -0.5f * HL_LOG2PI - 0.5f * __logf(ev);
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int r = HL_ROW(rg, i), gr = m0 + r;
        float* yrow = Cs + r * CLD + v * YD;
        float th = b, tl = b2;
        float y[YD];
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            y[k] = yrow[k] + byv[k];
            th += w[k] * y[k];
            if (LVN) tl += w2[k] * y[k];
        }
        if (LVN) {                                                   // variance of THIS entry from the head's second output
            if (!is_pos) {
                ev = vd * __expf(-8.f + softplus_f(tl + 8.f));       // :45-47, :56
                dp_fac = sigmoid_f(tl + 8.f);
            } else {
                ev = vd * __expf(tl);                                // :105
            }
            inv_ev = 1.f / ev;
            c0 = -0.5f * HL_LOG2PI - 0.5f * __logf(ev);
        }
        float lp_obs = 0.f, dth = 0.f, dtl = 0.f;
        if (gr < B) {
            const size_t o = (size_t)gr * D + d;
            const float x = xt[i * HL_THREADS] * xscale;         // raw x (real) or log1p x (pos): this thread's slot of the prefetched tile
            const bool ob = m8[i * HL_THREADS] != 0;
            float thv = th, dsig = 1.f;
            if (conv_real) {
                thv = sigmoid_f(th);
                dsig = thv * (1.f - thv);
            }
            const float mean = sd * thv + mean_d;                // :55 / :96
            const float rr = x - mean;
            float lp = -0.5f * rr * rr * inv_ev + c0;            // :58 / :102
            if (is_pos) lp -= x;
            logpx[o] = ob ? lp : 0.f;
            logpx_miss[o] = ob ? 0.f : lp;
            if (ob) {
                const float g = g_elem != nullptr ? g_elem[o] : g_scale;
                lp_obs = lp;
                dth = g * rr * inv_ev * sd * dsig;
                dtl = g * (0.5f * rr * rr * inv_ev - 0.5f) * dp_fac;
            }
            if (pfull != nullptr) {
                pfull[(size_t)gr * X + var.poff] = mean;         // loglik.py:64-67 (mean only unless logvar_network: [mean, var])
                if (LVN) pfull[(size_t)gr * X + var.poff2] = ev;
            }
            // read_functions.py:277, 283-290: under logvar_network the variance in the pos mean is est_var itself
            if (xhat != nullptr) xhat[o] = is_pos ? __expf(mean + 0.5f * (LVN ? ev : pos_var)) - 1.f : mean;
        }
        lpo[i] = lp_obs;
        acc[YD] += dth;
        acc[LVN ? 2 * YD + 1 : YD + 1] += dtl;
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            acc[k] += dth * y[k];
            if (LVN) {
                acc[YD + 1 + k] += dtl * y[k];
                yrow[k] = dth * w[k] + dtl * w2[k];
            } else {
                yrow[k] = dth * w[k];
            }
        }
    }
}

// ---- count:  HL_VAE/loglik.py:191-213 ------------------------------------------------------------
template <int YD, int BM, int CLD, int NACC>
__device__ __forceinline__ void proc_count(float* Cs, int v, int rg, int m0, int B, int D, int d, const hlvae_var& var,
                                           const float* __restrict__ P, const float (&byv)[YD],
                                           const float* __restrict__ xt, const uint8_t* __restrict__ m8,
                                           const float* __restrict__ g_elem, float g_scale, float* __restrict__ logpx,
                                           float* __restrict__ logpx_miss, float* __restrict__ pfull, int X,
                                           float* __restrict__ xhat, float (&acc)[NACC], float (&lpo)[BM / 16]) {
    float w[YD];
#pragma unroll
    for (int k = 0; k < YD; ++k) w[k] = P[var.w_off + k];
    const float b = P[var.b_off];
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int r = HL_ROW(rg, i), gr = m0 + r;
        float* yrow = Cs + r * CLD + v * YD;
        float th = b;
        float y[YD];
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            y[k] = yrow[k] + byv[k];
            th += w[k] * y[k];
        }
        float lp_obs = 0.f, dth = 0.f;
        if (gr < B) {
            const size_t o = (size_t)gr * D + d;
            const float x = xt[i * HL_THREADS];
            const bool ob = m8[i * HL_THREADS] != 0;
            const float sp = softplus_f(th);
            const float lam = fminf(fmaxf(sp, 1e-6f), 1e20f);    // :203
            const float lp = x * __logf(lam) - lam - lgammaf(x + 1.f);   // Poisson.log_prob
            logpx[o] = ob ? lp : 0.f;
            logpx_miss[o] = ob ? 0.f : lp;
            if (ob) {
                const float g = g_elem != nullptr ? g_elem[o] : g_scale;
                lp_obs = lp;
                if (sp >= 1e-6f && sp <= 1e20f) dth = g * (x / lam - 1.f) * sigmoid_f(th);
            }
            if (pfull != nullptr) pfull[(size_t)gr * X + var.poff] = lam;
            if (xhat != nullptr) xhat[o] = lam;                  // read_functions.py:294
        }
        lpo[i] = lp_obs;
        acc[YD] += dth;
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            acc[k] += dth * y[k];
            yrow[k] = dth * w[k];
        }
    }
}

// ---- categorical:  Observation_Cat (HLVAE.py:54-68) + loglik_cat (loglik.py:124-146) -------------
// theta_0 = 0, theta_j = b_j + sum_k W[k][j-1] y_k.  accumulators: gW at k*(KM-1)+(j-1), gb at YD*(KM-1)+(j-1)
template <int YD, int BM, int CLD, int NACC, int KM>
__device__ __forceinline__ void proc_cat(float* Cs, int v, int rg, int m0, int B, int D, int d, const hlvae_var& var,
                                         const float* __restrict__ P, const float (&byv)[YD],
                                         const float* __restrict__ xt, const uint8_t* __restrict__ m8,
                                         const float* __restrict__ g_elem, float g_scale, float* __restrict__ logpx,
                                         float* __restrict__ logpx_miss, float* __restrict__ pfull, int X,
                                         float* __restrict__ xhat, float (&acc)[NACC], float (&lpo)[BM / 16]) {
    const int K = var.ncls;
    float W[YD][KM - 1], bb[KM - 1];
#pragma unroll
    for (int j = 0; j < KM - 1; ++j) {
        bb[j] = j < K - 1 ? P[var.b_off + j] : 0.f;
#pragma unroll
        for (int k = 0; k < YD; ++k) W[k][j] = j < K - 1 ? P[var.w_off + k * (K - 1) + j] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int r = HL_ROW(rg, i), gr = m0 + r;
        float* yrow = Cs + r * CLD + v * YD;
        float y[YD], th[KM - 1];
#pragma unroll
        for (int k = 0; k < YD; ++k) y[k] = yrow[k] + byv[k];
        float mx = 0.f;                                          // theta_0 = 0 (HLVAE.py:66-67)
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) {
            float t = bb[j];
#pragma unroll
            for (int k = 0; k < YD; ++k) t += W[k][j] * y[k];
            th[j] = t;
            if (j < K - 1) mx = fmaxf(mx, t);
        }
        float se = __expf(-mx), ex[KM - 1];
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) {
            ex[j] = j < K - 1 ? __expf(th[j] - mx) : 0.f;
            se += ex[j];
        }
        const float lse = mx + __logf(se);                       // loglik.py:134
        const float inv_se = __frcp_rn(se);                      // softmax_j = ex[j] / se: the gradient reuses the exponentials
        float dth[KM - 1];
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) dth[j] = 0.f;
        float lp_obs = 0.f;
        if (gr < B) {
            const size_t o = (size_t)gr * D + d;
            const int cls = (int)xt[i * HL_THREADS];             // -1: all-zero one-hot row
            const bool ob = m8[i * HL_THREADS] != 0;
            float lp = 0.f;
            if (cls == 0) lp = -lse;
#pragma unroll
            for (int j = 0; j < KM - 1; ++j)
                if (cls == j + 1) lp = th[j] - lse;              // :135 (one-hot data)
            logpx[o] = ob ? lp : 0.f;
            logpx_miss[o] = ob ? 0.f : lp;
            if (ob && cls >= 0) {
                const float g = g_elem != nullptr ? g_elem[o] : g_scale;
                lp_obs = lp;
#pragma unroll
                for (int j = 0; j < KM - 1; ++j)
                    if (j < K - 1) dth[j] = g * ((cls == j + 1 ? 1.f : 0.f) - ex[j] * inv_se);
            } else if (ob) {
                lp_obs = lp;
            }
            if (pfull != nullptr) {
                float* pf = pfull + (size_t)gr * X + var.poff;
                pf[0] = -lse;                                    // params = normalised log_pi (:139)
#pragma unroll
                for (int j = 0; j < KM - 1; ++j)
                    if (j < K - 1) pf[j + 1] = th[j] - lse;
            }
            if (xhat != nullptr) {
                int am = 0;
                float best = 0.f;
#pragma unroll
                for (int j = 0; j < KM - 1; ++j)
                    if (j < K - 1 && th[j] > best) { best = th[j]; am = j + 1; }
                xhat[o] = (float)am;                             // read_functions.py:297-299 (first max)
            }
        }
        lpo[i] = lp_obs;
        float dy[YD];
#pragma unroll
        for (int k = 0; k < YD; ++k) dy[k] = 0.f;
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) {
            acc[YD * (KM - 1) + j] += dth[j];
#pragma unroll
            for (int k = 0; k < YD; ++k) {
                acc[k * (KM - 1) + j] += dth[j] * y[k];
                dy[k] += dth[j] * W[k][j];
            }
        }
#pragma unroll
        for (int k = 0; k < YD; ++k) yrow[k] = dy[k];
    }
}

// ---- ordinal:  Observation_Ordinal (HLVAE.py:70-89) + loglik_ordinal (loglik.py:149-188) ---------
// accumulators: gw[k] at k, gb at YD, g_thresholds[j] at YD+1+j
template <int YD, int BM, int CLD, int NACC, int KM>
__device__ __forceinline__ void proc_ord(float* Cs, int v, int rg, int m0, int B, int D, int d, const hlvae_var& var,
                                         const float* __restrict__ P, const float (&byv)[YD],
                                         const float* __restrict__ xt, const uint8_t* __restrict__ m8,
                                         const float* __restrict__ g_elem, float g_scale, float* __restrict__ logpx,
                                         float* __restrict__ logpx_miss, float* __restrict__ pfull, int X,
                                         float* __restrict__ xhat, float (&acc)[NACC], float (&lpo)[BM / 16]) {
    const int K = var.ncls;
    float w[YD];
#pragma unroll
    for (int k = 0; k < YD; ++k) w[k] = P[var.w_off + k];
    const float b = P[var.b_off];
    float thr[KM - 1], dthr_fac[KM - 1];                         // cumulative thresholds are row-independent
    {
        float cs = 0.f;
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) {
            float a = 0.f, f = 0.f;
            if (j < K - 1) {
                const float t = P[var.e_off + j];
                const float sp = softplus_f(t);
                a = fminf(fmaxf(sp, 1e-6f), 1e20f);              // loglik.py:164
                f = (sp >= 1e-6f && sp <= 1e20f) ? sigmoid_f(t) : 0.f;
            }
            cs += a;
            thr[j] = cs;
            dthr_fac[j] = f;
        }
    }
#pragma unroll
    for (int i = 0; i < BM / 16; ++i) {
        const int r = HL_ROW(rg, i), gr = m0 + r;
        float* yrow = Cs + r * CLD + v * YD;
        float y[YD];
        float reg = b;
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            y[k] = yrow[k] + byv[k];
            reg += w[k] * y[k];
        }
        const float mv = softplus_f(reg);                        // :163
        float sg[KM - 1];
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) sg[j] = j < K - 1 ? sigmoid_f(thr[j] - mv) : 1.f;   // :165
        float pc[KM];                                            // clamped class probabilities (:166-169)
        bool pass[KM];
        float S = 0.f;
#pragma unroll
        for (int c = 0; c < KM; ++c) {
            float pr = 0.f;
            if (c < K) {
                const float hi = (c < K - 1) ? sg[c < KM - 1 ? c : KM - 2] : 1.f;
                const float lo = (c > 0) ? sg[c - 1 >= 0 ? c - 1 : 0] : 0.f;
                pr = hi - lo;
            }
            pass[c] = (c < K) && pr >= 1e-6f && pr <= 1.f;
            pc[c] = c < K ? fminf(fmaxf(pr, 1e-6f), 1.f) : 0.f;
            S += pc[c];
        }
        const float invS = 1.f / S;
        float dreg = 0.f, lp_obs = 0.f;
        float du[KM - 1];
#pragma unroll
        for (int j = 0; j < KM - 1; ++j) du[j] = 0.f;
        if (gr < B) {
            const size_t o = (size_t)gr * D + d;
            const bool ob = m8[i * HL_THREADS] != 0;
            int cls = ob ? (int)xt[i * HL_THREADS] : 0;          // :172-174 (masked rows -> class 0)
            cls = cls < 0 ? 0 : (cls > K - 1 ? K - 1 : cls);
            float pcc = pc[0];
#pragma unroll
            for (int c = 1; c < KM; ++c)
                if (cls == c) pcc = pc[c];
            const float lp = __logf(pcc * invS);                 // :178-179
            logpx[o] = ob ? lp : 0.f;
            logpx_miss[o] = ob ? 0.f : lp;
            if (ob) {
                const float g = g_elem != nullptr ? g_elem[o] : g_scale;
                lp_obs = lp;
                float dp[KM];
#pragma unroll
                for (int c = 0; c < KM; ++c)
                    dp[c] = pass[c] ? g * ((cls == c ? 1.f / pcc : 0.f) - invS) : 0.f;
                float dmv = 0.f;
#pragma unroll
                for (int j = 0; j < KM - 1; ++j)
                    if (j < K - 1) {
                        const float ds = dp[j] - dp[j + 1];      // s_j enters p_j (+) and p_{j+1} (-)
                        du[j] = ds * sg[j] * (1.f - sg[j]);
                        dmv -= du[j];
                    }
                dreg = dmv * sigmoid_f(reg);
            }
            if (pfull != nullptr) {
                float* pf = pfull + (size_t)gr * X + var.poff;
#pragma unroll
                for (int c = 0; c < KM; ++c)
                    if (c < K) pf[c] = pc[c] * invS;             // params = normalised mean_probs (:183)
            }
            if (xhat != nullptr) {
                int am = 0;
                float best = pc[0];
#pragma unroll
                for (int c = 0; c < KM; ++c)
                    if (c < K && pc[c] > best) { best = pc[c]; am = c; }
                xhat[o] = (float)am;
            }
        }
        lpo[i] = lp_obs;
        // d thresholds: thr_j = sum_{i<=j} a_i  ->  d a_i = sum_{j>=i} du_j
        float run = 0.f;
#pragma unroll
        for (int j = KM - 2; j >= 0; --j) {
            run += du[j];
            acc[YD + 1 + j] += run * dthr_fac[j];
        }
        acc[YD] += dreg;
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            acc[k] += dreg * y[k];
            yrow[k] = dreg * w[k];
        }
    }
}

// arena offset of accumulator n of a variable (or -1)
template <int YD, int KMAX>
__device__ __forceinline__ int acc_dest(const hlvae_var& var, int n) {
    const int K = var.ncls;
    switch (var.kind) {
        case HLVAE_REAL:
        case HLVAE_POS:
            if (var.w2_off >= 0)      // logvar_network: [mean weights, mean bias, log-variance weights, log-variance bias]
                return n < YD ? var.w_off + n : (n == YD ? var.b_off : (n <= 2 * YD ? var.w2_off + (n - YD - 1) : (n == 2 * YD + 1 ? var.b2_off : -1)));
            return n < YD ? var.w_off + n : (n == YD ? var.b_off : (n == YD + 1 ? var.e_off : -1));
        case HLVAE_COUNT:
            return n < YD ? var.w_off + n : (n == YD ? var.b_off : -1);
        case HLVAE_CAT: {
            const int KM = (KMAX <= 3 || K <= 3) ? 3 : ((KMAX <= 5 || K <= 5) ? (KMAX < 5 ? KMAX : 5) : KMAX);
            if (n < YD * (KM - 1)) {
                const int k = n / (KM - 1), j = n % (KM - 1);
                return j < K - 1 ? var.w_off + k * (K - 1) + j : -1;
            }
            const int j = n - YD * (KM - 1);
            return j < K - 1 ? var.b_off + j : -1;
        }
        case HLVAE_ORDINAL:
            if (n < YD) return var.w_off + n;
            if (n == YD) return var.b_off;
            return (n - YD - 1) < K - 1 ? var.e_off + (n - YD - 1) : -1;
    }
    return -1;
}

// DMA: the U x Wy^T tile runs on the LDS-DMA core (gemm_dma.h, round 3: no staging registers, no ds_write pass, swizzled
// conflict-free fragment reads, two 20 KB buffers instead of two 23 KB ones); false: the register-staged core of rounds 1-2
template <int YD, int BM, int KMAX, int CORE = 2>
__global__ __launch_bounds__(HL_THREADS, (YD > 5 || KMAX > 8) ? 1 : (KMAX <= 5 ? 3 : 2)) void k_y_heads(      // (y_dim 8: 245 VGPRs)
    const bf16_t* __restrict__ U, int ldu, const bf16_t* __restrict__ Wy, int K, const hlvae_var* __restrict__ vars,
    const float* __restrict__ P, float* __restrict__ hgpart, long o_by, const float* __restrict__ norm, int n_stat,
    const float* __restrict__ xt, const uint8_t* __restrict__ m8, int D, const float* __restrict__ g_elem, float g_scale,
    bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dyT, int Bp, float* __restrict__ logpx,
    float* __restrict__ logpx_miss, float* __restrict__ rowpart, float* __restrict__ pfull, int X,
    float* __restrict__ xhat, int B, int want_grad, const float* __restrict__ ysrc, int ldys, long long* __restrict__ clk,
    int logvar, unsigned long long* stamp) {
    HL_STAMP_T0(stamp);
#define HL_CLK(i) do { if (clk != nullptr && (threadIdx.x & 63) == 0) clk[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 12 + (i)] = clock64(); } while (0)
    HL_CLK(0);
    // ysrc != nullptr: convolutional decoder -- the tile of y_grouped comes from the second ConvTranspose (csrc/conv.hip,
    // bias included) instead of the y_layer GEMM; d Y leaves in row-major layout only and d by is not ours
    constexpr int BN = 16 * YD;
    // CORE 2: B through LDS-DMA (four stages), A fragments in registers; 1: [A; B] through LDS-DMA (two stages); 0: register-staged
    using Gm = typename std::conditional<CORE == 2, GemmDMAB<BM, BN, (BN > 80 ? 3 : 4), BN + 4>,
                                         typename std::conditional<CORE == 1, GemmDMA<BM, BN, 4, 1, 2, BN + 4>, GemmNT<BM, BN, 64, 4, 1, 3, BN + 4>>::type>::type;
    constexpr int CLD = Gm::CLD;
    constexpr int RPT = BM / 16;                                  // rows per thread
    constexpr int NHEAD = HeadAcc<YD, KMAX>::N;                   // head-parameter gradient accumulators of a variable
    constexpr int NACC = NHEAD + YD;                              // + d by of its YD columns
    constexpr int RST = 272;                                      // row stride of the reduction image: 16 banks between rows
    // scratch behind the GEMM buffers, used strictly BEFORE the gradient reduction image that overlays them:
    //   [head parameters, statistics and y_layer bias of the 16 variables | likelihood targets | masks] of the tile
    constexpr int PW = YD * (KMAX - 1), PB = KMAX - 1, PS = PW + 2 * PB + 2 + YD;      // [w | b | e | mean, var | by]
    static_assert(PW >= 2 * YD && PB >= 2 && HeadAcc<YD, KMAX>::N >= 2 * YD + 2, "room for the log-variance head of real / pos variables");
    constexpr int SCR_BYTES = 16 * PS * 4 + RPT * HL_THREADS * 5;
    // after the epilogue: [C tile with dY | acc[n][thread] image]
    constexpr int POST_BYTES = BM * CLD * 4 + NACC * RST * 4;
    constexpr int SMEM_TOTAL = Gm::SMEM_BYTES + SCR_BYTES > POST_BYTES ? Gm::SMEM_BYTES + SCR_BYTES : POST_BYTES;
    __shared__ __attribute__((aligned(1024))) char smem[SMEM_TOTAL];
    // 1-D grid, XCD-aware.  Up to 8 row blocks (512 rows): the row blocks that share one 16-variable panel of Wy get
    // consecutive logical ids and therefore one XCD's L2; U (<= 0.5 MB) stays in every L2.  Larger batches: U no longer fits
    // beside the streams of a 4 MB L2 and was re-read for every panel an XCD owns (PMC at 4096 rows: 630 MB fetched + written
    // for 207 MB of algorithmic traffic, the kernel ran AT the HBM rate).  There every XCD owns tiles_m / 8 row blocks -- its
    // slice of U stays resident -- and walks all the panels, the row blocks of one panel side by side: Wy streams once per XCD.
    const int tiles_m = Bp / BM;
    int tn, m0;
    if (tiles_m >= 16 && (tiles_m & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, rbx = tiles_m >> 3;       // hardware XCD, index inside it
        tn = j / rbx;
        m0 = (xcd * rbx + j % rbx) * BM;
    } else {
        const int lid = xcd_remap(blockIdx.x, gridDim.x);
        tn = lid / tiles_m;
        m0 = (lid % tiles_m) * BM;
    }
    const int n0 = tn * BN;
    const int NY = D * YD;
    float* Cs = reinterpret_cast<float*>(smem);
    const bool conv = ysrc != nullptr;
    float* pscr = reinterpret_cast<float*>(smem + Gm::SMEM_BYTES);
    float* xs = pscr + 16 * PS;                                   // [RPT][256] this thread's targets at xs[i * 256 + tid]
    uint8_t* ms = reinterpret_cast<uint8_t*>(xs + RPT * HL_THREADS);
    const int tid = threadIdx.x, v = tid & 15, rg = tid >> 4;
    const int d = tn * 16 + v;                                    // position in the kernel's variable order (`vars` is in that order)
    hlvae_var var = vars[d < D ? d : D - 1];                     // unconditional (a conditional copy makes hipcc wait for it at once);
    const int dv = var.pad;                                       // the variable's own index: columns of the [B, D] buffers, y_layer's bias
    // Everything the epilogue reads from global memory is REQUESTED here, ahead of the GEMM, and parked in LDS by the hook
    // below while the GEMM's own first tiles are still in flight: the likelihood targets and masks of this thread's rows
    // (HBM-cold: they used to cost every wave a full memory latency at the top of the epilogue) and, by 16 lanes (one per
    // variable of the tile), the dependent chain vars[d] -> arena offsets -> P[...] of the heads' parameters.
    float xv[RPT];
    uint8_t mv[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int gr = min(m0 + HL_ROW(rg, i), B - 1);               // clamped, unconditional loads; rows >= B are masked when parked
        const size_t o = (size_t)gr * D + dv;
        xv[i] = xt[o];
        mv[i] = m8[o];
    }
    // hook, run between the issue of the GEMM's first three k-tiles and its first LDS write: park the targets / masks (their
    // loads are older than the tiles': no extra wait) and ISSUE the heads' parameter loads, which hang off vars[d] (one memory
    // latency, now overlapped with the tiles' instead of exposed ahead of the GEMM: 3.4 k of 43 k clocks per wave).  The values
    // stay in registers for the first three k-steps only and are parked then (keeping them through the loop spilled 29 VGPRs).
    float pre[PS];
    auto load_params = [&]() {
        if (tid < 16 && d < D) {
            const int K1 = var.ncls - 1;
            const bool cont = var.kind == HLVAE_REAL || var.kind == HLVAE_POS;
            const bool lvn = cont && var.w2_off >= 0;
            const int nw = var.kind == HLVAE_CAT ? YD * K1 : YD, nb = var.kind == HLVAE_CAT ? K1 : 1;
            const int ne = cont ? (lvn ? 0 : 1) : (var.kind == HLVAE_ORDINAL ? K1 : 0);
#pragma unroll
            for (int i = 0; i < PW; ++i)       // logvar_network: the log-variance weights ride behind the mean's (PW >= 2 YD)
                pre[i] = i < nw ? P[var.w_off + i] : ((lvn && i < 2 * YD) ? P[var.w2_off + (i - YD)] : 0.f);
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                pre[PW + i] = i < nb ? P[var.b_off + i] : ((lvn && i == 1) ? P[var.b2_off] : 0.f);
                pre[PW + PB + i] = i < ne ? P[var.e_off + i] : 0.f;
            }
            pre[PW + 2 * PB] = cont ? norm[var.sidx] : 0.f;
            pre[PW + 2 * PB + 1] = cont ? norm[n_stat + var.sidx] : 1.f;
#pragma unroll
            for (int k = 0; k < YD; ++k) pre[PW + 2 * PB + 2 + k] = conv ? 0.f : P[o_by + (long)dv * YD + k];
        }
    };
    auto park_params = [&]() {
        if (tid < 16 && d < D) {
#pragma unroll
            for (int i = 0; i < PS; ++i) pscr[v * PS + i] = pre[i];
        }
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            xs[i * HL_THREADS + tid] = xv[i];
            ms[i * HL_THREADS + tid] = (m0 + HL_ROW(rg, i) < B && d < D) ? mv[i] : (uint8_t)0;
        }
        // CORE 2: the parameter loads were issued ahead of the GEMM (right behind the targets': the arena offsets arrive in the
        // same cache line as var.pad) and are parked here too -- one straight-line k-step, so the compiler's wait for them is
        // counted; CORE 0 / 1: issued here, parked one k-step later
        if constexpr (CORE == 2) park_params(); else load_params();
    };
    auto park_late = [&]() {
        if constexpr (CORE != 2) park_params();
    };
    if constexpr (CORE == 2) load_params();
    HL_CLK(1);
    if (!conv) {
        typename Gm::Acc accm;
        Gm::zero(accm);
        Gm::run(U, ldu, Wy, ldu, m0, n0, Bp, NY, 0, K, smem, accm, park, park_late);
        HL_CLK(2);
        Gm::to_lds(accm, smem);
        HL_CLK(3);
    } else {
        park();
        park_late();
        for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
            const int r = idx / BN, c = idx % BN;
            Cs[r * CLD + c] = (m0 + r < B && n0 + c < NY) ? ysrc[(size_t)(m0 + r) * ldys + n0 + c] : 0.f;
        }
        __syncthreads();
    }
    float acc[NACC];
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = 0.f;
    float lpo[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) lpo[i] = 0.f;
    if (d < D) {
        float byv[YD];
#pragma unroll
        for (int k = 0; k < YD; ++k) byv[k] = pscr[v * PS + PW + 2 * PB + 2 + k];
        // from here on the "arena", the statistics, the targets and the masks are this thread's slices of the LDS scratch
        const float* P = pscr;
        const float* norm = pscr;
        const float* xt = xs + tid;
        const uint8_t* m8 = ms + tid;
        const int n_stat = 1;
        var.w_off = v * PS;
        var.b_off = v * PS + PW;
        var.e_off = v * PS + PW + PB;
        var.sidx = v * PS + PW + 2 * PB;
        float (&hacc)[NHEAD] = *reinterpret_cast<float (*)[NHEAD]>(&acc[0]);
        switch (var.kind) {
            case HLVAE_REAL:
            case HLVAE_POS: {
                const bool is_pos = var.kind == HLVAE_POS;
                if (logvar)      // (uniform: a kernel argument)
                    proc_realpos<YD, BM, CLD, NHEAD, true>(is_pos, Cs, v, rg, m0, B, D, dv, var, P, norm, n_stat, byv, xt, m8, g_elem,
                                                           g_scale, logpx, logpx_miss, pfull, X, xhat, hacc, lpo, conv && !is_pos);
                else
                    proc_realpos<YD, BM, CLD, NHEAD, false>(is_pos, Cs, v, rg, m0, B, D, dv, var, P, norm, n_stat, byv, xt, m8, g_elem,
                                                            g_scale, logpx, logpx_miss, pfull, X, xhat, hacc, lpo, conv && !is_pos);
                break;
            }
            case HLVAE_COUNT:
                proc_count<YD, BM, CLD, NHEAD>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem, g_scale, logpx,
                                               logpx_miss, pfull, X, xhat, hacc, lpo);
                break;
            case HLVAE_CAT:
                if (KMAX <= 3 || var.ncls <= 3)
                    proc_cat<YD, BM, CLD, NHEAD, 3>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem, g_scale, logpx,
                                                    logpx_miss, pfull, X, xhat, hacc, lpo);
                else if (KMAX <= 5 || var.ncls <= 5)
                    proc_cat<YD, BM, CLD, NHEAD, (KMAX < 5 ? KMAX : 5)>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem,
                                                                       g_scale, logpx, logpx_miss, pfull, X, xhat, hacc, lpo);
                else
                    proc_cat<YD, BM, CLD, NHEAD, KMAX>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem, g_scale, logpx,
                                                       logpx_miss, pfull, X, xhat, hacc, lpo);
                break;
            case HLVAE_ORDINAL:
                if (KMAX <= 5 || var.ncls <= 5)
                    proc_ord<YD, BM, CLD, NHEAD, (KMAX < 5 ? KMAX : 5)>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem,
                                                                       g_scale, logpx, logpx_miss, pfull, X, xhat, hacc, lpo);
                else
                    proc_ord<YD, BM, CLD, NHEAD, KMAX>(Cs, v, rg, m0, B, D, dv, var, P, byv, xt, m8, g_elem, g_scale, logpx,
                                                       logpx_miss, pfull, X, xhat, hacc, lpo);
                break;
        }
    } else {   // columns past the last variable: keep the tile clean
#pragma unroll
        for (int i = 0; i < RPT; ++i)
#pragma unroll
            for (int k = 0; k < YD; ++k) Cs[HL_ROW(rg, i) * CLD + v * YD + k] = 0.f;
    }
    HL_CLK(4);
    // ELBO row sums over the 16 variables of the tile (wavefront shuffles inside the 16-lane group)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const float s = row16_sum(lpo[i]);
        const int gr = m0 + HL_ROW(rg, i);
        if (v == 0 && gr < Bp) rowpart[(size_t)tn * Bp + gr] = gr < B ? s : 0.f;
    }
    HL_CLK(5);
    if (!want_grad) { HL_STAMP_END(stamp); return; }
    // d Y^T [NY][Bp] straight from this thread's own cells of the tile (written by the head functions above: no barrier):
    // four consecutive rows of one column = one 8-byte store, and the column sums d by on the way
    if (!conv && d < D) {
#pragma unroll
        for (int k = 0; k < YD; ++k) {
            float cs = 0.f;
#pragma unroll
            for (int q = 0; q < RPT / 4; ++q) {
                const float* src = Cs + HL_ROW(rg, 4 * q) * CLD + v * YD + k;
                const float a0 = src[0], a1 = src[CLD], a2 = src[2 * CLD], a3 = src[3 * CLD];
                uint2 pk;
                pk.x = (uint32_t)f2bf(a0) | ((uint32_t)f2bf(a1) << 16);
                pk.y = (uint32_t)f2bf(a2) | ((uint32_t)f2bf(a3) << 16);
                *reinterpret_cast<uint2*>(dyT + (size_t)(n0 + v * YD + k) * Bp + m0 + HL_ROW(rg, 4 * q)) = pk;
                cs += (a0 + a1) + (a2 + a3);
            }
            acc[NHEAD + k] = cs;
        }
    }
    HL_CLK(6);
    __syncthreads();                       // every wave is done with the scratch and the operand buffers; the dY tile is complete
    HL_CLK(7);
    // head-parameter gradients and d by: every thread's accumulators go to LDS as image[n][thread] (conflict-free), a second
    // pass sums the 16 row groups of each (variable, n) and leaves the tile's partial sums in ws->hgpart [row block][n][variable]
    // for k_head_grad_reduce.  (Before: two permlane swaps per accumulator + 464 global float atomics per workgroup through an
    // arena-offset switch: 2.4 k + 5.0 k of a wave's 43 k clocks.)
    float* img = reinterpret_cast<float*>(smem) + BM * CLD;
#pragma unroll
    for (int n = 0; n < NACC; ++n) img[n * RST + tid] = acc[n];
    HL_CLK(8);
    // dY tile -> HBM, row-major bf16 [Bp][NYp]: one 16-byte store (8 bf16) per task from two 16-byte LDS reads
    constexpr int CCH = BN / 8;                                   // 8-column chunks per tile row
    for (int idx = tid; idx < BM * CCH; idx += HL_THREADS) {
        const int r = idx / CCH, c = (idx % CCH) * 8;
        const float4 lo = *reinterpret_cast<const float4*>(Cs + r * CLD + c);
        const float4 hi = *reinterpret_cast<const float4*>(Cs + r * CLD + c + 4);
        if (n0 + c + 8 <= lddy) {
            uint4 o;
            o.x = (uint32_t)f2bf(lo.x) | ((uint32_t)f2bf(lo.y) << 16);
            o.y = (uint32_t)f2bf(lo.z) | ((uint32_t)f2bf(lo.w) << 16);
            o.z = (uint32_t)f2bf(hi.x) | ((uint32_t)f2bf(hi.y) << 16);
            o.w = (uint32_t)f2bf(hi.z) | ((uint32_t)f2bf(hi.w) << 16);
            *reinterpret_cast<uint4*>(dy + (size_t)(m0 + r) * lddy + n0 + c) = o;
        } else {
            const float e8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            for (int e = 0; e < 8; ++e)
                if (n0 + c + e < lddy) dy[(size_t)(m0 + r) * lddy + n0 + c + e] = f2bf(e8[e]);
        }
    }
    HL_CLK(9);
    __syncthreads();
    HL_CLK(10);
    const int NTV = (int)(gridDim.x / tiles_m) * 16;               // variables, padded to whole tiles
    float* out = hgpart + (size_t)(m0 / BM) * NACC * NTV + tn * 16;
    for (int idx = tid; idx < 16 * NACC; idx += HL_THREADS) {
        const int vv = idx & 15, n = idx >> 4;
        const float* src = img + n * RST + vv;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int q = 0; q < 16; q += 2) { s0 += src[q * 16]; s1 += src[q * 16 + 16]; }
        out[(size_t)n * NTV + vv] = s0 + s1;
    }
    HL_STAMP_END(stamp);
    HL_CLK(11);
#undef HL_CLK
}

// folds the per-row-block partial sums of k_y_heads into the gradient arena: one thread per (accumulator n, variable)
template <int YD, int KMAX>
__global__ __launch_bounds__(256) void k_head_grad_reduce(const float* __restrict__ hgpart, int tiles_m, int NTV, int D,
                                                          const hlvae_var* __restrict__ vars, float* __restrict__ G, long o_by,
                                                          int conv) {
    constexpr int NHEAD = HeadAcc<YD, KMAX>::N, NACC = NHEAD + YD;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int dd = idx % NTV, n = idx / NTV;
    if (n >= NACC || dd >= D) return;
    int dst;
    if (n >= NHEAD) {
        if (conv) return;                                        // y_layer's bias gradient belongs to the convolution kernels there
        dst = (int)o_by + vars[dd].pad * YD + (n - NHEAD);         // `vars` is in the kernel's order, the bias in the variables' own
    } else {
        dst = acc_dest<YD, KMAX>(vars[dd], n);
    }
    if (dst < 0) return;
    const float* src = hgpart + (size_t)n * NTV + dd;
    const size_t st = (size_t)NACC * NTV;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = 0;
    for (; t + 4 <= tiles_m; t += 4) { s0 += src[t * st]; s1 += src[(t + 1) * st]; s2 += src[(t + 2) * st]; s3 += src[(t + 3) * st]; }
    for (; t < tiles_m; ++t) s0 += src[t * st];
    atomicAdd(G + dst, (s0 + s1) + (s2 + s3));                   // the region is zero at the start of a step: other kernels add to
}                                                                // their own parts of it concurrently

int hl_launch_head_grad_reduce(const hlvae_plan* p, const hlvae_ws* ws, int Bp, hipStream_t s) {
    const hlvae_dims& d = p->d;
    const int NT = (d.D + 15) / 16, NTV = NT * 16, tiles_m = Bp / 64;
    HL_PROF("head_grad_reduce", s);
#define HL_RED(YDv, KMv)                                                                                               \
    k_head_grad_reduce<YDv, KMv><<<(NTV * (HeadAcc<YDv, KMv>::N + YDv) + 255) / 256, 256, 0, s>>>(ws->hgpart, tiles_m, NTV, d.D,    \
                                                                                                p->vars_sorted_dev, ws->G, d.o_by, d.conv)
    if (d.y_dim == 3) HL_RED(3, 8);
    else if (d.y_dim == 8) HL_RED(8, 8);
    else if (p->kmax <= 3) HL_RED(5, 3);
    else if (p->kmax <= 5) HL_RED(5, 5);
    else if (p->kmax <= 8) HL_RED(5, 8);
    else HL_RED(5, 16);
#undef HL_RED
    HL_LAUNCH_CHECK();
    return 0;
}

// ELBO bookkeeping in ONE block (no atomics, no memset, deterministic):
//   nll[b]  = -sum_tiles rowpart[t][b]                         (HLVAE.py:377-379)
//   scal[0] = sum_b nll[b]                                     (training.py:104)
//   scal[1] = sum of the per-block KL(q || N(0,I)) partials    (extension)
//   rng[1] += 1: advances the Philox offset of the reparameterisation noise for the next step
struct ElboFinArgs {
    const float* rowpart;
    float* nll;
    double* scal;
    const double* klpart;
    uint64_t* rng;
    int NT, Bp, B, nkl;
};

// one workgroup of `nthr` threads (a multiple of 64, <= 1024)
__device__ __forceinline__ void elbo_finalize_body(const ElboFinArgs& f, int tid, int nthr) {
    __shared__ double red[16];
    __shared__ double redk[16];
    const float* __restrict__ rowpart = f.rowpart;
    const int NT = f.NT, Bp = f.Bp, B = f.B;
    double tot = 0.0;
    for (int b = tid; b < Bp; b += nthr) {
        float s = 0.f;
        if (b < B) {                                    // 27 loads in flight per lane (one workgroup: pure latency; 9 took 8 us)
            float s0 = 0.f, s1 = 0.f, s2 = 0.f;
            int t = 0;
            for (; t + 27 <= NT; t += 27) {
                float v[27];
#pragma unroll
                for (int k = 0; k < 27; ++k) v[k] = rowpart[(size_t)(t + k) * Bp + b];
#pragma unroll
                for (int k = 0; k < 27; k += 3) { s0 += v[k]; s1 += v[k + 1]; s2 += v[k + 2]; }
            }
            for (; t < NT; ++t) s0 += rowpart[(size_t)t * Bp + b];
            s = (s0 + s1) + s2;
        }
        f.nll[b] = -s;
        tot += (double)(-s);
    }
    // the KL partials (one per 4 rows) in parallel too: a serial loop of one thread over them is a chain of global loads
    double kl = 0.0;
    if (f.klpart != nullptr)
        for (int i = tid; i < f.nkl; i += nthr) kl += f.klpart[i];
    tot = wave_sum_d(tot);
    kl = wave_sum_d(kl);
    if ((tid & 63) == 0) { red[tid >> 6] = tot; redk[tid >> 6] = kl; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0, k = 0.0;
        for (int w = 0; w < (nthr >> 6); ++w) { t += red[w]; k += redk[w]; }
        f.scal[0] = t;
        f.scal[1] = k;
        if (f.rng != nullptr) f.rng[1] += 1;
    }
}

__global__ __launch_bounds__(1024) void k_elbo_finalize(ElboFinArgs f) { elbo_finalize_body(f, threadIdx.x, blockDim.x); }

// Row M: per-variable reconstruction errors of the imputed values (reference read_functions.py:342-412 with
// true_miss_mask = 1, conv False): categorical 0/1 mismatch, ordinal |x - x_hat| / K, continuous (x_hat - x)^2 / range^2
// (range = max - min of the batch column, 1 if zero), averaged over observed / missing / all rows, sqrt for continuous.
// The range is a per-column constant, so ONE pass accumulates the un-normalised sums together with max / min:
//   k_metrics_partial  grid (D/64, 16 row chunks): part[chunk][6][D] = {max, min, sum_obs, sum_miss, n_obs, n_miss}
//   k_metrics_finish   combines the 16 partials -> err[3][D] = observed, missing, all
#define HL_MET_CHUNKS 16
__global__ __launch_bounds__(1024) void k_metrics_partial(const float* __restrict__ xt, const uint8_t* __restrict__ m8,
                                                         const float* __restrict__ xhat, const hlvae_var* __restrict__ vars,
                                                         int B, int D, float* __restrict__ part, int conv, ElboFinArgs fin,
                                                         int with_fin) {
    // with_fin: the grid has one more column of workgroups; its first one does the ELBO bookkeeping of the step (k_elbo_finalize's
    // work: both are deferred side work of a training step, and as launches of their own each is a dependent ~7-10 us link of the
    // side chain that the end of the step waits for)
    if (with_fin && blockIdx.x == gridDim.x - 1) {
        if (blockIdx.y == 0) elbo_finalize_body(fin, threadIdx.y * 64 + threadIdx.x, blockDim.x * blockDim.y);
        return;
    }
    // conv (types_info['conv'], read_functions.py:366-369): continuous data are divided by 255 (x_hat too for pos / count)
    // and the range normalisation is dropped
    __shared__ float red[6][16][64];                            // blockDim = (64 variables, RL row lanes), RL = 4 or 16
    const int d = blockIdx.x * 64 + threadIdx.x, g = threadIdx.y, RL = blockDim.y;
    const int rpc = (B + HL_MET_CHUNKS - 1) / HL_MET_CHUNKS;
    const int b_lo = blockIdx.y * rpc, b_hi = min(B, b_lo + rpc);
    int kind = -1, K = 1;
    if (d < D) { kind = vars[d].kind; K = vars[d].ncls; }
    float mx = -3.4e38f, mn = 3.4e38f, so = 0.f, sm = 0.f, no = 0.f, nm = 0.f;
    if (d < D)
      for (int b0 = b_lo + g; b0 < b_hi; b0 += RL * 8) {                      // 8 rows per pass: 24 loads in flight per lane (at 512
        float xv[8], xhv[8];                                                // rows the whole chunk is one pass: the kernel was a
        uint8_t mv[8];                                                      // chain of 8 dependent-latency iterations, 9.5 us)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t o = (size_t)min(b0 + RL * u, b_hi - 1) * D + d;
            xv[u] = xt[o];
            xhv[u] = xhat[o];
            mv[u] = m8[o];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (b0 + RL * u >= b_hi) break;
            float x = xv[u];
            float xh = xhv[u];
            float e;
            if (kind == HLVAE_CAT) {
                e = (fmaxf(x, 0.f) != xh) ? 1.f : 0.f;           // argmax of an all-zero one-hot row is class 0
            } else if (kind == HLVAE_ORDINAL) {
                e = fabsf(x - xh) / (float)K;
            } else {
                if (kind == HLVAE_POS) x = expm1f(x);            // the target buffer holds log1p(x) for pos
                if (conv) {
                    x *= 1.f / 255.f;
                    if (kind != HLVAE_REAL) xh *= 1.f / 255.f;
                }
                mx = fmaxf(mx, x);
                mn = fminf(mn, x);
                e = (xh - x) * (xh - x);
            }
            if (mv[u]) { so += e; no += 1.f; } else { sm += e; nm += 1.f; }
        }
      }
    red[0][g][threadIdx.x] = mx; red[1][g][threadIdx.x] = mn; red[2][g][threadIdx.x] = so;
    red[3][g][threadIdx.x] = sm; red[4][g][threadIdx.x] = no; red[5][g][threadIdx.x] = nm;
    __syncthreads();
    if (g == 0 && d < D) {
        float* out = part + (size_t)blockIdx.y * 6 * D + d;
        float a = red[0][0][threadIdx.x], b = red[1][0][threadIdx.x], t[4] = {red[2][0][threadIdx.x], red[3][0][threadIdx.x],
                                                                             red[4][0][threadIdx.x], red[5][0][threadIdx.x]};
        for (int y = 1; y < RL; ++y) {
            a = fmaxf(a, red[0][y][threadIdx.x]);
            b = fminf(b, red[1][y][threadIdx.x]);
#pragma unroll
            for (int k = 0; k < 4; ++k) t[k] += red[k + 2][y][threadIdx.x];
        }
        out[0] = a;
        out[D] = b;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(size_t)(k + 2) * D] = t[k];
    }
}

__global__ void k_metrics_finish(const float* __restrict__ part, const hlvae_var* __restrict__ vars, int B, int D,
                                 float* __restrict__ err, int conv) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    float mx = -3.4e38f, mn = 3.4e38f, s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < HL_MET_CHUNKS; ++c) {
        const float* p = part + (size_t)c * 6 * D + d;
        mx = fmaxf(mx, p[0]);
        mn = fminf(mn, p[D]);
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] += p[(size_t)(k + 2) * D];
    }
    const int kind = vars[d].kind;
    const bool disc = kind == HLVAE_CAT || kind == HLVAE_ORDINAL;
    float inv2 = 1.f;
    if (!disc && !conv) {
        float nt = mx - mn;
        if (nt == 0.f) nt = 1.f;                                 // read_functions.py:372
        inv2 = 1.f / (nt * nt);
    }
    float eo = s[0] * inv2 / fmaxf(s[2], 1.f), em = s[1] * inv2 / fmaxf(s[3], 1.f), ea = (s[0] + s[1]) * inv2 / fmaxf((float)B, 1.f);
    if (!disc) { eo = sqrtf(eo); em = sqrtf(em); ea = sqrtf(ea); }           // RMSE (read_functions.py:390-393)
    err[d] = eo;
    err[D + d] = em;
    err[2 * D + d] = ea;
}

static ElboFinArgs fin_args(const hlvae_plan* p, const hlvae_ws* ws, int B, int Bp) {
    return ElboFinArgs{ws->rowpart, ws->nll, ws->scal, ws->klpart, ws->rng, (p->d.D + 15) / 16, Bp, B, Bp / 4};   // one KL partial per 4 rows (k_mid_fwd_fused)
}

// fin_ws != nullptr: the ELBO bookkeeping of that workspace's step rides in the first launch
int hl_launch_step_metrics(const hlvae_plan* p, const hlvae_ws* ws, int B, float* err, hipStream_t s, const hlvae_ws* fin_ws, int fin_B) {
    const hlvae_dims& d = p->d;
    HL_REQUIRE(ws->xhat != nullptr && err != nullptr && ws->metpart != nullptr, HLVAE_EINVAL,
               "step_metrics: needs ws->xhat (decoder_fwd with want_params), ws->metpart and err");
    {
        HL_PROF("metrics_partial", s);
        const int with_fin = fin_ws != nullptr;
        const ElboFinArgs fa = with_fin ? fin_args(p, fin_ws, fin_B, (fin_B + 127) / 128 * 128) : ElboFinArgs{};
        k_metrics_partial<<<dim3((d.D + 63) / 64 + with_fin, HL_MET_CHUNKS), dim3(64, B >= 2048 ? 16 : 4), 0, s>>>(
            ws->xt, ws->m8, ws->xhat, p->vars_dev, B, d.D, ws->metpart, d.conv, fa, with_fin);
    }
    HL_LAUNCH_CHECK();
    HL_PROF("metrics_finish", s);
    k_metrics_finish<<<(d.D + 255) / 256, 256, 0, s>>>(ws->metpart, p->vars_dev, B, d.D, err, d.conv);
    HL_LAUNCH_CHECK();
    return 0;
}

// dY *= g[b][d] after the fact (autograd path with a non-uniform upstream gradient)
__global__ void k_scale_dy(bf16_t* __restrict__ dy, int lddy, bf16_t* __restrict__ dyT, int Bp,
                           const float* __restrict__ g, int B, int D, int YD, const hlvae_var* __restrict__ vars) {
    const int NY = D * YD;                                        // (dY's columns follow `vars`, the kernel's variable order; g the variables' own)
    const long n = (long)B * NY;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / NY), c = (int)(i % NY);
        const float s = g[(size_t)b * D + vars[c / YD].pad];
        const size_t o = (size_t)b * lddy + c;
        const bf16_t r = f2bf(bf2f(dy[o]) * s);
        dy[o] = r;
        dyT[(size_t)c * Bp + b] = r;
    }
}

// phase stamps of every wave (tools/heads_phases.py): HL_HEADS_CLK=1 makes the launcher hand the kernel a buffer
// [workgroup][wave][12] of clock64() values; nullptr (no stamps) otherwise
static long long* g_heads_clk = nullptr;
static int g_heads_clk_n = 0;
static long long* hl_heads_clk_buffer(int grid) {
    static const bool on = getenv("HL_HEADS_CLK") != nullptr;
    if (!on) return nullptr;
    if (grid > g_heads_clk_n) {
        if (g_heads_clk) (void)hipFree(g_heads_clk);
        if (hipMalloc(&g_heads_clk, sizeof(long long) * 48 * grid) != hipSuccess) { g_heads_clk = nullptr; g_heads_clk_n = 0; return nullptr; }
        g_heads_clk_n = grid;
    }
    return g_heads_clk;
}
extern "C" int hlvae_debug_heads_clk(long long* host, int max_wg) {
    if (!g_heads_clk) return 0;
    const int n = max_wg < g_heads_clk_n ? max_wg : g_heads_clk_n;
    if (hipMemcpy(host, g_heads_clk, sizeof(long long) * 48 * n, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}

int hl_launch_y_heads(const hlvae_plan* p, const hlvae_ws* ws, const float* g_elem, float g_scale, int want_grad,
                      int want_params, int B, int Bp, hipStream_t s) {
    const hlvae_dims& d = p->d;
    HL_REQUIRE(d.y_dim == 5 || d.y_dim == 3 || d.y_dim == 8, HLVAE_EINVAL,
               "y_dim=%d: instantiated for 5 (the reference configuration), 3 and 8", d.y_dim);
    const int NT = (d.D + 15) / 16;
    float* pf = want_params == 1 ? ws->pfull : nullptr;      // 1: p_params + x_hat, 2: x_hat only (training metrics)
    float* xh = want_params ? ws->xhat : nullptr;
    HL_REQUIRE(!want_params || (ws->xhat && (want_params != 1 || ws->pfull)), HLVAE_EINVAL, "want_params without pfull/xhat buffers");
    HL_REQUIRE(!want_grad || ws->hgpart, HLVAE_EINVAL, "want_grad without the ws->hgpart buffer");
    {
        HL_PROF("y_heads_loglik", s);
        // 64-row tiles at every batch size.  A 128-row instance (half the Wy panel re-reads, 2 workgroups per CU) was kept for
        // >= 512 workgroups in round 1; measured on MI355X at 4096 rows it is no faster (173.3 vs 171.7 us; 33.7 vs 29.6 us at
        // 512 rows).  Stand-alone the GEMM part takes 12 us and the epilogue 15 us of the kernel's 23 (512 rows): they barely
        // overlap, because the co-resident workgroups of a CU run in phase; PMC: VALU busy 42 %, 49 % of the wave-cycles waiting
        const int grid = NT * (Bp / 64);
        long long* clk = hl_heads_clk_buffer(grid);
#define HL_LAUNCH_HEADS(KMv) HL_LAUNCH_HEADS_Y(5, 64, KMv)
#define HL_LAUNCH_HEADS_Y(YDv, BMv, KMv) HL_LAUNCH_HEADS_V(YDv, BMv, KMv, 1)
#define HL_LAUNCH_HEADS_V(YDv, BMv, KMv, DMAv)                                                                         \
        k_y_heads<YDv, BMv, KMv, DMAv><<<grid, HL_THREADS, 0, s>>>(ws->u, d.hdp, ws->wys, d.hdp, p->vars_sorted_dev, ws->P, ws->hgpart, d.o_by,  \
                                                          ws->norm, d.n_stat, ws->xt, ws->m8, d.D, g_elem, g_scale, ws->dy, \
                                                          d.NYp, ws->dyT, Bp, ws->log_p_x, ws->log_p_x_missing,       \
                                                          ws->rowpart, pf, d.Theta, xh, B, want_grad, d.conv ? ws->yv : nullptr, d.NY, clk, d.Theta != d.X, hl_stamp_slot(HL_ST_HEADS))
        if (d.y_dim == 3) HL_LAUNCH_HEADS_Y(3, 64, 8);          // other y_dim (config/hlvae_config_file.txt: y_dim): all class counts up to 8
        else if (d.y_dim == 8) HL_LAUNCH_HEADS_Y(8, 64, 8);
        else if (p->kmax <= 3) { if (d.hdp % 256 == 0) HL_LAUNCH_HEADS_V(5, 64, 3, 2); else HL_LAUNCH_HEADS_V(5, 64, 3, 1); }
        else if (p->kmax <= 5) {          // (the D4 / tabular instance keeps the older cores too, for A/B runs on one box:
            static const int core_env = [] {  //  HL_GEMM_CORE=nt -> register-staged, HL_HEADS_CORE=1 -> [A; B] through LDS-DMA)
                const char* e = getenv("HL_GEMM_CORE");
                if (e != nullptr && e[0] == 'n') return 0;
                e = getenv("HL_HEADS_CORE");
                return (e != nullptr && e[0] == '1') ? 1 : 2;
            }();
            const int core = (core_env == 2 && d.hdp % 256 != 0) ? 1 : core_env;  // CORE 2: whole groups of four k-tiles
            if (core == 0) HL_LAUNCH_HEADS_V(5, 64, 5, 0); else if (core == 1) HL_LAUNCH_HEADS_V(5, 64, 5, 1); else HL_LAUNCH_HEADS_V(5, 64, 5, 2);
        }
        else if (p->kmax <= 8) HL_LAUNCH_HEADS(8);
        else HL_LAUNCH_HEADS(16);                                // 9..16 classes: one instance
#undef HL_LAUNCH_HEADS
#undef HL_LAUNCH_HEADS_Y
#undef HL_LAUNCH_HEADS_V
    }
    HL_LAUNCH_CHECK();
    return 0;
}

// ELBO bookkeeping (scalars, noise-stream advance); nothing of the backward pass depends on it
int hl_launch_elbo_finalize(const hlvae_plan* p, const hlvae_ws* ws, int B, int Bp, hipStream_t s) {
    const int NT = (p->d.D + 15) / 16;
    HL_PROF("elbo_finalize", s);
    k_elbo_finalize<<<1, 1024, 0, s>>>(fin_args(p, ws, B, Bp));
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_scale_dy(const hlvae_plan* p, const hlvae_ws* ws, const float* g, int B, int Bp, hipStream_t s) {
    const hlvae_dims& d = p->d;
    HL_PROF("scale_dy", s);
    k_scale_dy<<<1024, 256, 0, s>>>(ws->dy, d.NYp, ws->dyT, Bp, g, B, d.D, d.y_dim, p->vars_sorted_dev);
    HL_LAUNCH_CHECK();
    return 0;
}
