// Row A: masked per-batch normalisation (reference HL_VAE/utils.py:88-143) fused with the packing of the
// reference's fp64 inputs into what the rest of the step reads:
//   xn / xnT  bf16 normalised encoder input in both layouts (one-hot / thermometer columns are exact in bf16)
//   xt        fp32 likelihood target per (row, variable): raw x (real, count), log1p x (pos), class index
//   m8        uint8 observation mask
// The 93 kB/row of fp64 (data + mask + param_mask) is read exactly once per step.
#include "common.h"

// ---- pass 1: masked column sums, fp64, NO atomics and no memset: chunk c of the rows writes its partial
// sums to sums[c][0..2][col]; pass 2 adds the HL_STAT_CHUNKS partials.  (Data parallel: the whole partial
// buffer is all-reduced between the passes -- sums are linear.)
__global__ __launch_bounds__(256) void k_colstats(const double* __restrict__ data, const double* __restrict__ mask,
                                                  const hlvae_var* __restrict__ vars, const int32_t* __restrict__ stat_var,
                                                  int n_stat, int X, int D, int B, double* __restrict__ sums) {
    __shared__ double red[3][4][64];
    const int sc = blockIdx.x * 64 + threadIdx.x;
    const int rpc = (B + HL_STAT_CHUNKS - 1) / HL_STAT_CHUNKS;
    const int b_lo = blockIdx.y * rpc, b_hi = min(B, b_lo + rpc);
    double s0 = 0, s1 = 0, s2 = 0;
    if (sc < n_stat) {
        const int d = stat_var[sc];
        const hlvae_var var = vars[d];
        for (int b = b_lo + threadIdx.y; b < b_hi; b += 4) {
            const double m = mask[(size_t)b * D + d];
            double x = data[(size_t)b * X + var.xoff] * m;       // observed_data = d * m  (utils.py:98,124)
            if (var.kind == HLVAE_POS) x = log1p(x);             // :125
            s0 += m;
            s1 += x * m;                                         // :105,126
            s2 += x * x * m;
        }
    }
    red[0][threadIdx.y][threadIdx.x] = s0;
    red[1][threadIdx.y][threadIdx.x] = s1;
    red[2][threadIdx.y][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.y == 0 && sc < n_stat) {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            sums[((size_t)blockIdx.y * 3 + k) * n_stat + sc] =
                red[k][0][threadIdx.x] + red[k][1][threadIdx.x] + red[k][2][threadIdx.x] + red[k][3][threadIdx.x];
    }
}

// ---- pass 2: normalise + pack, 64 rows x 64 expanded columns per block --------------------------
// thread = one column of the tile (its type, statistics and variable are per-thread constants) x 4 rows;
// 1024 threads per block keep enough 8-byte loads in flight for a pure streaming kernel
#define HL_PACK_THREADS 1024
__global__ __launch_bounds__(HL_PACK_THREADS) void k_normalize_pack(
    const double* __restrict__ data, const double* __restrict__ mask, const hlvae_var* __restrict__ vars,
    const int32_t* __restrict__ col2var, const double* __restrict__ sums, float* __restrict__ norm, int n_stat, int X,
    int Xp, int D, int B, int Bp, bf16_t* __restrict__ xn, bf16_t* __restrict__ xnT, float* __restrict__ xt,
    uint8_t* __restrict__ m8) {
    constexpr int T = 64, CLD = T + 1;
    __shared__ float tile[T * CLD];
    const int x0 = blockIdx.x * T, b0 = blockIdx.y * T;
    const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int x = x0 + c;
    int kind = -1, d = 0, K = 1;
    bool first = false;
    __shared__ float s_mean[T], s_rstd[T];
    int sidx = -1;
    if (x < X) {
        d = col2var[x];
        const hlvae_var var = vars[d];
        kind = var.kind;
        K = var.ncls;
        first = (x == var.xoff);
        sidx = var.sidx;
    }
    if (rq == 0) {                      // one wave finishes the statistics of the tile's 64 columns
        float mean_c = 0.f, rstd_c = 1.f;
        if (kind == HLVAE_REAL || kind == HLVAE_POS) {
            double n = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int ch = 0; ch < HL_STAT_CHUNKS; ++ch) {          // 48 independent loads in flight
                n += sums[((size_t)ch * 3 + 0) * n_stat + sidx];
                s1 += sums[((size_t)ch * 3 + 1) * n_stat + sidx];
                s2 += sums[((size_t)ch * 3 + 2) * n_stat + sidx];
            }
            const double mu = s1 / n;                                        // utils.py:105,126
            double vv = (s2 - 2.0 * mu * s1 + mu * mu * n) / n;              // :106,127
            if (vv < 0.0) vv = 0.0;
            if (kind == HLVAE_POS) vv = fmin(fmax(vv, 1e-6), 1e20);          // :128
            mean_c = (float)mu;
            rstd_c = (float)(1.0 / sqrt(vv + 1e-5));                         // :107,129
            if (blockIdx.y == 0) {
                norm[sidx] = mean_c;
                norm[n_stat + sidx] = (float)vv;
            }
        }
        s_mean[c] = mean_c;
        s_rstd[c] = rstd_c;
    }
    __syncthreads();
    const float mean = s_mean[c], rstd = s_rstd[c];
    // Unconditional loads from clamped rows (no branches around the loads), 32-bit offsets from two base pointers.
    const bool has_col = kind >= 0;
    const double* dcol = data + (has_col ? x : 0);
    const double* mcol = mask + d;
    const bool logk = (kind == HLVAE_POS) || (kind == HLVAE_COUNT);
    const bool disc = (kind == HLVAE_CAT) || (kind == HLVAE_ORDINAL);
    const bool fits = c + K <= T;                      // the variable's K columns sit inside this 64-lane tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rq + 16 * i, b = b0 + r;
        const int bc = min(b, B - 1);
        const float raw = (float)dcol[(size_t)bc * X];
        const bool ob = (mcol[(size_t)bc * D] != 0.0) && (b < B) && has_col;
        float fr = raw;
        if (logk) fr = (kind == HLVAE_POS) ? log1pf(raw) : __logf(raw);       // utils.py:125 / :118
        tile[r * CLD + c] = ob ? (fr - mean) * rstd : 0.f;                    // mean = 0, rstd = 1 unless real / pos
        // likelihood target of the variable: the first lane of a cat / ordinal variable collects the other K-1
        // columns from its neighbour lanes (shuffles are executed by every lane of the wave)
        float nb[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) nb[k] = __shfl_down(raw, k + 1, 64);
        if (first && b < B) {
            float tv = logk && kind == HLVAE_POS ? fr : raw;                  // real, count: raw x; pos: log1p x
            if (disc) {
                float v[16];                              // K <= 16; columns past the 8th come straight from memory
                v[0] = raw;
#pragma unroll
                for (int k = 1; k < 8; ++k)
                    v[k] = (k < K) ? (fits ? nb[k - 1] : (float)dcol[(size_t)bc * X + k]) : 0.f;
#pragma unroll
                for (int k = 8; k < 16; ++k) v[k] = (k < K) ? (float)dcol[(size_t)bc * X + k] : 0.f;
                if (kind == HLVAE_CAT) {                  // one-hot -> class index, -1 if the row is all zero
                    int cls = -1;
                    float best = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < K && v[k] > best) { best = v[k]; cls = k; }
                    tv = (float)cls;
                } else {                                  // thermometer -> sum(int(data)) - 1 (loglik.py:172)
                    int sum = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k)
                        if (k < K) sum += (int)v[k];
                    tv = (float)(sum - 1);
                }
            }
            xt[(size_t)b * D + d] = tv;
            m8[(size_t)b * D + d] = ob ? 1 : 0;
        }
    }
    __syncthreads();
    // row-major: 2 columns per lane (one 32-bit store), transposed: 2 rows per lane
    for (int idx = threadIdx.x; idx < T * T / 2; idx += HL_PACK_THREADS) {
        const int r = idx / (T / 2), c2 = (idx % (T / 2)) * 2;
        if (b0 + r < Bp && x0 + c2 < Xp) {
            const uint32_t v = (uint32_t)f2bf(tile[r * CLD + c2]) | ((uint32_t)f2bf(tile[r * CLD + c2 + 1]) << 16);
            *reinterpret_cast<uint32_t*>(xn + (size_t)(b0 + r) * Xp + x0 + c2) = v;
        }
    }
    for (int idx = threadIdx.x; idx < T * T / 2; idx += HL_PACK_THREADS) {
        const int cc = idx / (T / 2), r2 = (idx % (T / 2)) * 2;
        if (b0 + r2 < Bp && x0 + cc < Xp) {
            const uint32_t v = (uint32_t)f2bf(tile[r2 * CLD + cc]) | ((uint32_t)f2bf(tile[(r2 + 1) * CLD + cc]) << 16);
            *reinterpret_cast<uint32_t*>(xnT + (size_t)(x0 + cc) * Bp + b0 + r2) = v;
        }
    }
}

int hl_launch_stats(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B, hipStream_t s) {
    const hlvae_dims& d = p->d;
    if (d.n_stat == 0) return 0;
    dim3 block(64, 4);
    dim3 grid((d.n_stat + 63) / 64, HL_STAT_CHUNKS);
    HL_PROF("colstats", s);
    k_colstats<<<grid, block, 0, s>>>(data, mask, p->vars_dev, p->stat_var_dev, d.n_stat, d.X, d.D, B, ws->sums);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_pack(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B, int Bp,
                   hipStream_t s) {
    const hlvae_dims& d = p->d;
    dim3 grid(d.Xp / 64, Bp / 64);
    HL_PROF("normalize_pack", s);
    k_normalize_pack<<<grid, HL_PACK_THREADS, 0, s>>>(data, mask, p->vars_dev, p->col2var_dev, ws->sums, ws->norm, d.n_stat,
                                                 d.X, d.Xp, d.D, B, Bp, ws->xn, ws->xnT, ws->xt, ws->m8);
    HL_LAUNCH_CHECK();
    return 0;
}
