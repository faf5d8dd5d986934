// Compact device-resident data feed.  The reference keeps the expanded fp64 matrices in
// pandas frames and builds every batch row by row on the host (dataset_def.py:67-92: 0.25 s per 400 rows), then ships
// 93 kB/row of fp64 (data + mask + param_mask).  Here the whole dataset lives in HBM as
//     values fp32 [N][D]   real / pos / count: the raw value;  cat: class index (-1 = none);  ordinal: level - 1
//     mask   u8   [N][D]   1 = observed
// (5 bytes per entry), a batch is a vector of row indices, and the input stage gathers straight from it: same outputs as
// k_colstats / k_normalize_pack on the expanded fp64 form (one-hot / thermometer columns are regenerated on the fly).
#include "common.h"

__global__ __launch_bounds__(1024) void k_colstats_compact(const float* __restrict__ vals, const uint8_t* __restrict__ mk,
                                                          const int32_t* __restrict__ rows, const hlvae_var* __restrict__ vars,
                                                          const int32_t* __restrict__ stat_var, int n_stat, int D, int B,
                                                          double* __restrict__ sums) {
    // blockDim = (64 statistic columns, RL row lanes): RL = 4, or 16 for chunks of >= 128 rows (batches of >= 2048 rows: with 4
    // lanes a 64-feature model's whole input stage was 16 workgroups walking 256 rows each, 28 us)
    __shared__ double red[3][16][64];
    const int RL = blockDim.y;
    const int sc = blockIdx.x * 64 + threadIdx.x;
    const int rpc = (B + HL_STAT_CHUNKS - 1) / HL_STAT_CHUNKS;
    const int b_lo = blockIdx.y * rpc, b_hi = min(B, b_lo + rpc);
    double s0 = 0, s1 = 0, s2 = 0;
    if (sc < n_stat) {
        const int d = stat_var[sc];
        const int kind = vars[d].kind;
        // two dependent gathers per entry (row index, then value + mask): 8 entries in flight per lane -- at 512 rows the
        // kernel is one such round, pure latency
        constexpr int U = 8;
        for (int bb = b_lo + threadIdx.y; bb < b_hi; bb += RL * U) {
            int rr[U];
            float xv[U];
            uint8_t mm[U];
#pragma unroll
            for (int u = 0; u < U; ++u) rr[u] = bb + RL * u < b_hi ? rows[bb + RL * u] : -1;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t o = (size_t)max(rr[u], 0) * D + d;
                xv[u] = vals[o];
                mm[u] = mk[o];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double m = (rr[u] >= 0 && mm[u]) ? 1.0 : 0.0;
                double x = (double)xv[u] * m;                    // observed_data = d * m  (utils.py:98,124)
                if (kind == HLVAE_POS) x = log1p(x);             // :125
                s0 += m;
                s1 += x * m;
                s2 += x * x * m;
            }
        }
    }
    red[0][threadIdx.y][threadIdx.x] = s0;
    red[1][threadIdx.y][threadIdx.x] = s1;
    red[2][threadIdx.y][threadIdx.x] = s2;
    __syncthreads();
    if (threadIdx.y == 0 && sc < n_stat) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double t = 0.0;
            for (int y = 0; y < RL; ++y) t += red[k][y][threadIdx.x];
            sums[((size_t)blockIdx.y * 3 + k) * n_stat + sc] = t;
        }
    }
}

// 64 batch rows x 64 expanded columns per block, thread = one expanded column x 4 rows (as k_normalize_pack)
#define HL_FEED_THREADS 1024
__global__ __launch_bounds__(HL_FEED_THREADS) void k_pack_compact(
    const float* __restrict__ vals, const uint8_t* __restrict__ mk, const int32_t* __restrict__ rows,
    const hlvae_var* __restrict__ vars, const int32_t* __restrict__ col2var, const double* __restrict__ sums,
    float* __restrict__ norm, int n_stat, int X, int Xp, int D, int B, int Bp, bf16_t* __restrict__ xn,
    bf16_t* __restrict__ xnT, float* __restrict__ xt, uint8_t* __restrict__ m8) {
    constexpr int T = 64, CLD = T + 1;
    __shared__ float tile[T * CLD];
    __shared__ float s_mean[T], s_rstd[T];
    __shared__ int s_row[T];
    const int x0 = blockIdx.x * T, b0 = blockIdx.y * T;
    const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int x = x0 + c;
    int kind = -1, d = 0, koff = 0, sidx = -1;
    if (x < X) {
        d = col2var[x];
        const hlvae_var var = vars[d];
        kind = var.kind;
        koff = x - var.xoff;
        sidx = var.sidx;
    }
    if (threadIdx.x < T) s_row[threadIdx.x] = b0 + threadIdx.x < B ? rows[b0 + threadIdx.x] : -1;
    if (rq == 0) {                      // one wave finishes the statistics of the tile's 64 columns
        float mean_c = 0.f, rstd_c = 1.f;
        if (kind == HLVAE_REAL || kind == HLVAE_POS) {
            double n = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int ch = 0; ch < HL_STAT_CHUNKS; ++ch) {
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
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rq + 16 * i, b = b0 + r, row = s_row[r];
        float out = 0.f;
        if (row >= 0 && kind >= 0) {
            const size_t o = (size_t)row * D + d;
            const float v = vals[o];
            const bool ob = mk[o] != 0;
            float tv = v;
            if (kind == HLVAE_CAT) out = ob && (koff == (int)v) ? 1.f : 0.f;              // one-hot column (read_functions.py:67-82)
            else if (kind == HLVAE_ORDINAL) out = ob && (koff <= (int)v) ? 1.f : 0.f;     // thermometer column (:84-100)
            else {
                float fr = v;
                if (kind == HLVAE_POS) fr = tv = log1pf(v);                              // utils.py:125
                else if (kind == HLVAE_COUNT) fr = __logf(v);                            // :118
                out = ob ? (fr - mean) * rstd : 0.f;
            }
            if (koff == 0) {
                xt[(size_t)b * D + d] = tv;
                m8[(size_t)b * D + d] = ob ? 1 : 0;
            }
        }
        tile[r * CLD + c] = out;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T / 2; idx += HL_FEED_THREADS) {
        const int r = idx / (T / 2), c2 = (idx % (T / 2)) * 2;
        if (b0 + r < Bp && x0 + c2 < Xp) {
            const uint32_t v = (uint32_t)f2bf(tile[r * CLD + c2]) | ((uint32_t)f2bf(tile[r * CLD + c2 + 1]) << 16);
            *reinterpret_cast<uint32_t*>(xn + (size_t)(b0 + r) * Xp + x0 + c2) = v;
        }
    }
    for (int idx = threadIdx.x; idx < T * T / 2; idx += HL_FEED_THREADS) {
        const int cc = idx / (T / 2), r2 = (idx % (T / 2)) * 2;
        if (b0 + r2 < Bp && x0 + cc < Xp) {
            const uint32_t v = (uint32_t)f2bf(tile[r2 * CLD + cc]) | ((uint32_t)f2bf(tile[(r2 + 1) * CLD + cc]) << 16);
            *reinterpret_cast<uint32_t*>(xnT + (size_t)(x0 + cc) * Bp + b0 + r2) = v;
        }
    }
}

int hl_launch_stats_compact(const hlvae_plan* p, const hlvae_ws* ws, const float* vals, const uint8_t* mk, const int32_t* rows,
                            int B, hipStream_t s) {
    const hlvae_dims& d = p->d;
    if (d.n_stat == 0) return 0;
    HL_PROF("colstats", s);
    k_colstats_compact<<<dim3((d.n_stat + 63) / 64, HL_STAT_CHUNKS), dim3(64, B >= 2048 ? 16 : 4), 0, s>>>(vals, mk, rows, p->vars_dev, p->stat_var_dev,
                                                                                          d.n_stat, d.D, B, ws->sums);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_pack_compact(const hlvae_plan* p, const hlvae_ws* ws, const float* vals, const uint8_t* mk, const int32_t* rows,
                           int B, int Bp, hipStream_t s) {
    const hlvae_dims& d = p->d;
    HL_PROF("normalize_pack", s);
    k_pack_compact<<<dim3(d.Xp / 64, Bp / 64), HL_FEED_THREADS, 0, s>>>(vals, mk, rows, p->vars_dev, p->col2var_dev, ws->sums,
                                                                       ws->norm, d.n_stat, d.X, d.Xp, d.D, B, Bp, ws->xn, ws->xnT,
                                                                       ws->xt, ws->m8);
    HL_LAUNCH_CHECK();
    return 0;
}
