// Row A: masked per-batch normalisation (reference HL_VAE/utils.py:88-143) fused with the packing of the
// reference's fp64 inputs into what the rest of the step reads:
//   xn / xnT  bf16 normalised encoder input in both layouts (one-hot / thermometer columns are exact in bf16)
//   xt        fp32 likelihood target per (row, variable): raw x (real, count), log1p x (pos), class index
//   m8        uint8 observation mask
// The 93 kB/row of fp64 (data + mask + param_mask) is read exactly once per step.
#include "common.h"

// ---- pass 1: masked column sums, fp64 accumulate ------------------------------------------------
// one lane per statistics column, rows strided over blockIdx.y/threadIdx.y, fp64 atomics to combine
__global__ void k_colstats(const double* __restrict__ data, const double* __restrict__ mask,
                           const hlvae_var* __restrict__ vars, const int32_t* __restrict__ stat_var, int n_stat, int X,
                           int D, int B, double* __restrict__ sums) {
    const int sc = blockIdx.x * 64 + threadIdx.x;
    if (sc >= n_stat) return;
    const int d = stat_var[sc];
    const hlvae_var var = vars[d];
    double s0 = 0, s1 = 0, s2 = 0;
    for (int b = blockIdx.y * blockDim.y + threadIdx.y; b < B; b += gridDim.y * blockDim.y) {
        const double m = mask[(size_t)b * D + d];
        double x = data[(size_t)b * X + var.xoff] * m;           // observed_data = d * m  (utils.py:98,124)
        if (var.kind == HLVAE_POS) x = log1p(x);                 // :125
        s0 += m;
        s1 += x * m;                                             // :105,126
        s2 += x * x * m;
    }
    atomicAdd(sums + sc, s0);
    atomicAdd(sums + n_stat + sc, s1);
    atomicAdd(sums + 2 * n_stat + sc, s2);
}

// mean / var from the sums:  var = sum((x - mean)^2 m) / sum m  (utils.py:106,127), pos var clamped (:128)
__global__ void k_finish_stats(const double* __restrict__ sums, const hlvae_var* __restrict__ vars,
                               const int32_t* __restrict__ stat_var, int n_stat, float* __restrict__ norm) {
    const int sc = blockIdx.x * blockDim.x + threadIdx.x;
    if (sc >= n_stat) return;
    const double n = sums[sc], s1 = sums[n_stat + sc], s2 = sums[2 * n_stat + sc];
    const double mean = s1 / n;
    double var = (s2 - 2.0 * mean * s1 + mean * mean * n) / n;
    if (var < 0.0) var = 0.0;
    if (vars[stat_var[sc]].kind == HLVAE_POS) var = fmin(fmax(var, 1e-6), 1e20);
    norm[sc] = (float)mean;
    norm[n_stat + sc] = (float)var;
}

// ---- pass 2: normalise + pack, 64 rows x 64 expanded columns per block --------------------------
__global__ __launch_bounds__(HL_THREADS) void k_normalize_pack(
    const double* __restrict__ data, const double* __restrict__ mask, const hlvae_var* __restrict__ vars,
    const int32_t* __restrict__ col2var, const float* __restrict__ norm, int n_stat, int X, int Xp, int D, int B, int Bp,
    bf16_t* __restrict__ xn, bf16_t* __restrict__ xnT, float* __restrict__ xt, uint8_t* __restrict__ m8) {
    constexpr int T = 64, CLD = T + 1;
    __shared__ float tile[T * CLD];
    const int x0 = blockIdx.x * T, b0 = blockIdx.y * T;
    for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
        const int r = idx / T, c = idx % T;
        const int b = b0 + r, x = x0 + c;
        float out = 0.f;
        if (b < B && x < X) {
            const int d = col2var[x];
            const hlvae_var var = vars[d];
            const double m = mask[(size_t)b * D + d];
            const double raw = data[(size_t)b * X + x];
            const bool ob = m != 0.0;
            switch (var.kind) {
                case HLVAE_REAL: {
                    const float mean = norm[var.sidx], vv = norm[n_stat + var.sidx];
                    out = ob ? (float)((raw - (double)mean) / sqrt((double)vv + 1e-5)) : 0.f;   // utils.py:107
                    break;
                }
                case HLVAE_POS: {
                    const float mean = norm[var.sidx], vv = norm[n_stat + var.sidx];
                    out = ob ? (float)((log1p(raw) - (double)mean) / sqrt((double)vv + 1e-5)) : 0.f;   // :129
                    break;
                }
                case HLVAE_COUNT:
                    out = ob ? (float)log(raw) : 0.f;            // :116-121
                    break;
                default:
                    out = ob ? (float)raw : 0.f;                 // cat / ordinal: d * mask (:133-139)
            }
            if (x == var.xoff) {                                 // first column of the variable: target + mask
                float tv;
                if (var.kind == HLVAE_REAL || var.kind == HLVAE_COUNT) {
                    tv = (float)raw;
                } else if (var.kind == HLVAE_POS) {
                    tv = (float)log1p(raw);                      // loglik.py:84
                } else if (var.kind == HLVAE_CAT) {              // one-hot -> class index, -1 if the row is all zero
                    int cls = -1;
                    double best = 0.0;
                    for (int k = 0; k < var.ncls; ++k) {
                        const double v = data[(size_t)b * X + x + k];
                        if (v > best) { best = v; cls = k; }
                    }
                    tv = (float)cls;
                } else {                                         // thermometer -> sum(int(data)) - 1 (loglik.py:172)
                    int sum = 0;
                    for (int k = 0; k < var.ncls; ++k) sum += (int)data[(size_t)b * X + x + k];
                    tv = (float)(sum - 1);
                }
                xt[(size_t)b * D + d] = tv;
                m8[(size_t)b * D + d] = ob ? 1 : 0;
            }
        }
        tile[r * CLD + c] = out;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
        const int r = idx / T, c = idx % T;
        if (b0 + r < Bp && x0 + c < Xp) xn[(size_t)(b0 + r) * Xp + x0 + c] = f2bf(tile[r * CLD + c]);
    }
    for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
        const int c = idx / T, r = idx % T;
        if (b0 + r < Bp && x0 + c < Xp) xnT[(size_t)(x0 + c) * Bp + b0 + r] = f2bf(tile[r * CLD + c]);
    }
}

int hl_launch_stats(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B, hipStream_t s) {
    const hlvae_dims& d = p->d;
    if (d.n_stat == 0) return 0;
    HL_CHECK(hipMemsetAsync(ws->sums, 0, sizeof(double) * 3 * d.n_stat, s));
    dim3 block(64, 4);
    int gy = (B + 4 * 16 - 1) / (4 * 16);
    if (gy > 64) gy = 64;
    dim3 grid((d.n_stat + 63) / 64, gy);
    HL_PROF("colstats", s);
    k_colstats<<<grid, block, 0, s>>>(data, mask, p->vars_dev, p->stat_var_dev, d.n_stat, d.X, d.D, B, ws->sums);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_pack(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B, int Bp,
                   hipStream_t s) {
    const hlvae_dims& d = p->d;
    if (d.n_stat > 0) {
        HL_PROF("finish_stats", s);
        k_finish_stats<<<(d.n_stat + 255) / 256, 256, 0, s>>>(ws->sums, p->vars_dev, p->stat_var_dev, d.n_stat, ws->norm);
        HL_LAUNCH_CHECK();
    }
    dim3 grid(d.Xp / 64, Bp / 64);
    HL_PROF("normalize_pack", s);
    k_normalize_pack<<<grid, HL_THREADS, 0, s>>>(data, mask, p->vars_dev, p->col2var_dev, ws->norm, d.n_stat, d.X, d.Xp,
                                                 d.D, B, Bp, ws->xn, ws->xnT, ws->xt, ws->m8);
    HL_LAUNCH_CHECK();
    return 0;
}
