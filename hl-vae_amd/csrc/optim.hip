// Adam on the flat fp32 arena (torch.optim.Adam semantics, reference HLVAE_main.py:277-278) and the
// bf16 shadow copies of the dense weights that the MFMA kernels read.  HBM-bound: per parameter
// 4 B grad + 3 x (4 B read + 4 B write) state, float4 per lane, grid-stride.
#include "common.h"

__global__ __launch_bounds__(HL_THREADS) void k_adam(float* __restrict__ P, const float* __restrict__ G,
                                                     float* __restrict__ M1, float* __restrict__ M2, long n4,
                                                     const int64_t* __restrict__ step_count, float lr, float b1,
                                                     float b2, float eps, float gscale) {
    const float t = (float)(*step_count + 1);
    const float bc1 = 1.f - powf(b1, t);
    const float bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1;
    const float rs_bc2 = rsqrtf(bc2);
    float4* P4 = reinterpret_cast<float4*>(P);
    const float4* G4 = reinterpret_cast<const float4*>(G);
    float4* M14 = reinterpret_cast<float4*>(M1);
    float4* M24 = reinterpret_cast<float4*>(M2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 p = P4[i], g = G4[i], m = M14[i], v = M24[i];
        float* pp = &p.x; float* gg = &g.x; float* mm = &m.x; float* vv = &v.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            pp[k] -= step_size * mm[k] / (sqrtf(vv[k]) * rs_bc2 + eps);
        }
        P4[i] = p;
        M14[i] = m;
        M24[i] = v;
    }
}

__global__ void k_inc_step(int64_t* step_count) { *step_count += 1; }

// fp32 [R][C] (dense, row stride C) -> bf16 [Rp][Cp] (zero padded) and optionally its transpose [Cp2][Rp2].
// 64 x 64 tiles through LDS so both writes are coalesced.
__global__ __launch_bounds__(HL_THREADS) void k_shadow(const float* __restrict__ src, int R, int C, bf16_t* __restrict__ dst,
                                                       int ldd, int row_off, bf16_t* __restrict__ dstT, int ldT,
                                                       int Rcover, int Ccover) {
    constexpr int T = 64, CLD = T + 1;
    __shared__ float tile[T * CLD];
    const int c0 = blockIdx.x * T, r0 = blockIdx.y * T;
    for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
        const int r = idx / T, c = idx % T;
        float v = 0.f;
        if (r0 + r < R && c0 + c < C) v = src[(size_t)(r0 + r) * C + c0 + c];
        tile[r * CLD + c] = v;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
        const int r = idx / T, c = idx % T;
        if (r0 + r < Rcover && c0 + c < Ccover) dst[(size_t)(row_off + r0 + r) * ldd + c0 + c] = f2bf(tile[r * CLD + c]);
    }
    if (dstT != nullptr) {
        for (int idx = threadIdx.x; idx < T * T; idx += HL_THREADS) {
            const int c = idx / T, r = idx % T;
            if (r0 + r < Rcover && c0 + c < Ccover)
                dstT[(size_t)(c0 + c) * ldT + row_off + r0 + r] = f2bf(tile[r * CLD + c]);
        }
    }
}

static int shadow(const float* src, int R, int C, bf16_t* dst, int ldd, int row_off, bf16_t* dstT, int ldT, int Rcover,
                  int Ccover, hipStream_t s) {
    dim3 grid((Ccover + 63) / 64, (Rcover + 63) / 64);
    HL_PROF("shadow_cast", s);
    k_shadow<<<grid, HL_THREADS, 0, s>>>(src, R, C, dst, ldd, row_off, dstT, ldT, Rcover, Ccover);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_refresh_shadows(const hlvae_plan* p, const hlvae_ws* ws, hipStream_t s) {
    const hlvae_dims& d = p->d;
    int rc;
    // W1 [h_e][X] -> w1s [hep][Xp]
    if ((rc = shadow(ws->P + d.o_w1, d.h_e, d.X, ws->w1s, d.Xp, 0, nullptr, 0, d.hep, d.Xp, s))) return rc;
    // [Wmu; Wlv] [L][h_e] each -> wmls [2Lp][hep] (+ transpose [hep][2Lp])
    if ((rc = shadow(ws->P + d.o_wmu, d.L, d.h_e, ws->wmls, d.hep, 0, ws->wmlTs, 2 * d.Lp, d.Lp, d.hep, s))) return rc;
    if ((rc = shadow(ws->P + d.o_wlv, d.L, d.h_e, ws->wmls, d.hep, d.Lp, ws->wmlTs, 2 * d.Lp, d.Lp, d.hep, s))) return rc;
    // Wd [h_d][L] -> wds [hdp][Lp] (+ [Lp][hdp])
    if ((rc = shadow(ws->P + d.o_wd, d.h_d, d.L, ws->wds, d.Lp, 0, ws->wdTs, d.hdp, d.hdp, d.Lp, s))) return rc;
    // Wy [NY][h_d] -> wys [NY][hdp] (+ [hdp][NYp])
    if ((rc = shadow(ws->P + d.o_wy, d.NY, d.h_d, ws->wys, d.hdp, 0, ws->wyTs, d.NYp, d.NY, d.hdp, s))) return rc;
    return 0;
}

int hl_adam(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr, float b1,
            float b2, float eps, float gscale, hipStream_t s) {
    const hlvae_dims& d = p->d;
    HL_REQUIRE(d.arena_size % 4 == 0, HLVAE_ESHAPE, "arena size %ld not a multiple of 4", (long)d.arena_size);
    const long n4 = d.arena_size / 4;
    int blocks = (int)((n4 + HL_THREADS - 1) / HL_THREADS);
    if (blocks > 2048) blocks = 2048;
    {
    HL_PROF("adam", s);
    k_adam<<<blocks, HL_THREADS, 0, s>>>(ws->P, ws->G, m1, m2, n4, step_count, lr, b1, b2, eps, gscale);
    }
    HL_LAUNCH_CHECK();
    k_inc_step<<<1, 1, 0, s>>>(step_count);
    HL_LAUNCH_CHECK();
    return hl_refresh_shadows(p, ws, s);
}
