// KL terms on device.
//  * closed-form KL(q(z|x) || N(0, I)) -- NOT in the reference (the reference's only KL is the
//    GP prior); used by the GP-free configurations (BASELINE.json configs 2-4), parity pinned analytically.
#include "common.h"

__global__ void k_kl_std(const float* __restrict__ mu, const float* __restrict__ lv, int n, float weight,
                         float* __restrict__ g_mu, float* __restrict__ g_lv, double* __restrict__ out) {
    double acc = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float m = mu[i], l = lv[i];
        const float e = __expf(l);
        acc += (double)(-0.5f * (1.f + l - m * m - e));
        if (g_mu != nullptr) g_mu[i] = weight * m;
        if (g_lv != nullptr) g_lv[i] = weight * 0.5f * (e - 1.f);
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc * (double)weight);
}

int hl_launch_kl_std(const hlvae_ws* ws, int B, int L, float weight, float* g_mu, float* g_lv, hipStream_t s) {
    HL_CHECK(hipMemsetAsync(ws->scal + 1, 0, sizeof(double), s));
    const int n = B * L;
    int blocks = (n + 255) / 256;
    if (blocks > 256) blocks = 256;
    HL_PROF("kl_std_normal", s);
    k_kl_std<<<blocks, 256, 0, s>>>(ws->mu, ws->lv, n, weight, g_mu, g_lv, ws->scal + 1);
    HL_LAUNCH_CHECK();
    return 0;
}
