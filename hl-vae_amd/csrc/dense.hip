// Dense-layer kernels of the HL-VAE step (SURVEY.md section 8(a) rows B, C, D and their backward):
// every product is the NT main loop of gemm_nt.h with a different LDS-staged epilogue.
#include "gemm_nt.h"

// ------------------------------------------------------------------------------------------------
// tile helpers: the fp32 tile sits in LDS as Cs[BM][CLD]
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int CLD>
__device__ __forceinline__ void tile_store_dual_bf16(const float* Cs, bf16_t* out, int ld, bf16_t* outT, int ldT,
                                                     int m0, int n0, int Mrows, int Ncols) {
    // normal layout: consecutive lanes -> consecutive columns
    for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
        const int r = idx / BN, c = idx % BN;
        if (m0 + r < Mrows && n0 + c < Ncols) out[(size_t)(m0 + r) * ld + n0 + c] = f2bf(Cs[r * CLD + c]);
    }
    if (outT != nullptr) {   // transposed layout: consecutive lanes -> consecutive rows
        for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
            const int c = idx / BM, r = idx % BM;
            if (m0 + r < Mrows && n0 + c < Ncols) outT[(size_t)(n0 + c) * ldT + m0 + r] = f2bf(Cs[r * CLD + c]);
        }
    }
}

template <int BM, int BN, int CLD>
__device__ __forceinline__ void tile_colsum_atomic(const float* Cs, float* dst, int n0, int Ncols) {
    for (int c = threadIdx.x; c < BN; c += HL_THREADS) {
        if (n0 + c >= Ncols) continue;
        float s = 0.f;
        for (int r = 0; r < BM; ++r) s += Cs[r * CLD + c];
        atomicAdd(dst + n0 + c, s);
    }
}

// ------------------------------------------------------------------------------------------------
// generic fp32-out GEMM (weight gradients, unit tests).  Rows of C may be remapped in two bands so
// that the stacked [mu; log_var] operand writes the two reference weight tensors directly.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(HL_THREADS) void k_gemm_f32(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb, float* __restrict__ C,
                                                         int ldc, int M, int N, int K, int band, int band_rows,
                                                         float* __restrict__ C2, int tiles_m, int tiles_n, int n_fast) {
    using G = GemmNT<BM, BN, BK, WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    // consecutive logical ids (= same XCD) walk the tiles that share the LARGER operand panel
    const int lid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
    const int tm = n_fast ? lid / tiles_n : lid % tiles_m, tn = n_fast ? lid % tiles_n : lid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    typename G::Acc acc;
    G::zero(acc);
    G::run(A, lda, B, ldb, m0, n0, M, N, 0, K, smem, acc);
    G::to_lds(acc, smem);
    const float* Cs = reinterpret_cast<const float*>(smem);
    for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
        const int r = idx / BN, c = idx % BN;
        const int gr = m0 + r, gc = n0 + c;
        if (gr >= M || gc >= N) continue;
        if (band <= 0) {
            C[(size_t)gr * ldc + gc] = Cs[r * G::CLD + c];
        } else {   // rows [0,band_rows) -> C, rows [band, band+band_rows) -> C2, others dropped
            if (gr < band_rows) C[(size_t)gr * ldc + gc] = Cs[r * G::CLD + c];
            else if (gr >= band && gr < band + band_rows) C2[(size_t)(gr - band) * ldc + gc] = Cs[r * G::CLD + c];
        }
    }
}

// split-K partial products into fp32 slabs  slab[s][M][ldn]
template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(HL_THREADS) void k_gemm_splitk(const bf16_t* __restrict__ A, int lda,
                                                            const bf16_t* __restrict__ B, int ldb,
                                                            float* __restrict__ slab, int ldn, int M, int N, int K,
                                                            int ksteps_per_split, int tiles_m, int tiles_n, int S) {
    using G = GemmNT<BM, BN, BK, WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    // all tiles of one K-slice run on one XCD: the slice of A and of B is pulled into that L2 once
    const int tiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, tiles * S);
    const int s = lid / tiles, tl = lid % tiles;
    const int m0 = (tl / tiles_n) * BM, n0 = (tl % tiles_n) * BN;
    const int kb = s * ksteps_per_split * BK;
    int ke = kb + ksteps_per_split * BK;
    if (ke > K) ke = K;
    typename G::Acc acc;
    G::zero(acc);
    G::run(A, lda, B, ldb, m0, n0, M, N, kb, ke, smem, acc);
    G::to_lds(acc, smem);
    const float* Cs = reinterpret_cast<const float*>(smem);
    float* out = slab + (size_t)s * M * ldn;
    for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
        const int r = idx / BN, c = idx % BN;
        if (m0 + r < M && n0 + c < N) out[(size_t)(m0 + r) * ldn + n0 + c] = Cs[r * G::CLD + c];
    }
}

// ------------------------------------------------------------------------------------------------
// slab reduction + activation.  MODE 0: out = relu(sum + bias)       (encoder trunk, HLVAE.py:316)
//                               MODE 1: out = sum * (ref > 0)        (ReLU backward), bias-grad colsum
// rows >= B are written as zero (batch padding).  32 x 32 tiles, one float4 per lane and slab.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(HL_THREADS) void k_reduce_act(const float* __restrict__ slab, int S, int M, int ldn,
                                                           const float* __restrict__ bias, int nvalid,
                                                           const bf16_t* __restrict__ ref, bf16_t* __restrict__ out,
                                                           bf16_t* __restrict__ outT, int ldT, int B,
                                                           float* __restrict__ gbias) {
    constexpr int BM = 32, BN = 32, CLD = BN + 1;
    __shared__ float Cs[BM * CLD];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
    const int gr = m0 + r, gc = n0 + c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gr < B) {
        const float* src = slab + (size_t)gr * ldn + gc;
        const size_t sstride = (size_t)M * ldn;
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            const float4 t = *reinterpret_cast<const float4*>(src + s * sstride);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float* vv = &v.x;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = (gc + k < nvalid) ? vv[k] + bias[gc + k] : 0.f;
                vv[k] = t > 0.f ? t : 0.f;
            }
        } else {
            const uint2 rf = *reinterpret_cast<const uint2*>(ref + (size_t)gr * ldn + gc);
            const bf16_t rr[4] = {(bf16_t)(rf.x & 0xffff), (bf16_t)(rf.x >> 16), (bf16_t)(rf.y & 0xffff), (bf16_t)(rf.y >> 16)};
#pragma unroll
            for (int k = 0; k < 4; ++k) vv[k] = (gc + k < nvalid && bf2f(rr[k]) > 0.f) ? vv[k] : 0.f;
        }
    }
    {   // row-major store: 4 bf16 = 8 bytes per lane
        uint2 pk;
        pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
        pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
        *reinterpret_cast<uint2*>(out + (size_t)gr * ldn + gc) = pk;
    }
    Cs[r * CLD + c4 + 0] = v.x; Cs[r * CLD + c4 + 1] = v.y; Cs[r * CLD + c4 + 2] = v.z; Cs[r * CLD + c4 + 3] = v.w;
    __syncthreads();
    {   // transposed store: lane -> (column, 4 consecutive rows)
        const int c = threadIdx.x >> 3, r4 = (threadIdx.x & 7) * 4;
        uint2 pk;
        pk.x = (uint32_t)f2bf(Cs[(r4 + 0) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 1) * CLD + c]) << 16);
        pk.y = (uint32_t)f2bf(Cs[(r4 + 2) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 3) * CLD + c]) << 16);
        *reinterpret_cast<uint2*>(outT + (size_t)(n0 + c) * ldT + m0 + r4) = pk;
    }
    if (MODE == 1 && gbias != nullptr && threadIdx.x < BN && n0 + threadIdx.x < nvalid) {
        float sum = 0.f;
#pragma unroll 8
        for (int rr = 0; rr < BM; ++rr) sum += Cs[rr * CLD + threadIdx.x];
        atomicAdd(gbias + n0 + threadIdx.x, sum);
    }
}

// ------------------------------------------------------------------------------------------------
// GEMM + elementwise epilogue into dual bf16.  MODE 0: relu(acc + bias)   (decoder trunk, HLVAE.py:336)
//                                             MODE 1: acc * (ref > 0)    (d trunk of the encoder), colsum
// ------------------------------------------------------------------------------------------------
template <int BK, int MODE>
__global__ __launch_bounds__(HL_THREADS) void k_gemm_act(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ Bm, int ldb, int M, int N, int K,
                                                         const float* __restrict__ bias, int nvalid,
                                                         const bf16_t* __restrict__ ref, bf16_t* __restrict__ out,
                                                         int ldo, bf16_t* __restrict__ outT, int ldT, int B,
                                                         float* __restrict__ gbias) {
    using G = GemmNT<64, 64, BK, 2, 2>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    typename G::Acc acc;
    G::zero(acc);
    G::run(A, lda, Bm, ldb, m0, n0, M, N, 0, K, smem, acc);
    G::to_lds(acc, smem);
    float* Cs = reinterpret_cast<float*>(smem);
    for (int idx = threadIdx.x; idx < 64 * 64; idx += HL_THREADS) {
        const int r = idx / 64, c = idx % 64;
        const int gr = m0 + r, gc = n0 + c;
        float v = Cs[r * G::CLD + c];
        if (gr < B && gc < nvalid) {
            if (MODE == 0) {
                v += bias[gc];
                v = v > 0.f ? v : 0.f;
            } else {
                v = bf2f(ref[(size_t)gr * ldo + gc]) > 0.f ? v : 0.f;
            }
        } else {
            v = 0.f;
        }
        Cs[r * G::CLD + c] = v;
    }
    __syncthreads();
    tile_store_dual_bf16<64, 64, G::CLD>(Cs, out, ldo, outT, ldT, m0, n0, M, N);
    if (MODE == 1 && gbias != nullptr) tile_colsum_atomic<64, 64, G::CLD>(Cs, gbias, n0, nvalid);
}

// ------------------------------------------------------------------------------------------------
// encoder head + reparameterisation (rows B tail + C): [mu | lv_raw] = T * [Wmu; Wlv]^T + bias,
// lv = clamp(lv_raw, -15, 15) (HLVAE.py:319), z = mu + eps * exp(lv / 2) (HLVAE.py:360-362).
// Tile: 64 rows x 2*LP columns (mu in [0,LP), log-var in [LP,2LP)).
// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. 2011): 4 x 32 random bits per (counter, key)
__device__ __forceinline__ uint4 philox4x32(uint4 ctr, uint2 key) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint32_t idx) {
    const uint4 r = philox4x32(make_uint4(idx, 0u, (uint32_t)offset, (uint32_t)(offset >> 32)),
                               make_uint2((uint32_t)seed, (uint32_t)(seed >> 32)));
    const float u1 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f);     // (0, 1]
    const float u2 = (float)(r.y >> 8) * (1.0f / 16777216.0f);              // [0, 1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);     // Box-Muller
}

template <int LP>
__global__ __launch_bounds__(HL_THREADS) void k_mid_fwd(const bf16_t* __restrict__ T, int ldt,
                                                        const bf16_t* __restrict__ Wml, int K,
                                                        const float* __restrict__ bmu, const float* __restrict__ blv,
                                                        const float* __restrict__ eps, float* __restrict__ eps_out,
                                                        const uint64_t* __restrict__ rng, uint64_t rng_host_offset,
                                                        float* __restrict__ mu,
                                                        float* __restrict__ lv, float* __restrict__ z,
                                                        bf16_t* __restrict__ zb, bf16_t* __restrict__ zbT, int Bp, int B,
                                                        int L, double* __restrict__ klpart) {
    // noise: eps given -> use it; else rng != NULL -> Philox(seed = rng[0], offset = rng[1] + host offset), stored to
    // eps_out for the backward pass; else z = mu (deterministic pass of get_test_samples, HLVAE.py:472)
    using G = GemmNT<64, 2 * LP, 64, 2, 2>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    __shared__ double klred[4];
    const int m0 = blockIdx.x * 64;
    typename G::Acc acc;
    G::zero(acc);
    G::run(T, ldt, Wml, ldt, m0, 0, Bp, 2 * LP, 0, K, smem, acc);
    G::to_lds(acc, smem);
    float* Cs = reinterpret_cast<float*>(smem);
    uint64_t seed = 0, off = 0;
    if (eps == nullptr && rng != nullptr) { seed = rng[0]; off = rng[1] + rng_host_offset; }
    double klacc = 0.0;
    for (int idx = threadIdx.x; idx < 64 * LP; idx += HL_THREADS) {
        const int r = idx / LP, j = idx % LP;
        const int gr = m0 + r;
        float zv = 0.f;
        if (gr < B && j < L) {
            const size_t o = (size_t)gr * L + j;
            const float m = Cs[r * G::CLD + j] + bmu[j];
            float l = Cs[r * G::CLD + LP + j] + blv[j];
            l = fminf(fmaxf(l, -15.f), 15.f);
            float e = 0.f;
            if (eps != nullptr) e = eps[o];
            else if (rng != nullptr) e = philox_normal(seed, off, (uint32_t)o);
            if (eps_out != nullptr) eps_out[o] = e;
            const float el = __expf(l);
            zv = m + e * sqrtf(el);
            mu[o] = m;
            lv[o] = l;
            z[o] = zv;
            klacc += (double)(-0.5f * (1.f + l - m * m - el));   // KL(q || N(0,I)) term (extension, see kl notes)
        }
        Cs[r * G::CLD + j] = zv;   // reuse the mu half of the tile as the z tile
    }
    klacc = wave_sum_d(klacc);
    if ((threadIdx.x & 63) == 0) klred[threadIdx.x >> 6] = klacc;
    __syncthreads();
    if (threadIdx.x == 0 && klpart != nullptr) klpart[blockIdx.x] = klred[0] + klred[1] + klred[2] + klred[3];
    tile_store_dual_bf16<64, LP, G::CLD>(Cs, zb, LP, zbT, Bp, m0, 0, Bp, LP);
}

// ------------------------------------------------------------------------------------------------
// backward through z, the clamp and the reparameterisation:
//   dz = dU * Wd  (A = du [Bp][hdp], B = WdT [LP][hdp]);   d mu = dz + g_mu;
//   d lv = (dz * eps * 0.5 * exp(lv/2) + g_lv) * [ -15 < lv < 15 ]
// writes dml = [d mu | d lv] (bf16, both layouts) and the two bias gradients.
// ------------------------------------------------------------------------------------------------
template <int LP>
__global__ __launch_bounds__(HL_THREADS) void k_mid_bwd(const bf16_t* __restrict__ dU, int ldu,
                                                        const bf16_t* __restrict__ WdT, int K,
                                                        const float* __restrict__ eps, const float* __restrict__ lv,
                                                        const float* __restrict__ g_mu, const float* __restrict__ g_lv,
                                                        const float* __restrict__ mu, float kl_w,
                                                        float* __restrict__ dz, bf16_t* __restrict__ dml,
                                                        bf16_t* __restrict__ dmlT, int Bp, int B, int L,
                                                        float* __restrict__ gbmu, float* __restrict__ gblv) {
    using G = GemmNT<64, LP, 64, 4, 1>;
    constexpr int CLD2 = 2 * LP + 1;
    __shared__ __attribute__((aligned(16))) char smem[(G::SMEM_BYTES > 64 * CLD2 * 4) ? G::SMEM_BYTES : 64 * CLD2 * 4];
    const int m0 = blockIdx.x * 64;
    typename G::Acc acc;
    G::zero(acc);
    G::run(dU, ldu, WdT, ldu, m0, 0, Bp, LP, 0, K, smem, acc);
    // the [64][2LP] output tile is wider than the accumulator tile: go through registers
    float vals[(64 * LP + HL_THREADS - 1) / HL_THREADS];
    {
        G::to_lds(acc, smem);
        const float* Cs = reinterpret_cast<const float*>(smem);
        int n = 0;
        for (int idx = threadIdx.x; idx < 64 * LP; idx += HL_THREADS, ++n) vals[n] = Cs[(idx / LP) * G::CLD + idx % LP];
        __syncthreads();
    }
    float* Ds = reinterpret_cast<float*>(smem);
    int n = 0;
    for (int idx = threadIdx.x; idx < 64 * LP; idx += HL_THREADS, ++n) {
        const int r = idx / LP, j = idx % LP;
        const int gr = m0 + r;
        float dm = 0.f, dl = 0.f;
        const float d = vals[n];
        if (gr < B && j < L) {
            const size_t o = (size_t)gr * L + j;
            const float l = lv[o];
            dm = d + (g_mu != nullptr ? g_mu[o] : 0.f);
            const float e = eps != nullptr ? eps[o] : 0.f;
            const float el = __expf(l);
            dl = d * e * 0.5f * sqrtf(el) + (g_lv != nullptr ? g_lv[o] : 0.f);
            if (kl_w != 0.f) {                       // d KL(q || N(0,I)): d/dmu = mu, d/dlv = (e^lv - 1)/2
                dm += kl_w * mu[o];
                dl += kl_w * 0.5f * (el - 1.f);
            }
            if (!(l > -15.f && l < 15.f)) dl = 0.f;
        }
        if (gr < Bp) dz[(size_t)gr * LP + j] = (gr < B && j < L) ? d : 0.f;
        Ds[r * CLD2 + j] = dm;
        Ds[r * CLD2 + LP + j] = dl;
    }
    __syncthreads();
    tile_store_dual_bf16<64, 2 * LP, CLD2>(Ds, dml, 2 * LP, dmlT, Bp, m0, 0, Bp, 2 * LP);
    for (int c = threadIdx.x; c < 2 * LP; c += HL_THREADS) {
        const int j = c < LP ? c : c - LP;
        if (j >= L) continue;
        float s = 0.f;
        for (int r = 0; r < 64; ++r) s += Ds[r * CLD2 + c];
        atomicAdd((c < LP ? gbmu : gblv) + j, s);
    }
}

// ------------------------------------------------------------------------------------------------
// host launchers (called from cabi.hip)
// ------------------------------------------------------------------------------------------------
int hl_launch_gemm_f32(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* C, int ldc, int M, int N, int K,
                       int band, int band_rows, float* C2, const char* label, hipStream_t s) {
    HL_REQUIRE(K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0, HLVAE_ESHAPE, "gemm_f32: K=%d lda=%d ldb=%d", K, lda, ldb);
    HL_PROF(label, s);
    const int n_fast = M >= N;      // A is the larger operand: its row panel is reused by consecutive ids
#define HL_GO(BMv, BNv, BKv, WMv, WNv)                                                                          \
    {                                                                                                           \
        const int tm = (M + BMv - 1) / BMv, tn = (N + BNv - 1) / BNv;                                           \
        k_gemm_f32<BMv, BNv, BKv, WMv, WNv><<<tm * tn, HL_THREADS, 0, s>>>(A, lda, B, ldb, C, ldc, M, N, K, band, \
                                                                          band_rows, C2, tm, tn, n_fast);       \
    }
    if (N <= 32) {
        if (K % 64 == 0) HL_GO(64, 32, 64, 4, 1) else HL_GO(64, 32, 32, 4, 1)
    } else if (M <= 64 || (long)((M + 127) / 128) * ((N + 63) / 64) < 192) {
        if (K % 64 == 0) HL_GO(64, 64, 64, 2, 2) else HL_GO(64, 64, 32, 2, 2)
    } else {
        if (K % 64 == 0) HL_GO(128, 64, 64, 2, 2) else HL_GO(128, 64, 32, 2, 2)
    }
#undef HL_GO
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_gemm_splitk(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* slab, int ldn, int M, int N, int K,
                          int S, const char* label, hipStream_t s) {
    HL_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && S >= 1, HLVAE_ESHAPE, "splitk: K=%d S=%d", K, S);
    const int ksteps = K / 64;
    const int per = (ksteps + S - 1) / S;
    HL_REQUIRE(per * (S - 1) < ksteps, HLVAE_ESHAPE, "splitk: S=%d leaves an empty split for %d k-steps", S, ksteps);
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    HL_PROF(label, s);
    k_gemm_splitk<64, 64, 64, 2, 2><<<tm * tn * S, HL_THREADS, 0, s>>>(A, lda, B, ldb, slab, ldn, M, N, K, per, tm, tn, S);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_reduce_act(int mode, const float* slab, int S, int M, int ldn, const float* bias, int nvalid,
                         const bf16_t* ref, bf16_t* out, bf16_t* outT, int ldT, int B, float* gbias, const char* label, hipStream_t s) {
    dim3 grid(ldn / 32, M / 32);
    HL_REQUIRE(ldn % 32 == 0 && M % 32 == 0 && ldT % 4 == 0, HLVAE_ESHAPE, "reduce_act: M=%d ldn=%d", M, ldn);
    HL_PROF(label, s);
    if (mode == 0)
        k_reduce_act<0><<<grid, HL_THREADS, 0, s>>>(slab, S, M, ldn, bias, nvalid, ref, out, outT, ldT, B, gbias);
    else
        k_reduce_act<1><<<grid, HL_THREADS, 0, s>>>(slab, S, M, ldn, bias, nvalid, ref, out, outT, ldT, B, gbias);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_gemm_act(int mode, const bf16_t* A, int lda, const bf16_t* Bm, int ldb, int M, int N, int K,
                       const float* bias, int nvalid, const bf16_t* ref, bf16_t* out, int ldo, bf16_t* outT, int ldT,
                       int B, float* gbias, const char* label, hipStream_t s) {
    HL_REQUIRE(K % 32 == 0 && M % 64 == 0 && N % 64 == 0, HLVAE_ESHAPE, "gemm_act: M=%d N=%d K=%d", M, N, K);
    dim3 grid(N / 64, M / 64);
    HL_PROF(label, s);
    if (K % 64 == 0) {
        if (mode == 0)
            k_gemm_act<64, 0><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
        else
            k_gemm_act<64, 1><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
    } else {
        if (mode == 0)
            k_gemm_act<32, 0><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
        else
            k_gemm_act<32, 1><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
    }
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_mid_fwd(int Lp, const bf16_t* T, int ldt, const bf16_t* Wml, int K, const float* bmu, const float* blv,
                      const float* eps, float* eps_out, const uint64_t* rng, uint64_t rng_off, float* mu, float* lv,
                      float* z, bf16_t* zb, bf16_t* zbT, int Bp, int B, int L, double* klpart, hipStream_t s) {
    HL_REQUIRE(K % 64 == 0 && Bp % 64 == 0, HLVAE_ESHAPE, "mid_fwd: K=%d Bp=%d", K, Bp);
    HL_PROF("enc_head_reparam", s);
    if (Lp == 32)
        k_mid_fwd<32><<<Bp / 64, HL_THREADS, 0, s>>>(T, ldt, Wml, K, bmu, blv, eps, eps_out, rng, rng_off, mu, lv, z, zb, zbT, Bp, B, L, klpart);
    else if (Lp == 64)
        k_mid_fwd<64><<<Bp / 64, HL_THREADS, 0, s>>>(T, ldt, Wml, K, bmu, blv, eps, eps_out, rng, rng_off, mu, lv, z, zb, zbT, Bp, B, L, klpart);
    else
        HL_REQUIRE(false, HLVAE_EINVAL, "latent_dim padded to %d is not supported (max 64)", Lp);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_mid_bwd(int Lp, const bf16_t* dU, int ldu, const bf16_t* WdT, int K, const float* eps, const float* lv,
                      const float* g_mu, const float* g_lv, const float* mu, float kl_w, float* dz, bf16_t* dml,
                      bf16_t* dmlT, int Bp, int B, int L, float* gbmu, float* gblv, hipStream_t s) {
    HL_REQUIRE(K % 64 == 0 && Bp % 64 == 0, HLVAE_ESHAPE, "mid_bwd: K=%d Bp=%d", K, Bp);
    HL_PROF("dz_reparam_bwd", s);
    if (Lp == 32)
        k_mid_bwd<32><<<Bp / 64, HL_THREADS, 0, s>>>(dU, ldu, WdT, K, eps, lv, g_mu, g_lv, mu, kl_w, dz, dml, dmlT, Bp, B, L, gbmu, gblv);
    else if (Lp == 64)
        k_mid_bwd<64><<<Bp / 64, HL_THREADS, 0, s>>>(dU, ldu, WdT, K, eps, lv, g_mu, g_lv, mu, kl_w, dz, dml, dmlT, Bp, B, L, gbmu, gblv);
    else
        HL_REQUIRE(false, HLVAE_EINVAL, "latent_dim padded to %d is not supported (max 64)", Lp);
    HL_LAUNCH_CHECK();
    return 0;
}
