// Dense-layer kernels of the HL-VAE step (SURVEY.md section 8(a) rows B, C, D and their backward):
// every product is the NT main loop of gemm_nt.h with a different LDS-staged epilogue.
#include "gemm_nt.h"
#include "gemm_dma.h"
#include "adam.h"
#include <stdlib.h>

// which GEMM main loop the K % 64 == 0 launches use: 1 = LDS-DMA staged (gemm_dma.h, round 3), 0 = register staged (gemm_nt.h).
// HL_GEMM_CORE=nt in the environment selects the old core (A/B runs on one box).
int g_hl_gemm_dma = -1;
static bool hl_use_dma() {
    if (g_hl_gemm_dma < 0) {
        const char* e = getenv("HL_GEMM_CORE");
        g_hl_gemm_dma = (e != nullptr && e[0] == 'n') ? 0 : 1;
    }
    return g_hl_gemm_dma != 0;
}

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
                                                         float* __restrict__ C2, int tiles_m, int tiles_n, int n_fast, int kper,
                                                         const int32_t* __restrict__ rowmap) {
    // rowmap != nullptr: output row gr is stored at row rowmap[gr] of C (y_layer's weight gradient: dY^T arrives in the head
    // kernel's variable order, the gradient arena keeps the master's)
    // kper < K: split-K -- gridDim.x = tiles x slices, slice s covers k in [s kper, (s + 1) kper) and ADDS its partial tile
    // (fp32 atomics) into a C the launcher has cleared.  For weight gradients with few output tiles and a long batch axis
    // (64-feature models at 4096 rows: 40 tiles x 64 k-steps otherwise)
    using G = GemmNT<BM, BN, BK, WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    // consecutive logical ids (= same XCD) walk the tiles that share the LARGER operand panel
    const int tiles = tiles_m * tiles_n;
    const int ks = blockIdx.x / tiles;
    const int lid = xcd_remap(blockIdx.x - ks * tiles, tiles);
    const int tm = n_fast ? lid / tiles_n : lid % tiles_m, tn = n_fast ? lid % tiles_n : lid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    typename G::Acc acc;
    G::zero(acc);
    G::run(A, lda, B, ldb, m0, n0, M, N, ks * kper, min(K, (ks + 1) * kper), smem, acc);
    G::to_lds(acc, smem);
    const float* Cs = reinterpret_cast<const float*>(smem);
    if (kper < K) {
        for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
            const int r = idx / BN, c = idx % BN;
            const int gr = m0 + r, gc = n0 + c;
            if (gr < M && gc < N) atomicAdd(C + (size_t)(rowmap != nullptr ? rowmap[gr] : gr) * ldc + gc, Cs[r * G::CLD + c]);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
        const int r = idx / BN, c = idx % BN;
        const int gr = m0 + r, gc = n0 + c;
        if (gr >= M || gc >= N) continue;
        if (band <= 0) {
            C[(size_t)(rowmap != nullptr ? rowmap[gr] : gr) * ldc + gc] = Cs[r * G::CLD + c];
        } else {   // rows [0,band_rows) -> C, rows [band, band+band_rows) -> C2, others dropped
            if (gr < band_rows) C[(size_t)gr * ldc + gc] = Cs[r * G::CLD + c];
            else if (gr >= band && gr < band + band_rows) C2[(size_t)(gr - band) * ldc + gc] = Cs[r * G::CLD + c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient AND optimiser step in one kernel (single-process training step):
//     g = A B^T (tile in LDS)  ->  Adam on the master / m / v cells the tile owns  ->  both bf16 shadows of the new values.
// As separate launches (k_gemm_f32*, k_adam_tiled) the gradient makes a round trip through HBM (8 of 36 B per parameter), the
// optimiser is two more dependent launches on the critical path of the replayed step, and the two HBM-bound optimiser launches
// must be kept apart.  Up to three products per launch (the ones that become computable together).  64 x 64 tiles: three
// workgroups per CU, and every lane requests its 12 float4 of optimiser state BEFORE the product, so the streaming part is
// in flight under the MFMA loop (round 1 applied Adam in the epilogue of the 128-row-tile gradient kernel, loads after the
// product: 1-2 workgroups per CU could not keep enough bytes in flight, 0.190 vs 0.166 ms/step).
// Registers decide the rest: with all 12 float4 prefetched the kernel needed 184 VGPRs = two workgroups per CU, 512 slots for
// the 816 tiles of y_layer's gradient, i.e. two rounds (34 us).  Only the masters are requested before the product; m and v
// follow once the product's registers are free: 124 VGPRs, four workgroups per CU (LDS 36.9 KB each), every tile resident at
// once: 34.1 -> 28.6 us (y_layer), 28.9 -> 24.8 us (the other three), step 0.1437 -> 0.136 ms.
// Completion tickets as in k_adam_tiled: the launches of one step may run concurrently; the last workgroup commits t.
// ------------------------------------------------------------------------------------------------
template <int BK>
__global__ __launch_bounds__(HL_THREADS, 4) void k_gemm_adam(AdamGemmGroup g, float* __restrict__ P, float* __restrict__ M1,
                                                          float* __restrict__ M2, int64_t* __restrict__ step_count, float lr,
                                                          float b1, float b2, float eps, float gscale, unsigned ticket_total,
                                                          float* __restrict__ Gflat, long flat_lo4, long flat_n4, unsigned long long* stamp,
                                                          unsigned long long* tick_shards) {
    HL_STAMP_T0(stamp);
    using G = GemmNT<64, 64, BK, 2, 2>;
    constexpr int CLD = G::CLD;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    int pi = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (k < g.n && (int)blockIdx.x >= g.p[k].tile0) pi = k;
    const AdamGemmProb& q = g.p[pi];
    // problem 0 starts at workgroup 0: its XCD-contiguous order is exact (consecutive ids share the row panel of A)
    const int lid = pi == 0 ? xcd_remap(blockIdx.x, q.tiles_m * q.tiles_n) : (int)blockIdx.x - q.tile0;
    const int m0 = (lid / q.tiles_n) * 64, n0 = (lid % q.tiles_n) * 64;
    const int M = q.M, N = q.N;
    const int c4 = (threadIdx.x & 15) * 4, rq = threadIdx.x >> 4;         // 16 float4 per tile row, 16 rows per pass
    float4 p[4], m[4], v[4];
    int o[4];                                                     // (arena offsets fit 31 bits: checked by the launcher)
    bool in[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gr = m0 + rq + 16 * i;
        long base = -1;
        if (n0 + c4 < N) {
            if (q.band <= 0) {
                if (gr < M) base = q.off + (long)(q.rowmap != nullptr ? q.rowmap[gr] : gr) * N;
            } else if (gr < q.band_rows) {
                base = q.off + (long)gr * N;
            } else if (gr >= q.band && gr < q.band + q.band_rows) {
                base = q.off2 + (long)(gr - q.band) * N;
            }
        }
        in[i] = base >= 0;
        o[i] = (int)(in[i] ? base + n0 + c4 : q.off);
        p[i] = *reinterpret_cast<const float4*>(P + o[i]);
    }
    typename G::Acc acc;
    G::zero(acc);
    G::run(q.A, q.lda, q.B, q.ldb, m0, n0, M, N, 0, g.K, smem, acc);
    G::to_lds(acc, smem);
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                                      // (the product's registers are free again)
        m[i] = *reinterpret_cast<const float4*>(M1 + o[i]);
        v[i] = *reinterpret_cast<const float4*>(M2 + o[i]);
    }
    const AdamScalars a = adam_scalars((float)(step_count[0] + 1), lr, b1, b2, eps, gscale);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rq + 16 * i;
        const float4 gr4 = *reinterpret_cast<const float4*>(Cs + r * CLD + c4);
        p[i].x = adam_one(p[i].x, gr4.x, m[i].x, v[i].x, a);
        p[i].y = adam_one(p[i].y, gr4.y, m[i].y, v[i].y, a);
        p[i].z = adam_one(p[i].z, gr4.z, m[i].z, v[i].z, a);
        p[i].w = adam_one(p[i].w, gr4.w, m[i].w, v[i].w, a);
        if (in[i]) {
            *reinterpret_cast<float4*>(P + o[i]) = p[i];
            *reinterpret_cast<float4*>(M1 + o[i]) = m[i];
            *reinterpret_cast<float4*>(M2 + o[i]) = v[i];
            uint2 pk;
            pk.x = (uint32_t)f2bf(p[i].x) | ((uint32_t)f2bf(p[i].y) << 16);
            pk.y = (uint32_t)f2bf(p[i].z) | ((uint32_t)f2bf(p[i].w) << 16);
            *reinterpret_cast<uint2*>(q.sh + (size_t)(m0 + r) * q.ldd + n0 + c4) = pk;
        }
        // (each lane rewrites only the four cells it has just read: no barrier needed before the write)
        *reinterpret_cast<float4*>(Cs + r * CLD + c4) = in[i] ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (q.shT != nullptr) {                // block-uniform
        __syncthreads();
        // transposed shadow: lane -> (column, 4 consecutive rows), one 8-byte store (rows outside the matrix: zeros = padding)
        const int r4 = (threadIdx.x & 15) * 4, cq = threadIdx.x >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cq + 16 * i;
            if (m0 + r4 < M && n0 + c < N) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(Cs[(r4 + 0) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 1) * CLD + c]) << 16);
                pk.y = (uint32_t)f2bf(Cs[(r4 + 2) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 3) * CLD + c]) << 16);
                *reinterpret_cast<uint2*>(q.shT + (size_t)(n0 + c) * q.ldT + m0 + r4) = pk;
            }
        }
    }
    if (flat_n4 > 0 && blockIdx.x == 0) {
        // workgroup 0, after its tile: Adam on a range of the small flat region whose gradients the kernel in FRONT of this launch
        // on the same stream accumulated with atomics (the four bias vectors of the fused middle), zeroing them for the next
        // step -- so that the small-region launch on the side stream depends on the head kernel only.  (As an extra workgroup
        // with a branch at the top of the kernel the whole launch was 60 % slower.)
        float4* P4 = reinterpret_cast<float4*>(P) + flat_lo4;
        float4* G4 = reinterpret_cast<float4*>(Gflat) + flat_lo4;
        float4* M14 = reinterpret_cast<float4*>(M1) + flat_lo4;
        float4* M24 = reinterpret_cast<float4*>(M2) + flat_lo4;
        for (long i = threadIdx.x; i < flat_n4; i += HL_THREADS) {
            float4 pp = P4[i], gr = G4[i], mm = M14[i], vv = M24[i];
            pp.x = adam_one(pp.x, gr.x, mm.x, vv.x, a);
            pp.y = adam_one(pp.y, gr.y, mm.y, vv.y, a);
            pp.z = adam_one(pp.z, gr.z, mm.z, vv.z, a);
            pp.w = adam_one(pp.w, gr.w, mm.w, vv.w, a);
            P4[i] = pp;
            M14[i] = mm;
            M24[i] = vv;
            G4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (ticket_total != 0) {
        __syncthreads();
        if (threadIdx.x == 0) hl_take_ticket(step_count, tick_shards, ticket_total);
    }
    HL_STAMP_END(stamp);
}

// The same kernel on the LDS-DMA core (gemm_dma.h).  The operand tiles need no staging registers, so ALL the optimiser state a
// lane owns is requested before the product (the register-staged kernel above could afford the masters only: m and v followed
// behind the product, and every workgroup of the single resident round then waited for them in phase).
// BM x BN = tile shape (template): 64 x 64 is one resident round for the D4 matrices (816 / 664 tiles on 1024 slots); smaller
// tiles trade operand re-reads through L2 for several rounds per CU, i.e. workgroups in different phases -- loads, product,
// stores -- at the same time instead of the whole chip reading, then multiplying, then writing (ubench: tools/ubench).
template <int BM, int BN, int NBUF, int MINW>
__global__ __launch_bounds__(HL_THREADS, MINW) void k_gemm_adam_dma(AdamGemmGroup g, float* __restrict__ P, float* __restrict__ M1,
                                                                 float* __restrict__ M2, int64_t* __restrict__ step_count, float lr,
                                                                 float b1, float b2, float eps, float gscale, unsigned ticket_total,
                                                                 float* __restrict__ Gflat, long flat_lo4, long flat_n4,
                                                                 unsigned long long* stamp, int stagger,
                                                                 unsigned long long* tick_shards) {
    HL_STAMP_T0(stamp);
    using G = GemmDMA<BM, BN, 2, 2, NBUF, BN + 4>;
    constexpr int CLD = G::CLD;
    constexpr int C4 = BN / 4, RPP = HL_THREADS / C4, NP = BM / RPP;          // float4 per tile row, rows per pass, passes
    static_assert(BM % RPP == 0 && NP >= 1, "tile rows split over the passes");
    __shared__ __attribute__((aligned(1024))) char smem[G::SMEM_BYTES];
    int pi = 0;
#pragma unroll
    for (int k = 1; k < 3; ++k)
        if (k < g.n && (int)blockIdx.x >= g.p[k].tile0) pi = k;
    const AdamGemmProb& q = g.p[pi];
    const int lid = pi == 0 ? xcd_remap(blockIdx.x, q.tiles_m * q.tiles_n) : (int)blockIdx.x - q.tile0;
    const int m0 = (lid / q.tiles_n) * BM, n0 = (lid % q.tiles_n) * BN;
    const int M = q.M, N = q.N;
    const int c4 = (threadIdx.x % C4) * 4, rq = threadIdx.x / C4;
    float4 p[NP], m[NP], v[NP];
    int o[NP];
    bool in[NP];
    // master rows of this lane's tile rows: all the (unconditional, clamped) row-map loads are issued before the first is used --
    // inside the address computation each one was followed by its own wait: dependent round trips in front of the state's loads
    int mrow[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int gr = m0 + rq + RPP * i;
        mrow[i] = q.rowmap != nullptr ? q.rowmap[min(gr, M - 1)] : gr;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int gr = m0 + rq + RPP * i;
        long base = -1;
        if (n0 + c4 < N) {
            if (q.band <= 0) {
                if (gr < M) base = q.off + (long)mrow[i] * N;
            } else if (gr < q.band_rows) {
                base = q.off + (long)gr * N;
            } else if (gr >= q.band && gr < q.band + q.band_rows) {
                base = q.off2 + (long)(gr - q.band) * N;
            }
        }
        in[i] = base >= 0;
        o[i] = (int)(in[i] ? base + n0 + c4 : q.off);
    }
    // stagger: the workgroups are all resident at once (one round), so they would all read their state, then all multiply (an
    // L2 -> LDS phase: 816 tiles x 131 KB of operands, HBM idle), then all write.  Odd tiles therefore run the product FIRST
    // and request their state behind it: while one half of the chip streams from HBM the other half multiplies out of L2.
    const bool late = stagger && (lid & 1);
    if (!late) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            p[i] = *reinterpret_cast<const float4*>(P + o[i]);
            m[i] = *reinterpret_cast<const float4*>(M1 + o[i]);
            v[i] = *reinterpret_cast<const float4*>(M2 + o[i]);
        }
    }
    typename G::Acc acc;
    G::zero(acc);
    G::run(q.A, q.lda, q.B, q.ldb, m0, n0, M, N, 0, g.K, smem, acc);
    if (late) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            p[i] = *reinterpret_cast<const float4*>(P + o[i]);
            m[i] = *reinterpret_cast<const float4*>(M1 + o[i]);
            v[i] = *reinterpret_cast<const float4*>(M2 + o[i]);
        }
    }
    G::to_lds(acc, smem);
    float* Cs = reinterpret_cast<float*>(smem);
    const AdamScalars a = adam_scalars((float)(step_count[0] + 1), lr, b1, b2, eps, gscale);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int r = rq + RPP * i;
        const float4 gr4 = *reinterpret_cast<const float4*>(Cs + r * CLD + c4);
        p[i].x = adam_one(p[i].x, gr4.x, m[i].x, v[i].x, a);
        p[i].y = adam_one(p[i].y, gr4.y, m[i].y, v[i].y, a);
        p[i].z = adam_one(p[i].z, gr4.z, m[i].z, v[i].z, a);
        p[i].w = adam_one(p[i].w, gr4.w, m[i].w, v[i].w, a);
        if (in[i]) {
            *reinterpret_cast<float4*>(P + o[i]) = p[i];
            *reinterpret_cast<float4*>(M1 + o[i]) = m[i];
            *reinterpret_cast<float4*>(M2 + o[i]) = v[i];
            uint2 pk;
            pk.x = (uint32_t)f2bf(p[i].x) | ((uint32_t)f2bf(p[i].y) << 16);
            pk.y = (uint32_t)f2bf(p[i].z) | ((uint32_t)f2bf(p[i].w) << 16);
            *reinterpret_cast<uint2*>(q.sh + (size_t)(m0 + r) * q.ldd + n0 + c4) = pk;
        }
        // (each lane rewrites only the four cells it has just read: no barrier needed before the write)
        *reinterpret_cast<float4*>(Cs + r * CLD + c4) = in[i] ? p[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (q.shT != nullptr) {                // block-uniform
        __syncthreads();
        // transposed shadow: lane -> (column, 4 consecutive rows), one 8-byte store (rows outside the matrix: zeros = padding)
        constexpr int R4 = BM / 4, CPP = HL_THREADS / R4;                       // row quads per column, columns per pass
        const int r4 = (threadIdx.x % R4) * 4, cq = threadIdx.x / R4;
#pragma unroll
        for (int i = 0; i < BN / CPP; ++i) {
            const int c = cq + CPP * i;
            if (m0 + r4 < M && n0 + c < N) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(Cs[(r4 + 0) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 1) * CLD + c]) << 16);
                pk.y = (uint32_t)f2bf(Cs[(r4 + 2) * CLD + c]) | ((uint32_t)f2bf(Cs[(r4 + 3) * CLD + c]) << 16);
                *reinterpret_cast<uint2*>(q.shT + (size_t)(n0 + c) * q.ldT + m0 + r4) = pk;
            }
        }
    }
    if (flat_n4 > 0 && blockIdx.x == 0) {
        float4* P4 = reinterpret_cast<float4*>(P) + flat_lo4;
        float4* G4 = reinterpret_cast<float4*>(Gflat) + flat_lo4;
        float4* M14 = reinterpret_cast<float4*>(M1) + flat_lo4;
        float4* M24 = reinterpret_cast<float4*>(M2) + flat_lo4;
        for (long i = threadIdx.x; i < flat_n4; i += HL_THREADS) {
            float4 pp = P4[i], gr = G4[i], mm = M14[i], vv = M24[i];
            pp.x = adam_one(pp.x, gr.x, mm.x, vv.x, a);
            pp.y = adam_one(pp.y, gr.y, mm.y, vv.y, a);
            pp.z = adam_one(pp.z, gr.z, mm.z, vv.z, a);
            pp.w = adam_one(pp.w, gr.w, mm.w, vv.w, a);
            P4[i] = pp;
            M14[i] = mm;
            M24[i] = vv;
            G4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    if (ticket_total != 0) {
        __syncthreads();
        if (threadIdx.x == 0) hl_take_ticket(step_count, tick_shards, ticket_total);
    }
    HL_STAMP_END(stamp);
}

// up to three independent products of the same depth K in ONE launch (the weight gradients that become computable at
// the same moment of the backward pass: dW1, dWd, d[Wmu; Wlv]).  The two small ones are 8 workgroups each: as launches
// of their own they cost a dependent ~10 us kernel (or a cross-queue graph edge) apiece, here they ride along.
// (GemmProb / GemmGroup: common.h)

template <int BM, int BN, int BK, int WM, int WN>
__global__ __launch_bounds__(HL_THREADS) void k_gemm_f32_group(GemmGroup g) {
    using G = GemmNT<BM, BN, BK, WM, WN>;
    __shared__ __attribute__((aligned(16))) char smem[G::SMEM_BYTES];
    // split-K (g.ksplit > 1): gridDim.x = tiles of all problems x slices; every slice adds its partial tiles with fp32 atomics
    // into outputs the launcher has cleared
    const int bid = (int)blockIdx.x % g.tiles_total, ks = (int)blockIdx.x / g.tiles_total;
    int pi = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < g.n && bid >= g.p[k].tile0) pi = k;
    const GemmProb& q = g.p[pi];
    const int32_t* __restrict__ rowmap = q.rowmap;
    // problem 0 starts at workgroup 0, so its XCD-contiguous remap is exact; the small problems do not care
    const int lid = pi == 0 ? xcd_remap(bid, q.tiles_m * q.tiles_n) : bid - q.tile0;
    const int tm = q.n_fast ? lid / q.tiles_n : lid % q.tiles_m, tn = q.n_fast ? lid % q.tiles_n : lid / q.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int M = q.M, N = q.N, ldc = q.ldc, band = q.band, band_rows = q.band_rows;
    float* __restrict__ C = q.C;
    float* __restrict__ C2 = q.C2;
    typename G::Acc acc;
    G::zero(acc);
    G::run(q.A, q.lda, q.B, q.ldb, m0, n0, M, N, ks * g.kper, min(g.K, (ks + 1) * g.kper), smem, acc);
    G::to_lds(acc, smem);
    const float* Cs = reinterpret_cast<const float*>(smem);
    if (g.ksplit > 1) {
        for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
            const int r = idx / BN, c = idx % BN;
            const int gr = m0 + r, gc = n0 + c;
            if (gr >= M || gc >= N) continue;
            float* dst = nullptr;
            if (band <= 0) dst = C + (size_t)(rowmap != nullptr ? rowmap[gr] : gr) * ldc + gc;
            else if (gr < band_rows) dst = C + (size_t)gr * ldc + gc;
            else if (gr >= band && gr < band + band_rows) dst = C2 + (size_t)(gr - band) * ldc + gc;
            if (dst != nullptr) atomicAdd(dst, Cs[r * G::CLD + c]);
        }
        return;
    }
    // (applying Adam to the finished tile right here -- no gradient round trip through HBM, no optimiser launch -- was
    // built and measured: 0.190 vs 0.166 ms/step.  1-2 workgroups per CU cannot keep enough bytes in flight for a streaming
    // epilogue, and the two fused launches contend for HBM exactly when the critical path needs it.)
    for (int idx = threadIdx.x; idx < BM * BN; idx += HL_THREADS) {
        const int r = idx / BN, c = idx % BN;
        const int gr = m0 + r, gc = n0 + c;
        if (gr >= M || gc >= N) continue;
        if (band <= 0) {
            C[(size_t)(rowmap != nullptr ? rowmap[gr] : gr) * ldc + gc] = Cs[r * G::CLD + c];
        } else {
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
                                                            int ksteps_per_split, int tiles_m, int tiles_n, int S, unsigned long long* stamp) {
    HL_STAMP_T0(stamp);
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
    }    HL_STAMP_END(stamp);
}

// split-K partial products on the LDS-DMA core (K % 64 == 0)
template <int BM, int BN, int WM, int WN, int NBUF>
__global__ __launch_bounds__(HL_THREADS) void k_gemm_splitk_dma(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                                int ldb, float* __restrict__ slab, int ldn, int M, int N, int K,
                                                                int ksteps_per_split, int tiles_m, int tiles_n, int S, unsigned long long* stamp) {
    HL_STAMP_T0(stamp);
    using G = GemmDMA<BM, BN, WM, WN, NBUF>;
    __shared__ __attribute__((aligned(1024))) char smem[G::SMEM_BYTES];
    const int tiles = tiles_m * tiles_n;
    const int lid = xcd_remap(blockIdx.x, tiles * S);
    const int s = lid / tiles, tl = lid % tiles;
    const int m0 = (tl / tiles_n) * BM, n0 = (tl % tiles_n) * BN;
    const int kb = s * ksteps_per_split * 64;
    int ke = kb + ksteps_per_split * 64;
    if (ke > K) ke = K;
    typename G::Acc acc;
    G::zero(acc);
    G::run(A, lda, B, ldb, m0, n0, M, N, kb, ke, smem, acc);
    G::to_lds(acc, smem);
    const float* Cs = reinterpret_cast<const float*>(smem);
    float* out = slab + (size_t)s * M * ldn;
    // one float4 per lane (the slab rows are 16-byte aligned: ldn % 4 == 0 is checked by the launcher)
    constexpr int C4 = BN / 4;
    for (int idx = threadIdx.x; idx < BM * C4; idx += HL_THREADS) {
        const int r = idx / C4, c = (idx % C4) * 4;
        if (m0 + r < M && n0 + c < N) {
            const float4 v = make_float4(Cs[r * G::CLD + c], Cs[r * G::CLD + c + 1], Cs[r * G::CLD + c + 2], Cs[r * G::CLD + c + 3]);
            if (n0 + c + 4 <= N) *reinterpret_cast<float4*>(out + (size_t)(m0 + r) * ldn + n0 + c) = v;
            else {
                const float e[4] = {v.x, v.y, v.z, v.w};
                for (int t = 0; t < 4 && n0 + c + t < N; ++t) out[(size_t)(m0 + r) * ldn + n0 + c + t] = e[t];
            }
        }
    }    HL_STAMP_END(stamp);
}

// ------------------------------------------------------------------------------------------------
// GEMM + elementwise epilogue into dual bf16.  MODE 0: relu(acc + bias)   (decoder trunk, HLVAE.py:336)
//                                             MODE 1: acc * (ref > 0)    (d trunk of the encoder), colsum
//                                             MODE 2: acc + bias         (y_layer under conv, HLVAE.py:337)
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
    // 1-D grid, XCD-aware: the row blocks that share one 64-column panel of the weight get consecutive logical ids and
    // therefore one L2 (2-D round-robin placement: every XCD pulled every panel, PMC 24.6 MB for a 2.7 MB weight)
    const int tiles_m = M / 64;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lid % tiles_m) * 64, n0 = (lid / tiles_m) * 64;
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
            } else if (MODE == 2) {
                v += bias[gc];                                   // plain Linear (y_layer of the convolutional decoder)
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
// host launchers (called from cabi.hip)
// ------------------------------------------------------------------------------------------------
// slices of K for a weight-gradient GEMM with `tiles` 64 x 64 output tiles: none when the tiles alone fill the chip
int hl_wgrad_ksplit(long tiles, int K) {
    int ks = 1;
    while (tiles * ks < 192 && K / (2 * ks) >= 256 && ks < 16) ks *= 2;
    return ks;
}

int hl_launch_gemm_f32(const bf16_t* A, int lda, const bf16_t* B, int ldb, float* C, int ldc, int M, int N, int K,
                       int band, int band_rows, float* C2, const char* label, hipStream_t s, const int32_t* rowmap = nullptr) {
    HL_REQUIRE(K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0, HLVAE_ESHAPE, "gemm_f32: K=%d lda=%d ldb=%d", K, lda, ldb);
    const int n_fast = M >= N;      // A is the larger operand: its row panel is reused by consecutive ids
    // split-K: few 64 x 64 output tiles and a long K (the batch axis of a weight gradient)
    const int ksplit = band <= 0 ? hl_wgrad_ksplit((long)((M + 63) / 64) * ((N + 63) / 64), K) : 1;
    const int kper = ksplit > 1 ? ru((K + ksplit - 1) / ksplit, 64) : K;
    if (ksplit > 1) {
        HL_REQUIRE(ldc == N, HLVAE_ESHAPE, "gemm_f32: split-K needs a dense C");
        HL_CHECK(hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * N, s));
    }
    HL_PROF(label, s);
#define HL_GO(BMv, BNv, BKv, WMv, WNv)                                                                          \
    {                                                                                                           \
        const int tm = (M + BMv - 1) / BMv, tn = (N + BNv - 1) / BNv;                                           \
        k_gemm_f32<BMv, BNv, BKv, WMv, WNv><<<tm * tn * ((K + kper - 1) / kper), HL_THREADS, 0, s>>>(A, lda, B, ldb, C, ldc, M, N, K, \
                                                                          band, band_rows, C2, tm, tn, n_fast, kper, rowmap); \
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

// can a weight gradient [M][N] over K batch rows take its optimiser step in the epilogue (k_gemm_adam)?  16-byte rows, whole
// 4-row groups for the transposed shadow, and enough tiles that the batch axis is not sliced (split-K adds with atomics)
bool hl_gemm_adam_ok(int M, int N, int K, bool may_be_small) {
    return N % 4 == 0 && M % 4 == 0 && K % 32 == 0 &&
           (may_be_small || hl_wgrad_ksplit((long)((M + 63) / 64) * ((N + 63) / 64), K) == 1);
}

int g_hl_adam_stagger = -1;
static int hl_adam_stagger() {       // HL_ADAM_STAGGER=0 switches the phase stagger off (A/B)
    if (g_hl_adam_stagger < 0) { const char* e = getenv("HL_ADAM_STAGGER"); g_hl_adam_stagger = (e != nullptr && e[0] == '0') ? 0 : 1; }
    return g_hl_adam_stagger;
}
// tile of the fused gradient + optimiser launches: 64 x 64 (32 x 64, 64 x 32, 32 x 32 and -- round 3, with 512-byte row segments
// of master / m / v -- 32 x 128 were measured: none is faster alone, 32 x 128 is 7 us slower inside the step)
static void hl_adam_tile_shape(int K, int& bm, int& bn) { (void)K; bm = bn = 64; }

// workgroups of the launch (what the completion tickets count)
int hl_gemm_adam_grid(const AdamGemmGroup& g) {
    int bm, bn;
    hl_adam_tile_shape(g.K, bm, bn);
    int t = 0;
    for (int i = 0; i < g.n; ++i) t += ((g.p[i].M + bm - 1) / bm) * ((g.p[i].N + bn - 1) / bn);
    return t;
}

int hl_launch_gemm_adam(AdamGemmGroup g, float* P, float* M1, float* M2, int64_t* step_count, float lr, float b1, float b2, float eps,
                        float gscale, unsigned ticket_total, const char* label, hipStream_t s, float* Gflat, long flat_lo, long flat_n,
                        unsigned long long* tick_shards) {
    // tick_shards: this launch's shard counters (common.h hl_take_ticket; ticket_total then counts shard units over the step's
    // launches), or nullptr: one-level tickets (ticket_total = workgroups)
    HL_REQUIRE(g.n >= 1 && g.n <= 3 && g.K % 32 == 0, HLVAE_ESHAPE, "gemm_adam: n=%d K=%d", g.n, g.K);
    HL_REQUIRE(flat_lo % 4 == 0 && flat_n % 4 == 0 && flat_n >= 0 && (flat_n == 0 || Gflat != nullptr), HLVAE_ESHAPE,
               "gemm_adam: flat range [%ld, +%ld) must be 4-aligned", flat_lo, flat_n);
    int bm, bn;
    hl_adam_tile_shape(g.K, bm, bn);
    int t = 0;
    for (int i = 0; i < g.n; ++i) {
        AdamGemmProb& q = g.p[i];
        HL_REQUIRE(q.N % 4 == 0 && q.M % 4 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldd % 4 == 0 && q.off % 4 == 0 && q.off2 % 4 == 0 &&
                       (q.shT == nullptr || q.ldT % 4 == 0), HLVAE_ESHAPE, "gemm_adam problem %d: M=%d N=%d lda=%d ldb=%d", i, q.M, q.N,
                   q.lda, q.ldb);
        HL_REQUIRE(q.off + (long)q.M * q.N < (1l << 31) && q.off2 + (long)q.M * q.N < (1l << 31), HLVAE_ESHAPE,
                   "gemm_adam problem %d: arena offsets beyond 2^31 elements", i);          // the kernel keeps them in 32 bits
        q.tiles_m = (q.M + bm - 1) / bm;
        q.tiles_n = (q.N + bn - 1) / bn;
        q.tile0 = t;
        t += q.tiles_m * q.tiles_n;
    }
    g.tiles_total = t;
    HL_PROF(label, s);
    const int grid = t;
    unsigned long long* stamp = hl_stamp_slot(g.n == 1 ? HL_ST_ADAM_WY : HL_ST_ADAM_REST);
#define HL_GA(BMv, BNv, MINWv) k_gemm_adam_dma<BMv, BNv, 2, MINWv><<<grid, HL_THREADS, 0, s>>>(g, P, M1, M2, step_count, lr, b1, b2, eps, gscale, ticket_total, Gflat, flat_lo / 4, flat_n / 4, stamp, hl_adam_stagger(), tick_shards)
    if (g.K % 64 == 0 && hl_use_dma()) {
        HL_GA(64, 64, 4);
    } else if (g.K % 64 == 0)
        k_gemm_adam<64><<<grid, HL_THREADS, 0, s>>>(g, P, M1, M2, step_count, lr, b1, b2, eps, gscale, ticket_total, Gflat, flat_lo / 4, flat_n / 4, stamp, tick_shards);
    else
        k_gemm_adam<32><<<grid, HL_THREADS, 0, s>>>(g, P, M1, M2, step_count, lr, b1, b2, eps, gscale, ticket_total, Gflat, flat_lo / 4, flat_n / 4, stamp, tick_shards);
#undef HL_GA
    HL_LAUNCH_CHECK();
    return 0;
}

// grouped launch: every problem uses the tile shape chosen for problem 0 (the large one)
int hl_launch_gemm_f32_group(GemmGroup g, const char* label, hipStream_t s) {
    HL_REQUIRE(g.n >= 1 && g.n <= 4 && g.K % 32 == 0, HLVAE_ESHAPE, "gemm group: n=%d K=%d", g.n, g.K);
    // g.ksplit (set by the caller, who clears the outputs when it is > 1) slices K for every problem of the group
    if (g.ksplit < 1) g.ksplit = 1;
    g.kper = g.ksplit > 1 ? ru((g.K + g.ksplit - 1) / g.ksplit, 64) : g.K;
    const int slices = (g.K + g.kper - 1) / g.kper;
    const bool big = !(g.p[0].M <= 64 || (long)((g.p[0].M + 127) / 128) * ((g.p[0].N + 63) / 64) < 192);
    const int BMv = big ? 128 : 64;
    int t = 0;
    for (int i = 0; i < g.n; ++i) {
        GemmProb& q = g.p[i];
        HL_REQUIRE(q.lda % 8 == 0 && q.ldb % 8 == 0, HLVAE_ESHAPE, "gemm group: lda=%d ldb=%d", q.lda, q.ldb);
        q.tiles_m = (q.M + BMv - 1) / BMv;
        q.tiles_n = (q.N + 63) / 64;
        q.n_fast = q.M >= q.N;
        q.tile0 = t;
        t += q.tiles_m * q.tiles_n;
    }
    g.tiles_total = t;
    t *= slices;
    HL_PROF(label, s);
    if (big) {
        if (g.K % 64 == 0) k_gemm_f32_group<128, 64, 64, 2, 2><<<t, HL_THREADS, 0, s>>>(g);
        else k_gemm_f32_group<128, 64, 32, 2, 2><<<t, HL_THREADS, 0, s>>>(g);
    } else {
        if (g.K % 64 == 0) k_gemm_f32_group<64, 64, 64, 2, 2><<<t, HL_THREADS, 0, s>>>(g);
        else k_gemm_f32_group<64, 64, 32, 2, 2><<<t, HL_THREADS, 0, s>>>(g);
    }
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
    unsigned long long* stamp = label[0] == 'e' ? hl_stamp_slot(HL_ST_ENC1) : (label[0] == 'd' && label[1] == 'U' && label[2] == '_' ? hl_stamp_slot(HL_ST_DU) : nullptr);
    if (hl_use_dma() && ldn % 4 == 0)
        k_gemm_splitk_dma<64, 64, 2, 2, 3><<<tm * tn * S, HL_THREADS, 0, s>>>(A, lda, B, ldb, slab, ldn, M, N, K, per, tm, tn, S, stamp);
    else
        k_gemm_splitk<64, 64, 64, 2, 2><<<tm * tn * S, HL_THREADS, 0, s>>>(A, lda, B, ldb, slab, ldn, M, N, K, per, tm, tn, S, stamp);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_gemm_act(int mode, const bf16_t* A, int lda, const bf16_t* Bm, int ldb, int M, int N, int K,
                       const float* bias, int nvalid, const bf16_t* ref, bf16_t* out, int ldo, bf16_t* outT, int ldT,
                       int B, float* gbias, const char* label, hipStream_t s) {
    HL_REQUIRE(K % 32 == 0 && M % 64 == 0 && N % 64 == 0, HLVAE_ESHAPE, "gemm_act: M=%d N=%d K=%d", M, N, K);
    const int grid = (N / 64) * (M / 64);
    HL_PROF(label, s);
    if (mode == 2) {
        if (K % 64 == 0)
            k_gemm_act<64, 2><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
        else
            k_gemm_act<32, 2><<<grid, HL_THREADS, 0, s>>>(A, lda, Bm, ldb, M, N, K, bias, nvalid, ref, out, ldo, outT, ldT, B, gbias);
    } else if (K % 64 == 0) {
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

