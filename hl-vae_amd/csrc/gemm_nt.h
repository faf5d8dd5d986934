// bf16 "NT" GEMM main loop for gfx950:  C[BM x BN] += A[m0.., k] * B[n0.., k]^T
// Both operands are K-contiguous (row-major [rows][K]); every dense product of the HL-VAE step is
// expressed in this form (weights are torch Linear [out][in]; transposed activation copies are
// produced by the kernels that own the tile), so there is exactly one MFMA main loop to tune.
//
// 256 threads = 4 waves laid out WM x WN over the block tile; each wave owns (BM/WM) x (BN/WN)
// as FM x FN fragments of v_mfma_f32_16x16x32_bf16.  Operands are register-staged:
// global (16 B per lane) -> VGPR -> LDS with an 8-element row pad, next k-tile's global loads are
// issued before the MFMAs of the current one.
//
// Fragment maps (cdna_hip_programming.md section 3):
//   A/B operand: lane l holds row (l & 15), k = 8*(l >> 4) + j, j = 0..7
//   C/D        : lane l, register r -> row 4*(l >> 4) + r, col (l & 15)
#pragma once
#include "common.h"

template <int BM, int BN, int BK, int WM, int WN>
struct GemmNT {
    static_assert(WM * WN == 4, "4 waves per block");
    static_assert(BK == 32 || BK == 64, "BK");
    static constexpr int LDS = BK + 8;                 // bf16 elements per LDS row (16-B multiple)
    static constexpr int TM = BM / WM, TN = BN / WN;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile");
    static constexpr int FM = TM / 16, FN = TN / 16;
    static constexpr int CPR = BK / 8;                 // 16-B chunks per tile row
    static constexpr int A_CHUNKS = BM * CPR, B_CHUNKS = BN * CPR;
    static constexpr int A_IT = (A_CHUNKS + HL_THREADS - 1) / HL_THREADS;
    static constexpr int B_IT = (B_CHUNKS + HL_THREADS - 1) / HL_THREADS;
    static constexpr int AB_BYTES = (BM + BN) * LDS * 2;
    static constexpr int CLD = BN + 1;                 // fp32 C tile row stride in LDS
    static constexpr int C_BYTES = BM * CLD * 4;
    static constexpr int SMEM_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;

    typedef f32x4_t Acc[FM][FN];

    __device__ static __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // rows >= M (resp. N) read as zero; K range [k_begin, k_end) must be a multiple of BK and in bounds.
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                               int ldb, int m0, int n0, int M, int N, int k_begin, int k_end,
                                               char* smem, Acc& acc) {
        bf16_t* As = reinterpret_cast<bf16_t*>(smem);
        bf16_t* Bs = As + BM * LDS;
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        uint4 ra[A_IT], rb[B_IT];

        auto gload = [&](int k) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int c = tid + i * HL_THREADS;
                const int r = c / CPR, kc = c % CPR;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < A_CHUNKS && m0 + r < M)
                    v = *reinterpret_cast<const uint4*>(A + (size_t)(m0 + r) * lda + k + kc * 8);
                ra[i] = v;
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int c = tid + i * HL_THREADS;
                const int r = c / CPR, kc = c % CPR;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (c < B_CHUNKS && n0 + r < N)
                    v = *reinterpret_cast<const uint4*>(B + (size_t)(n0 + r) * ldb + k + kc * 8);
                rb[i] = v;
            }
        };
        auto lstore = [&]() {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int c = tid + i * HL_THREADS;
                if (c < A_CHUNKS) *reinterpret_cast<uint4*>(As + (c / CPR) * LDS + (c % CPR) * 8) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int c = tid + i * HL_THREADS;
                if (c < B_CHUNKS) *reinterpret_cast<uint4*>(Bs + (c / CPR) * LDS + (c % CPR) * 8) = rb[i];
            }
        };

        if (k_begin >= k_end) return;
        gload(k_begin);
        lstore();
        __syncthreads();
        for (int k = k_begin; k < k_end; k += BK) {
            const bool more = (k + BK) < k_end;
            if (more) gload(k + BK);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8_t af[FM], bfr[FN];
                const int ko = kk * 32 + (lane >> 4) * 8;
#pragma unroll
                for (int i = 0; i < FM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * TM + i * 16 + (lane & 15)) * LDS + ko);
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * TN + j * 16 + (lane & 15)) * LDS + ko);
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            if (more) {
                lstore();
                __syncthreads();
            }
        }
    }

    // accumulators -> fp32 tile Cs[BM][CLD] in LDS (aliases the operand tiles: the main loop ends on a barrier)
    __device__ static __forceinline__ void to_lds(const Acc& acc, char* smem) {
        float* Cs = reinterpret_cast<float*>(smem);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int wm = wave / WN, wn = wave % WN;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    Cs[(wm * TM + i * 16 + (lane >> 4) * 4 + r) * CLD + wn * TN + j * 16 + (lane & 15)] = acc[i][j][r];
        __syncthreads();
    }
};
