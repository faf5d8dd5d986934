// bf16 "NT" GEMM main loop for gfx950:  C[BM x BN] += A[m0.., k] * B[n0.., k]^T
// Both operands are K-contiguous (row-major [rows][K]); every dense product of the HL-VAE step is
// expressed in this form (weights are torch Linear [out][in]; transposed activation copies are
// produced by the kernels that own the tile), so there is exactly one MFMA main loop to tune.
//
// 256 threads = 4 waves laid out WM x WN over the block tile; each wave owns (BM/WM) x (BN/WN)
// as FM x FN fragments of v_mfma_f32_16x16x32_bf16.  Operands are register-staged:
// global (16 B per lane) -> VGPR -> LDS with an 8-element row pad; NS k-tiles are kept in flight and
// the LDS tile is double buffered (one barrier per k-step).  These GEMMs are small (M = batch = 512),
// so the loop is load-latency bound, not MFMA bound: the depth of the register pipeline is what matters.
//
// Fragment maps (cdna_hip_programming.md section 3):
//   A/B operand: lane l holds row (l & 15), k = 8*(l >> 4) + j, j = 0..7
//   C/D        : lane l, register r -> row 4*(l >> 4) + r, col (l & 15)
#pragma once
#include <type_traits>
#include "common.h"

template <int BM, int BN, int BK, int WM, int WN, int NS = 3, int CLDV = 0>
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
    // every lane loads and stores UNCONDITIONALLY (a predicated load makes hipcc branch around it and drain
    // vmcnt(0) before each LDS write, which serialises the pipeline): rows past the end of an operand are
    // clamped to its last row (their products land in C rows/columns the epilogues never use) and the LDS
    // tiles are allocated for the full 256-lane passes.
    static constexpr int BM_ALLOC = A_IT * HL_THREADS / CPR, BN_ALLOC = B_IT * HL_THREADS / CPR;
    static constexpr int STAGE_ELEMS = (BM_ALLOC + BN_ALLOC) * LDS;   // one LDS buffer (A tile then B tile), two buffers
    static constexpr int AB_BYTES = 2 * STAGE_ELEMS * 2;
    static constexpr int CLD = CLDV ? CLDV : BN + 1;   // fp32 C tile row stride in LDS (CLDV: an epilogue's own choice)
    static constexpr int C_BYTES = BM * CLD * 4;
    static constexpr int SMEM_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;

    typedef f32x4_t Acc[FM][FN];

    __device__ static __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // rows >= M (resp. N) yield unspecified C rows (columns): epilogues must not use them.
    // K range [k_begin, k_end) must be a multiple of BK and in bounds.
    //
    // Software pipeline: NS k-tiles are in flight in registers (global -> VGPR, 16 B per lane), the tile after the
    // current one is written to the OTHER LDS buffer while the current one feeds the MFMAs, one barrier per k-step.
    // hipcc emits counted vmcnt waits for its own loads, so the LDS write of tile t+1 waits only for that tile.
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;   // native vector: stays an SSA value
    struct Stage {            // one k-tile in flight in registers
        u32x4_t a[A_IT];
        u32x4_t b[B_IT];
    };
    __device__ static __forceinline__ void gload(Stage& st, const bf16_t* const (&pa)[A_IT],
                                                 const bf16_t* const (&pb)[B_IT], int kg) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) st.a[i] = *reinterpret_cast<const u32x4_t*>(pa[i] + kg);
#pragma unroll
        for (int i = 0; i < B_IT; ++i) st.b[i] = *reinterpret_cast<const u32x4_t*>(pb[i] + kg);
    }
    __device__ static __forceinline__ void lstore(const Stage& st, bf16_t* As_, int tid) {
        bf16_t* Bs_ = As_ + BM_ALLOC * LDS;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int c = tid + i * HL_THREADS;
            *reinterpret_cast<u32x4_t*>(As_ + (c / CPR) * LDS + (c % CPR) * 8) = st.a[i];
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int c = tid + i * HL_THREADS;
            *reinterpret_cast<u32x4_t*>(Bs_ + (c / CPR) * LDS + (c % CPR) * 8) = st.b[i];
        }
    }
    __device__ static __forceinline__ void compute(const bf16_t* As, Acc& acc, int lane, int wm, int wn) {
        const bf16_t* Bs = As + BM_ALLOC * LDS;
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
    }

    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                               int ldb, int m0, int n0, int M, int N, int k_begin, int k_end,
                                               char* smem, Acc& acc) {
        run(A, lda, B, ldb, m0, n0, M, N, k_begin, k_end, smem, acc, NoHook(), NoHook());
    }
    // after_prologue: called once between the issue of the first three k-tiles' global loads and the first LDS write.  An
    // epilogue can consume loads IT issued before run() there (vmcnt retires in order: they cost no extra wait, their latency
    // overlaps the prologue's) without keeping their registers alive through the main loop.
    // after_group: called once behind the first three k-steps (or behind the loop when it is shorter): whatever after_prologue
    // requested has arrived by then, so the epilogue's registers need not stay alive through the whole main loop.
    template <class Hook, class Hook2>
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                               int ldb, int m0, int n0, int M, int N, int k_begin, int k_end,
                                               char* smem, Acc& acc, Hook after_prologue, Hook2 after_group) {
        static_assert(NS == 3, "the pipeline is written out for three register stages");
        bf16_t* const sbase = reinterpret_cast<bf16_t*>(smem);
        const int tid = threadIdx.x;
        const int lane = tid & 63, wave = tid >> 6;
        const int wm = wave / WN, wn = wave % WN;
        // per-thread source pointers are k-independent
        const bf16_t* pa[A_IT];
        const bf16_t* pb[B_IT];
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int c = tid + i * HL_THREADS;
            const int r = min(m0 + c / CPR, M - 1);
            pa[i] = A + (size_t)r * lda + (c % CPR) * 8 + k_begin;
        }
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int c = tid + i * HL_THREADS;
            const int r = min(n0 + c / CPR, N - 1);
            pb[i] = B + (size_t)r * ldb + (c % CPR) * 8 + k_begin;
        }
        const int nk = (k_end - k_begin) / BK;
        if (nk <= 0) { after_prologue(); after_group(); return; }
        // three NAMED register stages (a runtime- or loop-indexed array of stages ends up in scratch memory)
        Stage s0, s1, s2;
        // (loads past the last tile are clamped to it: an unconditional, redundant load keeps the stages in VGPRs)
        const int klast = (nk - 1) * BK;
        gload(s0, pa, pb, 0);
        gload(s1, pa, pb, min(BK, klast));
        gload(s2, pa, pb, min(2 * BK, klast));
        after_prologue();
        lstore(s0, sbase, tid);
        __syncthreads();
        // step t: refill the stage consumed one step ago with tile t+3, write tile t+1 to the other LDS buffer,
        // MFMAs on tile t, one barrier
#define HL_STEP(REFILL, NEXT, T)                                                              \
        {                                                                                     \
            gload(REFILL, pa, pb, min(((T) + 3) * BK, klast));                                \
            __builtin_amdgcn_sched_barrier(0); /* keep the refill ISSUED here: hipcc otherwise sinks it */ \
            if ((T) + 1 < nk) lstore(NEXT, sbase + (((T) + 1) & 1) * STAGE_ELEMS, tid);       \
            compute(sbase + ((T) & 1) * STAGE_ELEMS, acc, lane, wm, wn);                      \
            __syncthreads();                                                                  \
        }
        int kt = 0;
        if (nk >= 3) {
            HL_STEP(s0, s1, 0)
            HL_STEP(s1, s2, 1)
            HL_STEP(s2, s0, 2)
            kt = 3;
        }
        after_group();
        for (; kt + 3 <= nk; kt += 3) {      // full groups: no conditionals around the stage registers
            HL_STEP(s0, s1, kt)
            HL_STEP(s1, s2, kt + 1)
            HL_STEP(s2, s0, kt + 2)
        }
        if (kt < nk) {
            HL_STEP(s0, s1, kt)
            if (kt + 1 < nk) HL_STEP(s1, s2, kt + 1)
        }
#undef HL_STEP
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
