// bf16 "NT" GEMM main loop for gfx950, operands staged by LDS-DMA (global_load_lds_dwordx4):
//     C[BM x BN] += A[m0.., k] * B[n0.., k]^T          A, B row-major with K contiguous, BK = 64
//
// Round 3 core (the register-staged core of rounds 1-2 is gemm_nt.h; it stays for K % 64 != 0 and for the kernels not yet moved).
// What changed and why:
//   * operands go global -> LDS directly: no staging VGPRs (the three register stages of gemm_nt.h cost 48 of a lane's
//     registers), no ds_write instructions; the registers go to the epilogues (k_gemm_adam prefetches ALL of its optimiser
//     state under the product) or to larger wave tiles;
//   * the LDS image is what the DMA leaves: 128-byte rows (64 bf16), no padding possible (destination = wave-uniform base +
//     lane * 16).  Bank conflicts of the 16-byte fragment reads are removed by an XOR swizzle of the 16-byte chunk index with
//     (row >> 1) & 7, applied to the per-lane SOURCE address when the tile is requested and to the read address (the same
//     involution on both sides, cdna_hip_programming.md rule 21): every 16-lane service group of a ds_read_b128 covers the 16
//     slots of a 256-byte bank row exactly once;
//   * NBUF LDS buffers, one raw s_barrier per k-step, counted vmcnt: NBUF - 1 tiles stay in flight across the barrier;
//   * wave tiles up to 64 x 64 (FM = FN = 4: 8 fragment reads per 16 MFMAs instead of 4 per 4) where the tile counts allow.
//
// Fragment maps (cdna_hip_programming.md section 3), v_mfma_f32_16x16x32_bf16:
//   A/B operand: lane l holds row (l & 15), k = 8 (l >> 4) + j, j = 0..7  -> one 16-byte chunk: chunk index 4 kk + (l >> 4)
//   C/D        : lane l, register r -> row 4 (l >> 4) + r, col (l & 15)
#pragma once
#include "common.h"

#define HL_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define HL_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// s_waitcnt vmcnt(N) with a compile-time N (inline asm: hipcc's own counted waits do not see LDS-DMA tiles as separate events)
template <int N>
__device__ __forceinline__ void hl_wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BM, int BN, int WM, int WN, int NBUF = 3, int CLDV = 0>
struct GemmDMA {
    static constexpr int NW = WM * WN;                       // waves per workgroup
    static constexpr int NT = NW * 64;
    static constexpr int BK = 64, ROWB = 128;                // bytes per LDS row
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tile");
    static constexpr int TM = BM / WM, TN = BN / WN;
    static_assert(TM % 16 == 0 && TN % 16 == 0, "wave tile");
    static constexpr int FM = TM / 16, FN = TN / 16;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "8-row DMA pieces");
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static constexpr int NINST = (BM + BN) / 8;              // wave-instructions (8 rows x 128 B = 1 KB each) per k-tile
    static constexpr int IPW = (NINST + NW - 1) / NW;        // per wave; every wave issues the SAME number (vmcnt counts them): the
    static constexpr int STAGE = IPW * NW * 1024;            // pieces past the tile re-fetch its last rows into slack behind it
    static constexpr int AB_BYTES = NBUF * STAGE;
    static constexpr int CLD = CLDV ? CLDV : BN + 1;
    static constexpr int C_BYTES = BM * CLD * 4;
    static constexpr int SMEM_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    static_assert(NBUF >= 2 && (NBUF - 1) * IPW < 64, "tiles in flight must fit vmcnt");

    typedef f32x4_t Acc[FM][FN];
    __device__ static __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // per-lane source pointers of this wave's IPW instructions (k-independent part); rows past the operand's end are clamped
    // to its last row (their products land in C rows / columns the epilogues never use)
    struct Src {
        const bf16_t* p[IPW];
    };
    __device__ static __forceinline__ void src_init(Src& s, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B,
                                                    int ldb, int m0, int n0, int M, int N, int k_begin, int wave, int lane) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int q = wave + NW * j;                     // instruction index: rows 8 q .. 8 q + 7 of the stacked [A; B] tile
            const int rt = 8 * q + (lane >> 3);
            const bool isA = 8 * q < BM;                     // (wave-uniform: 8 | BM)
            const int r = isA ? rt : min(rt - BM, BN - 1);   // (padding pieces, q >= NINST: the tile's last row again)
            const int c = (lane & 7) ^ ((r >> 1) & 7);       // chunk this lane fetches for LDS chunk position (lane & 7)
            s.p[j] = isA ? A + (size_t)min(m0 + r, M - 1) * lda + k_begin + c * 8
                         : B + (size_t)min(n0 + r, N - 1) * ldb + k_begin + c * 8;
        }
    }
    __device__ static __forceinline__ void issue(const Src& s, int koff, char* buf, int wave) {
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int q = wave + NW * j;
            __builtin_amdgcn_global_load_lds(HL_GPTR(s.p[j] + koff), HL_LPTR(buf + q * 1024), 16, 0, 0);
        }
    }
    __device__ static __forceinline__ void compute(const char* As, Acc& acc, int lane, int wm, int wn) {
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[FM], bfr[FN];
            const int c = kk * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int r = wm * TM + i * 16 + (lane & 15);
                af[i] = *reinterpret_cast<const bf16x8_t*>(As + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int r = wn * TN + j * 16 + (lane & 15);
                bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    }

    // EXTRA = vector-memory operations the caller has issued BEFORE run() and wants left in flight through the whole main loop:
    // impossible -- vmcnt retires in order, so the first tile wait also waits for everything older.  Callers that prefetch
    // (k_gemm_adam: its optimiser state) simply issue first; the wait for tile 0 then covers them, and the product runs while
    // the OTHER workgroups of the CU stream.
    // K range [k_begin, k_end) must be a multiple of 64 and in bounds.  Ends on a barrier (the LDS tiles may be reused).
    struct NoHook { __device__ __forceinline__ void operator()() const {} };
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int m0,
                                               int n0, int M, int N, int k_begin, int k_end, char* smem, Acc& acc) {
        run(A, lda, B, ldb, m0, n0, M, N, k_begin, k_end, smem, acc, NoHook(), NoHook());
    }
    // first_tile: called once behind the first tile's wait and barrier -- every vector-memory operation the caller issued
    // BEFORE run() has completed by then (vmcnt retires in order), so an epilogue can park prefetched values in LDS there
    // without a wait of its own; second_tile: the same one k-step later (for loads first_tile itself issued).
    template <class Hook, class Hook2>
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int m0,
                                               int n0, int M, int N, int k_begin, int k_end, char* smem, Acc& acc, Hook first_tile,
                                               Hook2 second_tile) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int wm = wave / WN, wn = wave % WN;
        const int nk = (k_end - k_begin) / BK;
        if (nk <= 0) { first_tile(); second_tile(); return; }
        Src s;
        src_init(s, A, lda, B, ldb, m0, n0, M, N, k_begin, wave, lane);
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t)
            if (t < nk) issue(s, t * BK, smem + t * STAGE, wave);
        int cur = 0;                                         // buffer of tile kt
        for (int kt = 0; kt < nk; ++kt) {
            // my DMA pieces of tile kt have landed when at most the pieces of the younger tiles (<= NBUF - 2 of them) are pending
            const int younger = min(NBUF - 2, nk - 1 - kt);
            if (NBUF >= 4 && younger >= 2) hl_wait_vm<(NBUF >= 4 ? 2 : 0) * IPW>();
            else if (NBUF >= 3 && younger >= 1) hl_wait_vm<(NBUF >= 3 ? 1 : 0) * IPW>();
            else hl_wait_vm<0>();
            __builtin_amdgcn_s_barrier();                    // everyone's pieces of tile kt are in LDS; everyone is done with tile kt - 1
            const int nt = kt + NBUF - 1;                    // refill the buffer tile kt - 1 occupied
            int nb = cur + NBUF - 1;
            nb = nb >= NBUF ? nb - NBUF : nb;
            if (nt < nk) issue(s, nt * BK, smem + nb * STAGE, wave);
            if (kt == 0) first_tile();
            if (kt == 1 || (kt == 0 && nk == 1)) second_tile();
            compute(smem + cur * STAGE, acc, lane, wm, wn);
            cur = cur + 1 == NBUF ? 0 : cur + 1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // accumulators -> fp32 tile Cs[BM][CLD] in LDS (aliases the operand tiles: run() ends on a barrier)
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

// Variant for tall-thin wave layouts (every wave owns 16 rows and ALL BN columns: the head kernel): only the B operand goes
// through LDS.  A wave's A fragments are nobody else's -- lane l needs row (l & 15), k = 8 (l >> 4) + j of its own 16 rows,
// one 16-byte global load per 32 k -- so they are loaded straight into registers, NBUF stages deep, and the LDS the A tile
// occupied buys B stages instead: with BN = 80 a stage is 10 KB and FOUR fit where two 18 KB [A; B] stages did, i.e. three
// k-tiles in flight instead of one (the head kernel's main loop was eight exposed memory latencies: 1850 clocks per k-step
// for 160 clocks of MFMA work).  Same barrier / vmcnt discipline as GemmDMA; a wave's operations per k-tile are IPW DMA
// pieces + 2 register loads, in that order.
template <int BM, int BN, int NBUF = 4, int CLDV = 0>
struct GemmDMAB {
    static constexpr int NW = BM / 16;
    static constexpr int NT = NW * 64;
    static constexpr int BK = 64, ROWB = 128;
    static_assert(BM % 16 == 0 && BN % 16 == 0, "tile");
    static constexpr int FN = BN / 16;
    static constexpr int NINST = BN / 8;                     // 1 KB DMA pieces of a B k-tile
    static constexpr int IPW = (NINST + NW - 1) / NW;        // per wave (every wave the same count); pieces past the tile repeat
    static constexpr int STAGE = NINST * 1024;               // its LAST piece (same bytes to the same place): no slack needed
    static constexpr int OPS = IPW + 2;                      // vector-memory operations of one wave per k-tile
    static constexpr int AB_BYTES = NBUF * STAGE;
    static constexpr int CLD = CLDV ? CLDV : BN + 1;
    static constexpr int C_BYTES = BM * CLD * 4;
    static constexpr int SMEM_BYTES = AB_BYTES > C_BYTES ? AB_BYTES : C_BYTES;
    static_assert(NBUF >= 2 && NBUF <= 4 && (NBUF - 1) * OPS < 64, "tiles in flight must fit vmcnt");

    typedef f32x4_t Acc[1][FN];
    __device__ static __forceinline__ void zero(Acc& acc) {
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[0][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    struct AStage { u32x4_t f[2]; };
    // The A loads are inline assembly: with compiler-visible loads hipcc puts s_waitcnt vmcnt(0) in front of every k-step's first
    // MFMA (its scoreboard cannot count through the k loop), i.e. one exposed memory latency per k-step again.  The counted
    // waits of the k-steps cover them.  Everything is straight-line per k-step (no conditional request, one static wait count):
    // a load whose destination registers meet a compiler-made copy before the wait would hand stale data on, so the loop is
    // shaped to give the register allocator no reason for one -- check the disassembly when touching this (no v_mov of a stage
    // register between its global_load_dwordx4 and the s_waitcnt that covers it).
    __device__ static __forceinline__ void load_a(AStage& st, const bf16_t* p) {
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(st.f[0]) : "v"(p) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(st.f[1]) : "v"(p) : "memory");
    }

    // K range: a multiple of NBUF * 64 (the caller's launch rule), so that k-tile kt lives in LDS buffer and register stage
    // kt % NBUF with static indices everywhere.  first_tile runs behind the first tile's wait and barrier and BEFORE the next
    // tile is requested: register loads the caller issued ahead of run() have landed (vmcnt retires in order), and the
    // s_waitcnt vmcnt(0) hipcc puts in front of its uses drains only tiles 1 .. NBUF - 2, requested together with tile 0.
    template <class Hook, class Hook2>
    __device__ static __forceinline__ void run(const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb, int m0,
                                               int n0, int M, int N, int k_begin, int k_end, char* smem, Acc& acc, Hook first_tile,
                                               Hook2) {
        const int tid = threadIdx.x, lane = tid & 63;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int nk = (k_end - k_begin) / BK;
        // B pieces of this wave: rows 8 q .. 8 q + 7 of the tile, q = wave + NW j (clamped to the last piece)
        const bf16_t* bp[IPW];
        int bq[IPW];
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            const int q = min(wave + NW * j, NINST - 1);
            const int r = 8 * q + (lane >> 3);
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            bp[j] = B + (size_t)min(n0 + r, N - 1) * ldb + k_begin + c * 8;
            bq[j] = q;
        }
        const bf16_t* ap = A + (size_t)min(m0 + wave * 16 + (lane & 15), M - 1) * lda + k_begin + (lane >> 4) * 8;
        AStage as[NBUF];
        auto issue = [&](int t, int buf, AStage& st) {
#pragma unroll
            for (int j = 0; j < IPW; ++j)
                __builtin_amdgcn_global_load_lds(HL_GPTR(bp[j] + t * BK), HL_LPTR(smem + buf * STAGE + bq[j] * 1024), 16, 0, 0);
            load_a(st, ap + t * BK);
        };
        auto compute = [&](const char* Bs, const AStage& st) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t bfr[FN];
                const bf16x8_t af = __builtin_bit_cast(bf16x8_t, st.f[kk]);
                const int c = kk * 4 + (lane >> 4);
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    const int r = j * 16 + (lane & 15);
                    bfr[j] = *reinterpret_cast<const bf16x8_t*>(Bs + r * ROWB + ((c ^ ((r >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[0][j], 0, 0, 0);
            }
        };
#pragma unroll
        for (int t = 0; t < NBUF - 1; ++t) issue(t, t, as[t]);
        hl_wait_vm<(NBUF - 2) * OPS>();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);       // (keeps the hook's uses of the caller's loads, and the wait they bring, down here)
        first_tile();
        issue(NBUF - 1, NBUF - 1, as[NBUF - 1]);
        compute(smem, as[0]);
        for (int kt0 = 1; kt0 + NBUF - 1 < nk; kt0 += NBUF) {       // k-steps 1 .. nk - NBUF: every one requests tile kt + NBUF - 1
#pragma unroll
            for (int j = 0; j < NBUF; ++j) {
                const int st = (1 + j) % NBUF, jn = (st + NBUF - 1) % NBUF;      // (static after unrolling)
                hl_wait_vm<(NBUF - 2) * OPS>();
                __builtin_amdgcn_s_barrier();
                issue(kt0 + j + NBUF - 1, jn, as[jn]);
                compute(smem + st * STAGE, as[st]);
            }
        }
#pragma unroll
        for (int j = 1; j < NBUF; ++j) {                             // the last NBUF - 1 k-steps: nothing left to request
            if (j == NBUF - 1) hl_wait_vm<0>();
            else if (j == NBUF - 2) hl_wait_vm<(NBUF >= 3 ? 1 : 0) * OPS>();
            else hl_wait_vm<(NBUF >= 4 ? 2 : 0) * OPS>();
            __builtin_amdgcn_s_barrier();
            compute(smem + j * STAGE, as[j]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    __device__ static __forceinline__ void to_lds(const Acc& acc, char* smem) {
        float* Cs = reinterpret_cast<float*>(smem);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(wave * 16 + (lane >> 4) * 4 + r) * CLD + j * 16 + (lane & 15)] = acc[0][j][r];
        __syncthreads();
    }
};
