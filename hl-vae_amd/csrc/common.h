// Shared device helpers for the gfx950 HL-VAE kernels (wave = 64 lanes, hard-coded).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hlvae_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // one 16x16 accumulator fragment
typedef uint16_t bf16_t;                                        // storage type of bf16 buffers

#define HL_THREADS 256
#define HL_STAT_CHUNKS 16   // row chunks of the batch statistics (partial sums instead of atomics)
#define HL_WAVE 64

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;   // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
    return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// torch.nn.functional.softplus (beta = 1, threshold = 20)
__device__ __forceinline__ float softplus_f(float t) { return t > 20.f ? t : log1pf(__expf(t)); }
__device__ __forceinline__ float sigmoid_f(float t) { return 1.f / (1.f + __expf(-t)); }

static inline int ru(int v, int m) { return (v + m - 1) / m * m; }

// XCD-aware block id: MI355X deals workgroups round-robin over its 8 XCDs (block b and b + 8 share an XCD and
// its private 4 MiB L2).  This bijective remap gives every XCD a CONTIGUOUS range of logical ids, so that tiles
// which share an operand panel (consecutive logical ids) hit the same L2 instead of pulling the panel through
// the fabric eight times.  Speed only: results never depend on the placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// one product C[M][N] (fp32, row stride ldc) = A[M][K] B[N][K]^T of a grouped launch (dense.hip: k_gemm_f32_group).
// band > 0: rows [0, band_rows) -> C, rows [band, band + band_rows) -> C2 (stacked [mu; log_var] operand).
// tiles_* / n_fast / tile0 are filled by the launcher.
struct GemmProb {
    const bf16_t* A;
    const bf16_t* B;
    float* C;
    float* C2;
    int lda, ldb, ldc, M, N, band, band_rows, tiles_m, tiles_n, n_fast, tile0;
    const int32_t* rowmap;      // band <= 0 only: output row gr is stored at row rowmap[gr] of C (nullptr: gr)
};
struct GemmGroup {
    GemmProb p[4];
    int n, K;
    int ksplit, kper, tiles_total;      // split-K over the whole group (launcher fills kper / tiles_total)
};

// weight-gradient products whose epilogue applies the optimiser step (dense.hip: k_gemm_adam): g = A B^T [M][N] stays in LDS.
// Tile row gr belongs to master row rowmap[gr] (or gr) of the fp32 matrix at arena offset `off`; band > 0: rows [0, band_rows)
// -> matrix at `off`, rows [band, band + band_rows) -> matrix at `off2`, others dropped (stacked [mu; log_var] operand).
// The bf16 shadows take the NEW values at the tile's own row index: sh [..][ldd] row-major, shT [..][ldT] transposed (or nullptr).
struct AdamGemmProb {
    const bf16_t* A;
    const bf16_t* B;
    const int32_t* rowmap;
    bf16_t* sh;
    bf16_t* shT;
    long off, off2;
    int lda, ldb, M, N, band, band_rows, ldd, ldT, tiles_m, tiles_n, tile0;
};
struct AdamGemmGroup {
    AdamGemmProb p[3];
    int n, K, tiles_total;
};

enum { HL_PEND_METRICS = 1, HL_PEND_FINALIZE = 2, HL_PEND_RUNNING = 4, HL_PEND_FEED = 8, HL_PEND_RUNNING2 = 16,
       HL_PEND_DEFERRED = HL_PEND_METRICS | HL_PEND_FINALIZE | HL_PEND_FEED };

// a + (the value of the lane 16 / 32 positions away, lane ^ 16 / lane ^ 32): the cross-row steps of a wave reduction on the
// VALU (gfx950 v_permlane16_swap / v_permlane32_swap: a half exchange between two registers) instead of ds_bpermute through
// the LDS crossbar.  swap(a, a) leaves {own, partner} in the two results on every lane, so their sum is the all-reduce step.
__device__ __forceinline__ float xor16_sum(float a) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float a) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// ---- completion tickets of an optimiser step --------------------------------------------------------------------------------
// The launches that apply one optimiser step may run concurrently on different streams; the workgroup that finishes LAST, over all
// of them, commits the device-side step number (step_count[0] += 1).  Rounds 1-2: every workgroup took a returning atomic on ONE
// word (step_count[1]).  In a single-round streaming launch all ~800 workgroups reach that line within a microsecond or two and
// the atomics serialise at the memory side (~12-20 ns each): the LAST return -- the kernel's end -- came 10-17 us after the last
// store (round-3 micro-benchmark: 25 -> 43 us for y_layer's launch with tickets on).  Two levels now: a workgroup takes its ticket
// on one of 32 shard counters of ITS launch (own 64-byte lines, library-owned memory, <= ~26 arrivals each); whoever completes a
// shard takes one ticket on step_count[1], whose expected total is the number of shards with arrivals over all launches of the
// step (hl_ticket_units per launch).  shards == nullptr: the one-level protocol (units_total = workgroups).
#define HL_TICK_SHARDS 32
#define HL_TICK_WORDS (8 * HL_TICK_SHARDS)                  // uint64 words per launch slot
#define HL_TICK_SLOTS 4                                     // launch slots per plan
static inline unsigned hl_ticket_units(int grid) { return (unsigned)(grid < HL_TICK_SHARDS ? grid : HL_TICK_SHARDS); }
// thread 0 of the workgroup, behind the barrier that follows its last store
__device__ __forceinline__ void hl_take_ticket(int64_t* step_count, unsigned long long* shards, unsigned units_total) {
    if (shards != nullptr) {
        const unsigned nblk = gridDim.x * gridDim.y, b = blockIdx.x + blockIdx.y * gridDim.x, s = b & (HL_TICK_SHARDS - 1);
        const unsigned expect = (nblk - s + HL_TICK_SHARDS - 1) / HL_TICK_SHARDS;       // workgroups of this launch on shard s
        const unsigned long long got = atomicAdd(shards + 8 * s, 1ull);
        if (got != expect - 1) return;
        shards[8 * s] = 0ull;                               // (the last arrival: nobody else touches the shard before the next step)
    }
    const unsigned long long done = atomicAdd(reinterpret_cast<unsigned long long*>(step_count + 1), 1ull);
    if (done == units_total - 1) { step_count[1] = 0; step_count[0] += 1; }
}

struct hlvae_plan {
    hlvae_dims d;
    hlvae_var* vars_dev;      // [D]
    hlvae_var* vars_sorted_dev;   // [D] the same rows in the head kernel's order (grouped by kind); .pad = the variable's own index
    int32_t* wy_rowsrc_dev;       // [NYlp] row of y_layer's weight (master order) that shadow / dY row j (kernel order) holds;
                                  // nullptr: identity (convolutional model, or variables already grouped)
    int32_t* col2var_dev;     // [Xp]  variable of an expanded column, -1 in the padding
    int32_t* stat_var_dev;    // [n_stat] variable index of each statistics row
    unsigned long long* tick_dev;   // [HL_TICK_SLOTS][HL_TICK_WORDS] shard counters of the optimiser launches' completion tickets (zero between steps)
    int kmax;                 // largest class count among cat / ordinal variables (selects the head-kernel instance)
    // fork/join side streams: independent weight-gradient GEMMs run beside the critical path of the backward
    hipStream_t side[2];
    hipEvent_t ev[6];
    // deferred side work: hlvae_step_metrics and hlvae_decoder_fwd(want_grad = 2) only record their dependency (ev[5]); the
    // launches are queued on a side stream by the next hlvae_backward* / hlvae_join (cabi.hip: hl_flush_deferred)
    mutable int pend_flags;        // HL_PEND_*
    mutable int defer_join;        // hlvae_set_defer_join: hlvae_backward* return without joining the deferred side chain
    mutable hlvae_ws pend_ws, pend_fin_ws;
    mutable int pend_B, pend_fin_B;
    mutable float* pend_err;
    mutable hlvae_ws pend_feed_ws;
    mutable int pend_feed_B;
    mutable const float* pend_feed_vals;
    mutable const uint8_t* pend_feed_mask;
    mutable const int32_t* pend_feed_rows;
};

// ---- in-graph kernel stamps (bench.py's "roofline.in_step"): when the host has handed the library a stamp buffer
// (hlvae_stamp_buffer) BEFORE the step was launched / captured, the stamped kernels record the first start and the last end of
// their workgroups on the 100 MHz s_memrealtime clock -- inside the replayed HIP graph, beside whatever else runs then, with no
// extra graph node.  Layout: kernel k owns HL_STAMP_SUB sub-slots of 8 words (one 64-byte line each) at buf[8 HL_STAMP_SUB k]:
// word 0 = min start (armed by the host with ~0; sub-slot 0's word 0 = 0 disarms the kernel: nothing is recorded and the cost is
// one load per workgroup), word 1 = max end; a workgroup uses sub-slot (block id mod HL_STAMP_SUB), the host folds them.
enum { HL_ST_ENC1 = 0, HL_ST_MID_FWD, HL_ST_HEADS, HL_ST_DU, HL_ST_MID_BWD, HL_ST_ADAM_REST, HL_ST_ADAM_WY, HL_ST_N };
unsigned long long* hl_stamp_slot(int slot);       // cabi.hip: nullptr when no buffer is set
// Cost when no buffer is set: none (uniform null check).  Buffer set but disarmed: thread 0 of the first 32 workgroups reads one
// cached word at the top, thread 0 of every workgroup one at the end.  Armed: those 32 take s_memrealtime at the top, every
// workgroup at its end + one or two atomics on its sub-slot.  (Measured the hard way, round 3: s_memrealtime executed by every wave
// at the top of k_gemm_adam -- or one uncached same-address load per workgroup there -- made that kernel 29 -> 46 us: 3264 waves
// queue on one counter / one memory channel in front of the prefetch loads the kernel lives on.)
#define HL_STAMP_SUB 32          // sub-slots per kernel, each on a 64-byte line of its own (same-address atomics from ~1000 workgroups
                                 // cost ~20 ns each and held the kernel's end back by 15-30 us); the host folds them
#define HL_STAMP_T0(st)                                                                                                          \
    unsigned long long hl_t0_ = 0ull;                                                                                            \
    if ((st) != nullptr && blockIdx.x < HL_STAMP_SUB && blockIdx.y == 0 && threadIdx.x == 0 && threadIdx.y == 0) {               \
        if (*(const volatile unsigned long long*)(st) != 0ull) hl_t0_ = (unsigned long long)__builtin_amdgcn_s_memrealtime();   \
    }
__device__ __forceinline__ void hl_stamp_commit(unsigned long long* st, unsigned long long t0) {
    if (st == nullptr || threadIdx.x != 0 || threadIdx.y != 0) return;
    if (*(const volatile unsigned long long*)st == 0ull) return;                                   // disarmed (word 0 of sub-slot 0)
    unsigned long long* sub = st + 8 * ((blockIdx.x + blockIdx.y * gridDim.x) & (HL_STAMP_SUB - 1));
    if (t0 != 0ull) atomicMin(sub, t0);
    atomicMax(sub + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
}
#define HL_STAMP_END(st) hl_stamp_commit(st, hl_t0_)

void hl_set_error(const char* fmt, ...);

// optional per-kernel HIP-event timing (hlvae_prof_enable / hlvae_prof_report); a no-op when disabled
void hl_prof_begin(const char* name, hipStream_t s);
void hl_prof_end(hipStream_t s);
struct HlProfScope {
    hipStream_t s;
    HlProfScope(const char* name, hipStream_t st) : s(st) { hl_prof_begin(name, st); }
    ~HlProfScope() { hl_prof_end(s); }
};
#define HL_PROF(name, stream) HlProfScope _hl_prof_scope(name, stream)
#define HL_CHECK(expr)                                                            \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            hl_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return (int)_e;                                                       \
        }                                                                         \
    } while (0)
#define HL_LAUNCH_CHECK() HL_CHECK(hipGetLastError())
#define HL_REQUIRE(cond, code, ...)                                               \
    do {                                                                          \
        if (!(cond)) { hl_set_error(__VA_ARGS__); return (code); }                \
    } while (0)
