// Adam on the flat fp32 arena (torch.optim.Adam semantics, reference HLVAE_main.py:277-278) fused with the
// refresh of the bf16 shadow copies that the MFMA kernels read.  HBM-bound: per parameter 4 B grad +
// 3 x (4 B read + 4 B write) state (+ 2-4 B of shadow).
//
// k_adam_tiled covers the arena in one launch:
//   tiles        the dense weight matrices in 64 x 64 tiles: the updated tile is staged in LDS and written as padded
//                bf16 in both layouts (row-major shadow + transposed shadow);
//   flat blocks  the "small" region [0, atomic_region): head parameters, biases, convolution weights, float4 per
//                lane.  They also ZERO the gradients they have consumed: that region is accumulated with atomics by the
//                next step, so no separate memset is needed.
// hlvae_backward_adam splits the step into launches that each start as soon as their gradients are final (y_layer's
// weight on a side stream under the backward pass, ...); the launches share one completion ticket.
#include "common.h"
#include "adam.h"

struct ShadowMat {
    long off;          // arena offset of the fp32 matrix [R][C] (dense)
    int R, C;
    bf16_t* dst;       // [..][ldd] row-major bf16, rows shifted by row_off
    int ldd, row_off;
    bf16_t* dstT;      // [..][ldT] transposed bf16 or nullptr
    int ldT;
    int Rcover, Ccover;   // region of dst (rows, cols) to fill, zero outside [R][C]
    const int32_t* rowsrc;   // master row that tile row r holds (y_layer: the head kernel's variable order), nullptr = r
    int tiles_c, tile0;   // tiles per row of tiles, first block index (a multiple of 8)
    int npad, remap;      // workgroups owned (tiles rounded up to a multiple of 8); XCD-contiguous tile order
};
constexpr int SHADOW_MAX = 5 + 2 * HLVAE_MAX_EXTRA;     // Wy, W1, Wd, Wmu, Wlv + the extra hidden layers of deeper trunks
struct ShadowSet {
    ShadowMat m[SHADOW_MAX];
    int n, total_tiles;
};

// update == 1: Adam step then shadows;  update == 0: shadows only (after load_state_dict / external optimiser)
// float4 per lane along the columns (C % 4 == 0 and 16-B aligned segments are checked on the host)
__global__ __launch_bounds__(HL_THREADS) void k_adam_tiled(ShadowSet set, float* __restrict__ P, const float* __restrict__ G,
                                                           float* __restrict__ M1, float* __restrict__ M2,
                                                           int64_t* __restrict__ step_count, float lr, float b1,
                                                           float b2, float eps, float gscale, int update, long n4_flat,
                                                           unsigned ticket_total, const bf16_t* __restrict__ Pb16,
                                                           long frozen_lo4, long frozen_hi4, unsigned long long* tick_shards) {
    // update: 0 = shadows only; 1 = Adam with step number step_count[0] + 1.  Workgroups past the tiles (if the launch
    // has any) update the small flat region [0, 4 n4_flat) and zero its gradients.  ticket_total = workgroups of ALL the
    // launches of this optimiser step (0: this launch takes no tickets).
    // Pb16 != nullptr (update == 0 only): the shadows are built from the flat bf16 copy of the arena that the sharded
    // optimiser all-gathers (data parallel) instead of from the fp32 masters.  [frozen_lo4, frozen_hi4): float4 range of the
    // flat region whose parameters do not train (vy_fixed, reference HLVAE.py:209-211): gradients cleared, nothing updated.
    constexpr int T = 64, CLD = T + 1;
    __shared__ float tile[T * CLD];
    if ((int)blockIdx.x >= set.total_tiles) {                       // flat region (update == 1 only)
        const AdamScalars a = adam_scalars((float)(step_count[0] + 1), lr, b1, b2, eps, gscale);
        float4* P4 = reinterpret_cast<float4*>(P);
        float4* G4 = reinterpret_cast<float4*>(const_cast<float*>(G));
        float4* M14 = reinterpret_cast<float4*>(M1);
        float4* M24 = reinterpret_cast<float4*>(M2);
        const long nfb = (long)gridDim.x - set.total_tiles;
        for (long i = ((long)blockIdx.x - set.total_tiles) * HL_THREADS + threadIdx.x; i < n4_flat; i += nfb * HL_THREADS) {
            if (i >= frozen_lo4 && i < frozen_hi4) {              // torch.optim.Adam skips parameters without a gradient
                G4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                continue;
            }
            float4 p = P4[i], g = G4[i], m = M14[i], v = M24[i];
            p.x = adam_one(p.x, g.x, m.x, v.x, a);
            p.y = adam_one(p.y, g.y, m.y, v.y, a);
            p.z = adam_one(p.z, g.z, m.z, v.z, a);
            p.w = adam_one(p.w, g.w, m.w, v.w, a);
            P4[i] = p;
            M14[i] = m;
            M24[i] = v;
            G4[i] = make_float4(0.f, 0.f, 0.f, 0.f);           // the atomically accumulated gradients start the next step at 0
        }
        __syncthreads();
        if (threadIdx.x == 0 && ticket_total != 0) hl_take_ticket(step_count, tick_shards, ticket_total);
        return;
    }
    int mi = 0;
#pragma unroll
    for (int k = 1; k < SHADOW_MAX; ++k)
        if (k < set.n && (int)blockIdx.x >= set.m[k].tile0) mi = k;
    const ShadowMat mt = set.m[mi];
    // Neighbouring column tiles of a row band share their boundary cache lines whenever the row stride is not a multiple of
    // 128 B (y_layer: 2000 B).  For such a matrix the XCD-contiguous remap puts them behind the same L2 (PMC: 1.45x -> 1.0x
    // of the algorithmic reads).  Matrices with aligned rows keep the round-robin placement: there the remap only
    // multiplies the number of DRAM streams by 8 (all-in-one launch: 81 vs 37 us).  Every matrix starts at a workgroup
    // index that is a multiple of 8 and owns a multiple of 8 workgroups, so the local index keeps the hardware XCD; the
    // padding workgroups land beyond the last row band and do nothing.
    int tl = blockIdx.x - mt.tile0;
    if (mt.remap) tl = xcd_remap(tl, mt.npad);
    const int r0 = (tl / mt.tiles_c) * T, c0 = (tl % mt.tiles_c) * T;
    AdamScalars a;
    if (update) a = adam_scalars((float)(step_count[0] + 1), lr, b1, b2, eps, gscale);
    const int c4 = (threadIdx.x & 15) * 4, rq = threadIdx.x >> 4;      // 16 float4 per tile row, 16 rows per pass
    // all 16 global loads of this lane are issued before the first store (a store to P / M1 / M2 orders every later load
    // from the same array behind it): unconditional loads from clamped addresses, predicated stores
    float4 p[4], g[4], m[4], v[4];
    long o[4];
    bool in[4];
    // rows are 16-byte aligned when C % 4 == 0 (block-uniform): one float4 per access; otherwise (an input width such as
    // 38 expanded columns) element-wise accesses clipped to the row
    const bool vec = (mt.C & 3) == 0;
    const int nv = min(4, mt.C - (c0 + c4));
    auto ld4 = [&](const float* b, long off) -> float4 {
        if (Pb16 != nullptr) {                                  // (block-uniform) bf16 source, same indexing
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (vec) {
                const uint2 q = *reinterpret_cast<const uint2*>(Pb16 + off);
                r.x = bf2f((bf16_t)(q.x & 0xffff)); r.y = bf2f((bf16_t)(q.x >> 16));
                r.z = bf2f((bf16_t)(q.y & 0xffff)); r.w = bf2f((bf16_t)(q.y >> 16));
                return r;
            }
            if (nv > 0) r.x = bf2f(Pb16[off]);
            if (nv > 1) r.y = bf2f(Pb16[off + 1]);
            if (nv > 2) r.z = bf2f(Pb16[off + 2]);
            if (nv > 3) r.w = bf2f(Pb16[off + 3]);
            return r;
        }
        if (vec) return *reinterpret_cast<const float4*>(b + off);
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (nv > 0) r.x = b[off];
        if (nv > 1) r.y = b[off + 1];
        if (nv > 2) r.z = b[off + 2];
        if (nv > 3) r.w = b[off + 3];
        return r;
    };
    auto st4 = [&](float* b, long off, const float4& x) {
        if (vec) { *reinterpret_cast<float4*>(b + off) = x; return; }
        if (nv > 0) b[off] = x.x;
        if (nv > 1) b[off + 1] = x.y;
        if (nv > 2) b[off + 2] = x.z;
        if (nv > 3) b[off + 3] = x.w;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rq + 16 * i;
        in[i] = r0 + r < mt.R && c0 + c4 < mt.C;
        const long srow = (in[i] && mt.rowsrc != nullptr) ? mt.rowsrc[r0 + r] : r0 + r;       // shadows: tile order; masters: their own
        o[i] = in[i] ? mt.off + srow * mt.C + c0 + c4 : mt.off;
        p[i] = ld4(P, o[i]);
    }
    if (update) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            g[i] = ld4(G, o[i]);       // (non-temporal hints measured slower: 41 vs 38 us)
            m[i] = ld4(M1, o[i]);
            v[i] = ld4(M2, o[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = rq + 16 * i;
        if (!in[i]) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (update && in[i]) {
            p[i].x = adam_one(p[i].x, g[i].x, m[i].x, v[i].x, a);
            p[i].y = adam_one(p[i].y, g[i].y, m[i].y, v[i].y, a);
            p[i].z = adam_one(p[i].z, g[i].z, m[i].z, v[i].z, a);
            p[i].w = adam_one(p[i].w, g[i].w, m[i].w, v[i].w, a);
            st4(P, o[i], p[i]);
            st4(M1, o[i], m[i]);
            st4(M2, o[i], v[i]);
        }
        if (r0 + r < mt.Rcover && c0 + c4 < mt.Ccover) {        // row-major shadow: 4 bf16 = one 8-byte store
            uint2 pk;
            pk.x = (uint32_t)f2bf(p[i].x) | ((uint32_t)f2bf(p[i].y) << 16);
            pk.y = (uint32_t)f2bf(p[i].z) | ((uint32_t)f2bf(p[i].w) << 16);
            *reinterpret_cast<uint2*>(mt.dst + (size_t)(mt.row_off + r0 + r) * mt.ldd + c0 + c4) = pk;
        }
        if (mt.dstT != nullptr) {
            tile[r * CLD + c4 + 0] = p[i].x; tile[r * CLD + c4 + 1] = p[i].y;
            tile[r * CLD + c4 + 2] = p[i].z; tile[r * CLD + c4 + 3] = p[i].w;
        }
    }
    if (mt.dstT != nullptr) {          // block-uniform branch
        __syncthreads();
        const int r4 = (threadIdx.x & 15) * 4, cq = threadIdx.x >> 4;   // lane -> (column, 4 consecutive rows)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cq + 16 * i;
            if (r0 + r4 < mt.Rcover && c0 + c < mt.Ccover) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(tile[(r4 + 0) * CLD + c]) | ((uint32_t)f2bf(tile[(r4 + 1) * CLD + c]) << 16);
                pk.y = (uint32_t)f2bf(tile[(r4 + 2) * CLD + c]) | ((uint32_t)f2bf(tile[(r4 + 3) * CLD + c]) << 16);
                *reinterpret_cast<uint2*>(mt.dstT + (size_t)(c0 + c) * mt.ldT + mt.row_off + r0 + r4) = pk;
            }
        }
    }
    if (update && ticket_total != 0) {
        __syncthreads();
        if (threadIdx.x == 0) hl_take_ticket(step_count, tick_shards, ticket_total);
    }
}

// bit i of `which` selects matrix i: 0 Wy, 1 W1, 2 Wd, 3 Wmu, 4 Wlv, 5 + i extra encoder layer i, 5 + HLVAE_MAX_EXTRA + j extra decoder layer j
static unsigned all_matrices(const hlvae_dims& d) {
    unsigned m = 0x1f;
    for (int i = 0; i < d.n_xe; ++i) m |= 1u << (5 + i);
    for (int j = 0; j < d.n_xd; ++j) m |= 1u << (5 + HLVAE_MAX_EXTRA + j);
    return m;
}
unsigned hl_all_matrices(const hlvae_plan* p) { return all_matrices(p->d); }

static ShadowSet make_set(const hlvae_plan* p, const hlvae_ws* ws, unsigned which) {
    const hlvae_dims& d = p->d;
    ShadowSet s;
    auto put = [&](int i, long off, int R, int C, bf16_t* dst, int ldd, int row_off, bf16_t* dstT, int ldT, int Rc, int Cc) {
        ShadowMat& m = s.m[i];
        m.off = off; m.R = R; m.C = C; m.dst = dst; m.ldd = ldd; m.row_off = row_off; m.dstT = dstT; m.ldT = ldT;
        m.Rcover = Rc; m.Ccover = Cc;
        m.rowsrc = nullptr;
        m.tiles_c = (Cc + 63) / 64;
    };
    // Wy [NY][h_d] -> wys [NY][hdp] (+ [hdp][NYp]);  W1 [h_e][X] -> w1s [hep][Xp];
    // Wd [h_d][L] -> wds [hdp][Lp] (+ [Lp][hdp]);  [Wmu; Wlv] [L][h_e] each -> wmls [2Lp][hep] (+ [hep][2Lp])
    // (conv: y_layer is [2592][h_d], the first encoder Linear is [h_e][2592] and also needs its transpose: the convolutional
    //  features receive a gradient, the raw inputs of the MLP path do not)
    put(0, d.o_wy, d.NYl, d.h_d, ws->wys, d.hdp, 0, ws->wyTs, d.NYlp, d.NYl, d.hdp);
    s.m[0].rowsrc = d.conv ? nullptr : p->wy_rowsrc_dev;
    // (deeper trunks: W1 is the last encoder Linear [h_e][K1], Wd the first decoder Linear [h_d0][L])
    put(1, d.o_w1, d.h_e, d.K1, ws->w1s, d.K1p, 0, (d.conv || d.n_xe > 0) ? ws->w1Ts : nullptr, d.hep, d.hep, d.K1p);
    put(2, d.o_wd, d.h_d0, d.L, ws->wds, d.Lp, 0, ws->wdTs, d.hd0p, d.hd0p, d.Lp);
    put(3, d.o_wmu, d.L, d.h_e, ws->wmls, d.hep, 0, ws->wmlTs, 2 * d.Lp, d.Lp, d.hep);
    put(4, d.o_wlv, d.L, d.h_e, ws->wmls, d.hep, d.Lp, ws->wmlTs, 2 * d.Lp, d.Lp, d.hep);
    for (int i = 0; i < HLVAE_MAX_EXTRA; ++i) {
        if (i < d.n_xe) put(5 + i, d.xe[i].o_w, d.xe[i].n_out, d.xe[i].n_in, ws->xe[i].w, d.xe[i].n_in_p, 0, ws->xe[i].wT, d.xe[i].n_out_p,
                            d.xe[i].n_out_p, d.xe[i].n_in_p);
        if (i < d.n_xd) put(5 + HLVAE_MAX_EXTRA + i, d.xd[i].o_w, d.xd[i].n_out, d.xd[i].n_in, ws->xd[i].w, d.xd[i].n_in_p, 0, ws->xd[i].wT,
                            d.xd[i].n_out_p, d.xd[i].n_out_p, d.xd[i].n_in_p);
    }
    which &= all_matrices(d);
    int n = 0;                                   // keep the selected matrices (bit i of `which`), compacted
    for (int i = 0; i < SHADOW_MAX; ++i)
        if (which & (1u << i)) s.m[n++] = s.m[i];
    s.n = n;
    int t = 0;
    for (int i = 0; i < n; ++i) {
        s.m[i].tile0 = t;
        s.m[i].npad = ru(s.m[i].tiles_c * ((s.m[i].Rcover + 63) / 64), 8);
        s.m[i].remap = ((long)s.m[i].C * 4) % 128 != 0;
        t += s.m[i].npad;
    }
    s.total_tiles = t;
    return s;
}

static int check_set(const ShadowSet& s) {
    for (int i = 0; i < s.n; ++i) {
        const ShadowMat& m = s.m[i];
        // (a 4-row group of the transposed shadow may run past Rcover: it lands in the zero padding, ldT >= ru(Rcover, 4))
        HL_REQUIRE(m.off % 4 == 0 && m.ldd % 4 == 0 && m.Ccover % 4 == 0 &&
                       (m.dstT == nullptr || (m.ldT % 4 == 0 && m.row_off % 4 == 0 && m.ldT >= m.row_off + ru(m.Rcover, 4))),
                   HLVAE_ESHAPE, "weight matrix %d: hidden / latent / input widths must be multiples of 4 (C=%d off=%ld)", i,
                   m.C, m.off);
    }
    return 0;
}

int hl_conv_pack_weights(const hlvae_plan* p, const hlvae_ws* ws, hipStream_t s);

int hl_refresh_shadows(const hlvae_plan* p, const hlvae_ws* ws, hipStream_t s) {
    const ShadowSet set = make_set(p, ws, all_matrices(p->d));
    if (int rc = check_set(set)) return rc;
    {
        HL_PROF("shadow_cast", s);
        k_adam_tiled<<<set.total_tiles, HL_THREADS, 0, s>>>(set, ws->P, ws->G, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 0, 0u,
                                                            nullptr, 0, 0, nullptr);
        HL_LAUNCH_CHECK();
    }
    if (p->d.conv) return hl_conv_pack_weights(p, ws, s);
    return 0;
}

static int flat_blocks(const hlvae_dims& d) {
    const long n4 = d.atomic_region / 4;
    int blocks = (int)((n4 + HL_THREADS - 1) / HL_THREADS);
    return blocks > 64 ? 64 : (blocks < 1 ? 1 : blocks);
}

// workgroups of the launch that covers the matrices `which` (bit i = matrix i of make_set) (+ the flat region)
int hl_adam_grid(const hlvae_plan* p, const hlvae_ws* ws, unsigned which, int with_flat) {
    return make_set(p, ws, which).total_tiles + (with_flat ? flat_blocks(p->d) : 0);
}

// One launch of the optimiser step: Adam + shadows of the matrices `which` (+ the small flat region).  `ticket_total` =
// workgroups of all the launches that make up this step (hl_adam_grid), 0 = takes no ticket (the caller orders a later
// launch of the same step behind this one).
int hl_adam_part(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr, float b1,
                 float b2, float eps, float gscale, unsigned which, int with_flat, unsigned ticket_total, const char* label,
                 hipStream_t s, long flat_n = -1, int tick_slot = 0) {      // flat_n >= 0: only [0, flat_n) of the small region (same workgroup count)
    // ticket_total: shard units of ALL the launches of this step (common.h hl_ticket_units per launch); tick_slot: this launch's slot
    const hlvae_dims& d = p->d;
    HL_REQUIRE(d.atomic_region % 4 == 0, HLVAE_ESHAPE, "atomic region %ld not a multiple of 4", (long)d.atomic_region);
    const ShadowSet set = make_set(p, ws, which);
    if (int rc = check_set(set)) return rc;
    const int grid = set.total_tiles + (with_flat ? flat_blocks(d) : 0);
    {
        HL_PROF(label, s);
        k_adam_tiled<<<grid, HL_THREADS, 0, s>>>(set, ws->P, ws->G, m1, m2, step_count, lr, b1, b2, eps, gscale, 1,
                                                 with_flat ? (flat_n >= 0 ? flat_n : d.atomic_region) / 4 : 0, ticket_total, nullptr, d.frozen_lo / 4,
                                                 (d.frozen_hi + 3) / 4, p->tick_dev + (size_t)tick_slot * HL_TICK_WORDS);
        HL_LAUNCH_CHECK();
    }
    if (with_flat && d.conv) return hl_conv_pack_weights(p, ws, s);   // the convolution weights live in the flat region
    return 0;
}

// the whole step in ONE launch: weight tiles plus the workgroups of the small flat region (disjoint parts of the arena)
int hl_adam(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr, float b1,
            float b2, float eps, float gscale, hipStream_t s, int skip_wy) {
    const unsigned which = all_matrices(p->d) & (skip_wy ? ~0x01u : ~0u);
    return hl_adam_part(p, ws, m1, m2, step_count, lr, b1, b2, eps, gscale, which, 1, hl_ticket_units(hl_adam_grid(p, ws, which, 1)),
                        skip_wy ? "adam_weights_shadows" : "adam_all_in_one", s);
}


// ---- sharded optimiser (data parallel; hl-vae_amd/parallel.py) ------------------------------------------------------------
// Every rank owns one contiguous slice of the DENSE part of the arena [atomic_region, arena_size).  Its gradients arrive
// from the reduce-scatter as a compact fp32 buffer; the rank updates master / m / v of the slice only (1 / world of the
// optimiser's HBM traffic) and leaves a bf16 copy of the updated values in the flat bf16 arena that the all-gather then
// completes on every rank; hl_shadows_from_bf16 turns that arena into the padded shadows the MFMA kernels read.
__global__ __launch_bounds__(HL_THREADS) void k_adam_flat(float* __restrict__ P, const float* __restrict__ gsh,
                                                          float* __restrict__ M1, float* __restrict__ M2,
                                                          bf16_t* __restrict__ Pb, long lo4, long n4,
                                                          const int64_t* __restrict__ step_count, float lr, float b1, float b2,
                                                          float eps, float gscale) {
    const AdamScalars a = adam_scalars((float)(step_count[0] + 1), lr, b1, b2, eps, gscale);
    float4* P4 = reinterpret_cast<float4*>(P) + lo4;
    const float4* G4 = reinterpret_cast<const float4*>(gsh);
    float4* M14 = reinterpret_cast<float4*>(M1) + lo4;
    float4* M24 = reinterpret_cast<float4*>(M2) + lo4;
    uint2* B4 = reinterpret_cast<uint2*>(Pb) + lo4;
    // two float4 per lane and pass: 8 loads in flight before the first store
    const long stride = (long)gridDim.x * HL_THREADS;
    for (long i = (long)blockIdx.x * HL_THREADS + threadIdx.x; i < n4; i += 2 * stride) {
        const long j = i + stride;
        const bool two = j < n4;
        float4 p0 = P4[i], g0 = G4[i], m0 = M14[i], v0 = M24[i];
        float4 p1 = p0, g1 = g0, m1 = m0, v1 = v0;
        if (two) { p1 = P4[j]; g1 = G4[j]; m1 = M14[j]; v1 = M24[j]; }
        auto upd = [&](float4& p, const float4& g, float4& m, float4& v) {
            p.x = adam_one(p.x, g.x, m.x, v.x, a);
            p.y = adam_one(p.y, g.y, m.y, v.y, a);
            p.z = adam_one(p.z, g.z, m.z, v.z, a);
            p.w = adam_one(p.w, g.w, m.w, v.w, a);
            uint2 pk;
            pk.x = (uint32_t)f2bf(p.x) | ((uint32_t)f2bf(p.y) << 16);
            pk.y = (uint32_t)f2bf(p.z) | ((uint32_t)f2bf(p.w) << 16);
            return pk;
        };
        const uint2 k0 = upd(p0, g0, m0, v0);
        P4[i] = p0; M14[i] = m0; M24[i] = v0; B4[i] = k0;
        if (two) {
            const uint2 k1 = upd(p1, g1, m1, v1);
            P4[j] = p1; M14[j] = m1; M24[j] = v1; B4[j] = k1;
        }
    }
}

int hl_adam_flat(const hlvae_plan* p, const hlvae_ws* ws, const float* gsh, float* m1, float* m2, uint16_t* pb16,
                 const int64_t* step_count, long lo, long n, float lr, float b1, float b2, float eps, float gscale, hipStream_t s) {
    const hlvae_dims& d = p->d;
    HL_REQUIRE(lo % 4 == 0 && n % 4 == 0 && lo >= d.atomic_region && lo + n <= d.arena_size, HLVAE_ESHAPE,
               "adam_flat: slice [%ld, +%ld) must be 4-aligned and inside the dense region [%ld, %ld)", lo, n,
               (long)d.atomic_region, (long)d.arena_size);
    if (n == 0) return 0;
    const long n4 = n / 4;
    long blocks = (n4 + 2 * HL_THREADS - 1) / (2 * HL_THREADS);
    if (blocks > 2048) blocks = 2048;
    HL_PROF("adam_flat_shard", s);
    k_adam_flat<<<(int)blocks, HL_THREADS, 0, s>>>(ws->P, gsh, m1, m2, pb16, lo / 4, n4, step_count, lr, b1, b2, eps, gscale);
    HL_LAUNCH_CHECK();
    return 0;
}

// padded bf16 shadows (row-major + transposed) of the matrices `which` from the flat bf16 arena
int hl_shadows_from_bf16(const hlvae_plan* p, const hlvae_ws* ws, const uint16_t* pb16, unsigned which, const char* label,
                         hipStream_t s) {
    const ShadowSet set = make_set(p, ws, which);
    if (int rc = check_set(set)) return rc;
    if (set.total_tiles == 0) return 0;
    HL_PROF(label, s);
    k_adam_tiled<<<set.total_tiles, HL_THREADS, 0, s>>>(set, ws->P, ws->G, nullptr, nullptr, nullptr, 0.f, 0.f, 0.f, 0.f, 0.f, 0, 0, 0u,
                                                        pb16, 0, 0, nullptr);
    HL_LAUNCH_CHECK();
    return 0;
}
