// extern "C" boundary of libhlvae_hip.so (include/hlvae_hip.h): argument validation against what the
// kernels' grids assume, then stream-ordered launches.  No allocation of caller memory, no device sync.
#include <stdarg.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "common.h"

// launchers implemented next to their kernels
int hl_launch_gemm_f32(const bf16_t*, int, const bf16_t*, int, float*, int, int, int, int, int, int, float*, const char*, hipStream_t,
                       const int32_t* rowmap = nullptr);
int hl_launch_gemm_splitk(const bf16_t*, int, const bf16_t*, int, float*, int, int, int, int, int, const char*, hipStream_t);
int hl_launch_gemm_act(int, const bf16_t*, int, const bf16_t*, int, int, int, int, const float*, int, const bf16_t*,
                       bf16_t*, int, bf16_t*, int, int, float*, const char*, hipStream_t);
int hl_launch_y_heads(const hlvae_plan*, const hlvae_ws*, const float*, float, int, int, int, int, hipStream_t);
int hl_launch_elbo_finalize(const hlvae_plan*, const hlvae_ws*, int, int, hipStream_t);
int hl_launch_head_grad_reduce(const hlvae_plan*, const hlvae_ws*, int, hipStream_t);
int hl_launch_scale_dy(const hlvae_plan*, const hlvae_ws*, const float*, int, int, hipStream_t);
int hl_launch_step_metrics(const hlvae_plan*, const hlvae_ws*, int, float*, hipStream_t, const hlvae_ws* fin_ws = nullptr, int fin_B = 0);
int hl_launch_stats(const hlvae_plan*, const hlvae_ws*, const double*, const double*, int, hipStream_t);
int hl_launch_pack(const hlvae_plan*, const hlvae_ws*, const double*, const double*, int, int, hipStream_t);
int hl_refresh_shadows(const hlvae_plan*, const hlvae_ws*, hipStream_t);
int hl_launch_mid_fwd_fused(const hlvae_plan*, const hlvae_ws*, const float*, int, uint64_t, int, int, hipStream_t, const bf16_t* xin = nullptr);
bool hl_mid_direct_fwd(const hlvae_dims&);
bool hl_mid_direct_bwd(const hlvae_dims&);
int hl_launch_mid_bwd_fused(const hlvae_plan*, const hlvae_ws*, const float*, const float*, float, int, int, hipStream_t, const bf16_t* dyin = nullptr,
                            float* zero_ptr = nullptr, long zero_n = 0);
int hl_adam(const hlvae_plan*, const hlvae_ws*, float*, float*, int64_t*, float, float, float, float, float, hipStream_t, int);
int hl_launch_conv_enc_fwd(const hlvae_plan*, const hlvae_ws*, const double*, const double*, const float*, const uint8_t*,
                           const int32_t*, int, int, hipStream_t);
int hl_launch_stats_compact(const hlvae_plan*, const hlvae_ws*, const float*, const uint8_t*, const int32_t*, int, hipStream_t);
int hl_launch_pack_compact(const hlvae_plan*, const hlvae_ws*, const float*, const uint8_t*, const int32_t*, int, int, hipStream_t);
int hl_launch_conv_dec_fwd(const hlvae_plan*, const hlvae_ws*, int, hipStream_t);
int hl_launch_conv_dec_bwd(const hlvae_plan*, const hlvae_ws*, int, int, hipStream_t);
int hl_launch_conv_enc_bwd(const hlvae_plan*, const hlvae_ws*, int, hipStream_t);
int hl_launch_gemm_f32_group(GemmGroup, const char*, hipStream_t);
bool hl_gemm_adam_ok(int, int, int, bool);
int hl_gemm_adam_grid(const AdamGemmGroup&);
int hl_launch_gemm_adam(AdamGemmGroup, float*, float*, float*, int64_t*, float, float, float, float, float, unsigned, const char*, hipStream_t,
                        float* Gflat, long flat_lo, long flat_n, unsigned long long* tick_shards);
int hl_wgrad_ksplit(long, int);
int hl_launch_transpose_bf16(const bf16_t*, int, bf16_t*, int, int, int, const char*, hipStream_t);
int hl_adam_grid(const hlvae_plan*, const hlvae_ws*, unsigned, int);
unsigned hl_all_matrices(const hlvae_plan*);
int hl_adam_part(const hlvae_plan*, const hlvae_ws*, float*, float*, int64_t*, float, float, float, float, float, unsigned, int,
                 unsigned, const char*, hipStream_t, long flat_n = -1, int tick_slot = 0);
int hl_adam_flat(const hlvae_plan*, const hlvae_ws*, const float*, float*, float*, uint16_t*, const int64_t*, long, long, float, float,
                 float, float, float, hipStream_t);
int hl_shadows_from_bf16(const hlvae_plan*, const hlvae_ws*, const uint16_t*, unsigned, const char*, hipStream_t);

static thread_local char g_err[512] = "";
void hl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static inline int padded_batch(int B) { return ru(B, 128); }

// ---- per-kernel event timing -----------------------------------------------------------------------
#include <map>
#include <string>
struct ProfRec { const char* name; hipEvent_t a, b; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}
void hl_prof_begin(const char* name, hipStream_t s) {
    if (!g_prof_on) return;
    ProfRec r{name, prof_event(), prof_event()};
    if (r.a) (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
}
void hl_prof_end(hipStream_t s) {
    if (!g_prof_on || g_prof.empty()) return;
    if (g_prof.back().b) (void)hipEventRecord(g_prof.back().b, s);
}

static unsigned long long* g_stamp_buf = nullptr;
unsigned long long* hl_stamp_slot(int slot) { return g_stamp_buf != nullptr ? g_stamp_buf + 8 * HL_STAMP_SUB * slot : nullptr; }

extern "C" {

int hlvae_abi_version(void) { return HLVAE_ABI_VERSION; }

int hlvae_stamp_slots(void) { return HL_ST_N; }
int hlvae_stamp_words(void) { return 8 * HL_STAMP_SUB * HL_ST_N; }
void hlvae_stamp_buffer(uint64_t* buf) { g_stamp_buf = reinterpret_cast<unsigned long long*>(buf); }

void hlvae_prof_enable(int on) { g_prof_on = on != 0; }

int hlvae_prof_report(char* buf, int buflen) {
    HL_CHECK(hipDeviceSynchronize());
    std::map<std::string, std::pair<long, double>> agg;
    for (auto& r : g_prof) {
        float ms = 0.f;
        if (r.a && r.b && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            auto& e = agg[r.name];
            e.first += 1;
            e.second += ms;
        }
        if (r.a) g_prof_pool.push_back(r.a);
        if (r.b) g_prof_pool.push_back(r.b);
    }
    g_prof.clear();
    int n = 0;
    for (auto& kv : agg) {
        int w = snprintf(buf + n, buflen > n ? buflen - n : 0, "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
        if (w < 0 || n + w >= buflen) break;
        n += w;
    }
    return 0;
}
const char* hlvae_last_error(void) { return g_err; }
void hlvae_struct_sizes(int32_t* dims_bytes, int32_t* var_bytes, int32_t* ws_bytes) {
    if (dims_bytes) *dims_bytes = (int32_t)sizeof(hlvae_dims);
    if (var_bytes) *var_bytes = (int32_t)sizeof(hlvae_var);
    if (ws_bytes) *ws_bytes = (int32_t)sizeof(hlvae_ws);
}

void hlvae_dims_fill(hlvae_dims* d) {
    d->Xp = ru(d->X, 64);
    d->hep = ru(d->h_e, 64);
    d->hdp = ru(d->h_d, 64);
    d->Lp = ru(d->L, 32);
    d->NY = d->D * d->y_dim;
    d->NYp = ru(d->NY, 64);
    d->n_stat = d->n_real + d->n_pos;
    d->Xe = d->conv ? 32 * 9 * 9 : d->X;              // HLVAE.py:155
    d->Xep = ru(d->Xe, 64);
    d->NYl = d->conv ? 32 * 9 * 9 : d->NY;            // HLVAE.py:246
    d->NYlp = ru(d->NYl, 64);
    // deeper trunks: the fused kernels' "first encoder Linear" is the LAST one of the stack, their decoder trunk the FIRST
    if (d->n_xe < 0 || d->n_xe > HLVAE_MAX_EXTRA) d->n_xe = 0;      // (rejected by plan_create; keep the loops below in range)
    if (d->n_xd < 0 || d->n_xd > HLVAE_MAX_EXTRA) d->n_xd = 0;
    for (int i = 0; i < d->n_xe; ++i) { d->xe[i].n_in_p = ru(d->xe[i].n_in, 64); d->xe[i].n_out_p = ru(d->xe[i].n_out, 64); }
    for (int i = 0; i < d->n_xd; ++i) { d->xd[i].n_in_p = ru(d->xd[i].n_in, 64); d->xd[i].n_out_p = ru(d->xd[i].n_out, 64); }
    d->K1 = d->n_xe > 0 ? d->xe[d->n_xe - 1].n_out : d->Xe;
    d->K1p = ru(d->K1, 64);
    if (d->h_d0 <= 0 || d->n_xd == 0) d->h_d0 = d->h_d;
    d->hd0p = ru(d->h_d0, 64);
    if (d->n_xe == 0 && d->n_xd == 0) d->o_xw = d->o_wy;
}

int hlvae_plan_create(hlvae_plan** out, const hlvae_dims* dims, const hlvae_var* vars, const int32_t* var_order) {
    HL_REQUIRE(out && dims && vars, HLVAE_EINVAL, "plan_create: null argument");
    hlvae_dims d = *dims;
    hlvae_dims_fill(&d);
    HL_REQUIRE(d.D > 0 && d.X >= d.D && d.h_e > 0 && d.h_d > 0 && d.L > 0 && d.Theta >= d.X, HLVAE_EINVAL, "plan_create: bad dims");
    HL_REQUIRE(d.y_dim == 5 || d.y_dim == 3 || d.y_dim == 8, HLVAE_EINVAL, "plan_create: y_dim=%d (instantiated: 3, 5, 8)", d.y_dim);
    HL_REQUIRE(!d.conv || d.y_dim == 5, HLVAE_EINVAL, "plan_create: the convolutional decoder has y_dim = 5 output channels");
    HL_REQUIRE(d.Lp <= 64, HLVAE_EINVAL, "plan_create: latent_dim=%d > 64 unsupported", d.L);
    HL_REQUIRE(d.arena_size % 4 == 0, HLVAE_EINVAL, "plan_create: arena_size must be a multiple of 4 floats");
    HL_REQUIRE(dims->n_xe >= 0 && dims->n_xe <= HLVAE_MAX_EXTRA && dims->n_xd >= 0 && dims->n_xd <= HLVAE_MAX_EXTRA, HLVAE_EINVAL,
               "plan_create: at most %d extra hidden layers per side (got %d / %d)", HLVAE_MAX_EXTRA, dims->n_xe, dims->n_xd);
    for (int i = 0; i < d.n_xe + d.n_xd; ++i) {     // chains: X -> xe[0] -> .. -> K1 (-> h_e);  h_d0 -> xd[0] -> .. -> h_d
        const bool enc = i < d.n_xe;
        const int k = enc ? i : i - d.n_xe;
        const hlvae_layer& l = enc ? d.xe[k] : d.xd[k];
        const int want_in = k > 0 ? (enc ? d.xe[k - 1].n_out : d.xd[k - 1].n_out) : (enc ? d.Xe : d.h_d0);
        HL_REQUIRE(l.n_in == want_in && l.n_out > 0, HLVAE_EINVAL, "plan_create: extra %s layer %d is %d -> %d, its input has %d",
                   enc ? "encoder" : "decoder", k, l.n_in, l.n_out, want_in);
        HL_REQUIRE(l.o_w % 4 == 0 && l.o_w >= d.o_xw && l.o_w + (int64_t)l.n_in * l.n_out <= d.o_wy && l.o_b >= 0 &&
                       l.o_b + l.n_out <= d.atomic_region, HLVAE_EINVAL,
                   "plan_create: extra %s layer %d: weight must lie in [o_xw, o_wy) at a multiple of 4 floats, bias in the atomic region",
                   enc ? "encoder" : "decoder", k);
    }
    HL_REQUIRE(d.n_xd == 0 || d.xd[d.n_xd - 1].n_out == d.h_d, HLVAE_EINVAL, "plan_create: the last decoder layer must have h_d = %d outputs", d.h_d);
    HL_REQUIRE(d.o_xw > d.o_w1 && d.o_xw <= d.o_wy, HLVAE_EINVAL, "plan_create: o_xw outside (o_w1, o_wy]");
    HL_REQUIRE(!d.lin_e || (d.n_xe == 0 && d.h_e == 2 * d.L), HLVAE_EINVAL, "plan_create: lin_e (no encoder hidden layer) needs h_e = 2 L = %d "
               "and no extra encoder layers (h_e = %d)", 2 * d.L, d.h_e);
    HL_REQUIRE(!d.lin_d || (d.n_xd == 0 && d.h_d == d.L), HLVAE_EINVAL, "plan_create: lin_d (no decoder hidden layer) needs h_d = L = %d and no "
               "extra decoder layers (h_d = %d)", d.L, d.h_d);
    HL_REQUIRE(!d.conv || d.D == 36 * 36, HLVAE_EINVAL, "plan_create: the convolutional model needs 36 x 36 = 1296 variables "
               "(HLVAE.py:305), got %d", d.D);
    if (d.conv) {
        const int64_t offs[8] = {d.o_c1w, d.o_c1b, d.o_c2w, d.o_c2b, d.o_t1w, d.o_t1b, d.o_t2w, d.o_t2b};
        for (int i = 0; i < 8; ++i)
            HL_REQUIRE(offs[i] >= 0 && offs[i] < d.atomic_region, HLVAE_EINVAL, "plan_create: convolution parameter %d outside "
                       "the atomic gradient region", i);
        HL_REQUIRE(d.o_cv_lo >= 0 && d.cv_n > 0 && d.o_cv_lo + d.cv_n <= d.atomic_region && d.o_by + d.NYl == d.o_cv_lo + d.cv_n,
                   HLVAE_EINVAL, "plan_create: convolution gradient range [%ld, +%ld) must end with y_layer's bias inside the "
                   "atomic region", (long)d.o_cv_lo, (long)d.cv_n);
        for (int i = 0; i < 8; ++i)
            HL_REQUIRE(offs[i] >= d.o_cv_lo, HLVAE_EINVAL, "plan_create: convolution parameter %d below the gradient range", i);
        // 16-byte stores into the per-workgroup partial rows (csrc/conv.hip)
        HL_REQUIRE(d.cv_n % 4 == 0 && (d.o_c2w - d.o_cv_lo) % 4 == 0 && (d.o_t1w - d.o_cv_lo) % 4 == 0, HLVAE_EINVAL,
                   "plan_create: convolution weights must start at multiples of 4 floats inside the gradient range");
    }
    std::vector<int32_t> col2var(d.Xp, -1), stat_var(d.n_stat > 0 ? d.n_stat : 1, 0);
    int x = 0, nstat_seen = 0;
    for (int i = 0; i < d.D; ++i) {
        const hlvae_var& v = vars[i];
        HL_REQUIRE(v.kind >= HLVAE_REAL && v.kind <= HLVAE_ORDINAL, HLVAE_EINVAL, "variable %d: kind %d", i, v.kind);
        const bool disc = v.kind == HLVAE_CAT || v.kind == HLVAE_ORDINAL;
        HL_REQUIRE(disc ? (v.ncls >= 2 && v.ncls <= 16) : v.ncls == 1, HLVAE_EINVAL,
                   "variable %d: nclass %d unsupported (cat/ordinal: 2..16)", i, v.ncls);
        HL_REQUIRE(!(disc && v.ncls > 8) || d.y_dim == 5, HLVAE_EINVAL, "variable %d: more than 8 classes is instantiated for "
                   "y_dim = 5 only", i);
        HL_REQUIRE(v.xoff == x, HLVAE_EINVAL, "variable %d: xoff %d, expected %d", i, v.xoff, x);
        HL_REQUIRE(v.poff >= 0 && v.poff + v.ncls <= d.Theta, HLVAE_EINVAL, "variable %d: poff %d outside the %d parameter columns", i, v.poff, d.Theta);
        HL_REQUIRE(v.w_off >= 0 && v.b_off >= 0 && v.w_off < d.atomic_region && v.b_off < d.atomic_region,
                   HLVAE_EINVAL, "variable %d: head offsets outside the atomic gradient region", i);
        if (v.kind == HLVAE_REAL || v.kind == HLVAE_POS) {
            const bool lvn = v.w2_off >= 0;      // logvar_network: a second head (log-variance) instead of the free parameter
            HL_REQUIRE(v.sidx >= 0 && v.sidx < d.n_stat && (lvn ? (v.b2_off >= 0 && v.w2_off < d.atomic_region && v.b2_off < d.atomic_region &&
                                                                   v.poff2 >= 0 && v.poff2 < d.Theta)
                                                                : v.e_off >= 0), HLVAE_EINVAL, "variable %d: sidx / variance parameter offsets", i);
            stat_var[v.sidx] = i;
            ++nstat_seen;
        }
        if (v.kind == HLVAE_ORDINAL) HL_REQUIRE(v.e_off >= 0, HLVAE_EINVAL, "variable %d: thresholds offset", i);
        if (d.conv && disc)
            HL_REQUIRE(v.r_off >= 0 && v.rb_off >= 0 && v.r_off + v.ncls <= d.atomic_region && v.rb_off < d.atomic_region,
                       HLVAE_EINVAL, "variable %d: representation-layer offsets", i);
        for (int k = 0; k < v.ncls; ++k) col2var[x + k] = i;
        x += v.ncls;
    }
    HL_REQUIRE(x == d.X, HLVAE_EINVAL, "plan_create: variables cover %d columns, X=%d", x, d.X);
    HL_REQUIRE(nstat_seen == d.n_stat, HLVAE_EINVAL, "plan_create: %d statistic rows, n_stat=%d", nstat_seen, d.n_stat);
    // kernel-facing order of the variables (head kernel, dY, y_layer's shadows)
    std::vector<hlvae_var> sorted(vars, vars + d.D);
    std::vector<int32_t> rowsrc;
    bool permuted = false;
    if (var_order != nullptr) {
        HL_REQUIRE(!d.conv, HLVAE_EINVAL, "plan_create: var_order is not available for the convolutional model (its y_grouped is a pixel grid)");
        std::vector<char> seen(d.D, 0);
        for (int j = 0; j < d.D; ++j) {
            const int o = var_order[j];
            HL_REQUIRE(o >= 0 && o < d.D && !seen[o], HLVAE_EINVAL, "plan_create: var_order is not a permutation (position %d)", j);
            seen[o] = 1;
            sorted[j] = vars[o];
            permuted |= o != j;
        }
    }
    for (int j = 0; j < d.D; ++j) sorted[j].pad = var_order != nullptr ? var_order[j] : j;
    if (permuted) {
        rowsrc.resize(d.NYlp);
        for (int r = 0; r < d.NYlp; ++r) rowsrc[r] = r < d.NY ? sorted[r / d.y_dim].pad * d.y_dim + r % d.y_dim : r;
    }
    hlvae_plan* p = new hlvae_plan();
    p->d = d;
    p->vars_dev = nullptr; p->col2var_dev = nullptr; p->stat_var_dev = nullptr; p->vars_sorted_dev = nullptr; p->wy_rowsrc_dev = nullptr;
    p->tick_dev = nullptr;
    p->kmax = 2;
    p->pend_flags = 0;
    p->defer_join = 0;
    for (int i = 0; i < d.D; ++i)
        if ((vars[i].kind == HLVAE_CAT || vars[i].kind == HLVAE_ORDINAL) && vars[i].ncls > p->kmax) p->kmax = vars[i].ncls;
    for (auto& st : p->side) st = nullptr;
    for (auto& ev : p->ev) ev = nullptr;
    hipError_t e = hipSuccess;
    for (auto& st : p->side)
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (auto& ev : p->ev)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&p->tick_dev, sizeof(unsigned long long) * HL_TICK_SLOTS * HL_TICK_WORDS);
    if (e == hipSuccess) e = hipMemset(p->tick_dev, 0, sizeof(unsigned long long) * HL_TICK_SLOTS * HL_TICK_WORDS);
    if (e == hipSuccess) e = hipMalloc(&p->vars_dev, sizeof(hlvae_var) * d.D);
    if (e == hipSuccess) e = hipMalloc(&p->col2var_dev, sizeof(int32_t) * d.Xp);
    if (e == hipSuccess) e = hipMalloc(&p->stat_var_dev, sizeof(int32_t) * stat_var.size());
    if (e == hipSuccess) e = hipMemcpy(p->vars_dev, vars, sizeof(hlvae_var) * d.D, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&p->vars_sorted_dev, sizeof(hlvae_var) * d.D);
    if (e == hipSuccess) e = hipMemcpy(p->vars_sorted_dev, sorted.data(), sizeof(hlvae_var) * d.D, hipMemcpyHostToDevice);
    if (e == hipSuccess && permuted) e = hipMalloc(&p->wy_rowsrc_dev, sizeof(int32_t) * rowsrc.size());
    if (e == hipSuccess && permuted) e = hipMemcpy(p->wy_rowsrc_dev, rowsrc.data(), sizeof(int32_t) * rowsrc.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->col2var_dev, col2var.data(), sizeof(int32_t) * d.Xp, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->stat_var_dev, stat_var.data(), sizeof(int32_t) * stat_var.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        hl_set_error("plan_create: %s", hipGetErrorString(e));
        hlvae_plan_destroy(p);
        return (int)e;
    }
    *out = p;
    return 0;
}

void hlvae_plan_destroy(hlvae_plan* p) {
    if (!p) return;
    if (p->vars_dev) (void)hipFree(p->vars_dev);
    if (p->tick_dev) (void)hipFree(p->tick_dev);
    if (p->vars_sorted_dev) (void)hipFree(p->vars_sorted_dev);
    if (p->wy_rowsrc_dev) (void)hipFree(p->wy_rowsrc_dev);
    if (p->col2var_dev) (void)hipFree(p->col2var_dev);
    if (p->stat_var_dev) (void)hipFree(p->stat_var_dev);
    for (auto& st : p->side)
        if (st) (void)hipStreamDestroy(st);
    for (auto& ev : p->ev)
        if (ev) (void)hipEventDestroy(ev);
    delete p;
}

#define CHECK_B()                                                                                         \
    HL_REQUIRE(p && ws, HLVAE_EINVAL, "null plan/workspace");                                              \
    HL_REQUIRE(B > 0 && padded_batch(B) <= ws->Bp_max, HLVAE_ESHAPE, "batch %d exceeds workspace (Bp_max=%d)", B, ws->Bp_max); \
    const int Bp = padded_batch(B);                                                                       \
    const hlvae_dims& d = p->d;                                                                            \
    hipStream_t st = (hipStream_t)s;                                                                       \
    (void)d; (void)Bp; (void)st

int hlvae_refresh_shadows(const hlvae_plan* p, const hlvae_ws* ws, hlvae_stream s) {
    HL_REQUIRE(p && ws, HLVAE_EINVAL, "null plan/workspace");
    return hl_refresh_shadows(p, ws, (hipStream_t)s);
}

int hlvae_normalize_stats(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B,
                          hlvae_stream s) {
    CHECK_B();
    return hl_launch_stats(p, ws, data, mask, B, st);
}

int hlvae_normalize_pack(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B,
                         hlvae_stream s) {
    CHECK_B();
    if (d.conv) {   // representation layer + conv1 + conv2 (HLVAE.py:293-308) produce the encoder input
        HL_REQUIRE(ws->cpack && ws->img, HLVAE_EINVAL, "convolutional model without its workspace buffers");
        return hl_launch_conv_enc_fwd(p, ws, data, mask, nullptr, nullptr, nullptr, B, Bp, st);
    }
    return hl_launch_pack(p, ws, data, mask, B, Bp, st);
}

int hlvae_normalize_fused(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B,
                          hlvae_stream s) {
    CHECK_B();
    // real columns carry no batch statistics under conv (utils.py:99-102): only pos columns need the sums.
    // (A pack kernel that sums its own columns over all rows was measured for the MLP path: the 48 workgroups that own real
    //  columns become a long tail -- 0.208 vs 0.205 ms/step -- so the two-pass form stays.)
    int rc;
    if ((!d.conv || d.n_pos > 0) && (rc = hl_launch_stats(p, ws, data, mask, B, st))) return rc;
    return hlvae_normalize_pack(p, ws, data, mask, B, s);
}

int hlvae_feed_stats(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                     int B, hlvae_stream s) {
    CHECK_B();
    HL_REQUIRE(values && mask8 && rows, HLVAE_EINVAL, "feed_stats: null pointer");
    return hl_launch_stats_compact(p, ws, values, mask8, rows, B, st);
}

int hlvae_feed_pack(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                    int B, hlvae_stream s) {
    CHECK_B();
    HL_REQUIRE(values && mask8 && rows, HLVAE_EINVAL, "feed_pack: null pointer");
    if (d.conv) {
        HL_REQUIRE(ws->cpack && ws->img, HLVAE_EINVAL, "convolutional model without its workspace buffers");
        return hl_launch_conv_enc_fwd(p, ws, nullptr, nullptr, values, mask8, rows, B, Bp, st);
    }
    return hl_launch_pack_compact(p, ws, values, mask8, rows, B, Bp, st);
}

int hlvae_feed_fused(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                     int B, hlvae_stream s) {
    CHECK_B();
    HL_REQUIRE(values && mask8 && rows, HLVAE_EINVAL, "feed_fused: null pointer");
    int rc;
    if ((!d.conv || d.n_pos > 0) && (rc = hl_launch_stats_compact(p, ws, values, mask8, rows, B, st))) return rc;
    return hlvae_feed_pack(p, ws, values, mask8, rows, B, s);
}

int hlvae_feed_prefetch(const hlvae_plan* p, const hlvae_ws* ws_next, const float* values, const uint8_t* mask8,
                        const int32_t* rows, int B, hlvae_stream s) {
    const hlvae_ws* ws = ws_next;
    CHECK_B();
    HL_REQUIRE(values && mask8 && rows, HLVAE_EINVAL, "feed_prefetch: null pointer");
    HL_REQUIRE(!d.conv, HLVAE_EINVAL, "feed_prefetch: the convolutional input stage reads the weights (conv1 / conv2) -- it cannot "
               "run ahead of the optimiser step");
    // like hlvae_step_metrics: only the dependency is recorded; the two kernels are queued on a side stream by the next
    // hlvae_backward* / hlvae_join (behind whatever else was deferred)
    if (p->pend_flags & (HL_PEND_FEED | HL_PEND_RUNNING))
        if (int rc = hlvae_join(p, s)) return rc;
    HL_CHECK(hipEventRecord(p->ev[5], st));
    p->pend_feed_ws = *ws_next;
    p->pend_feed_B = B;
    p->pend_feed_vals = values;
    p->pend_feed_mask = mask8;
    p->pend_feed_rows = rows;
    p->pend_flags |= HL_PEND_FEED;
    return 0;
}

// ---- deeper trunks (dims[1] / dims[3] with several entries, HLVAE.py:125-137, 232-242): the layers around the fused middle run
// as plain GEMM + ReLU launches with both bf16 layouts of every activation kept for the backward pass -------------------------
static int hl_check_extra(const hlvae_dims& d, const hlvae_ws* ws) {
    for (int i = 0; i < d.n_xe; ++i)
        HL_REQUIRE(ws->xe[i].w && ws->xe[i].wT && ws->xe[i].a && ws->xe[i].aT && ws->xe[i].d && ws->xe[i].dT, HLVAE_EINVAL,
                   "extra encoder layer %d: workspace buffers missing", i);
    for (int i = 0; i < d.n_xd; ++i)
        HL_REQUIRE(ws->xd[i].w && ws->xd[i].wT && ws->xd[i].a && ws->xd[i].aT && ws->xd[i].d && ws->xd[i].dT, HLVAE_EINVAL,
                   "extra decoder layer %d: workspace buffers missing", i);
    HL_REQUIRE(ws->u0 && ws->u0T, HLVAE_EINVAL, "ws->u0 / u0T missing (== u / uT for one decoder layer)");
    HL_REQUIRE(d.n_xd == 0 || (ws->xd[d.n_xd - 1].a == ws->u && ws->xd[d.n_xd - 1].aT == ws->uT), HLVAE_EINVAL,
               "the last decoder layer must write ws->u / ws->uT (y_layer's input)");
    HL_REQUIRE(d.n_xe == 0 || (ws->w1Ts && ws->dt), HLVAE_EINVAL, "extra encoder layers need ws->w1Ts and ws->dt");
    return 0;
}

// u0 -> .. -> u  (HLVAE.py:336: self.hidden(z), all but its first Linear + ReLU)
static int hl_extra_decoder_fwd(const hlvae_plan* p, const hlvae_ws* ws, int B, int Bp, hipStream_t st) {
    const hlvae_dims& d = p->d;
    const bf16_t* in = ws->u0;
    for (int j = 0; j < d.n_xd; ++j) {
        const hlvae_layer& l = d.xd[j];
        if (int rc = hl_launch_gemm_act(0, in, l.n_in_p, ws->xd[j].w, l.n_in_p, Bp, l.n_out_p, l.n_in_p, ws->P + l.o_b, l.n_out, nullptr,
                                        ws->xd[j].a, l.n_out_p, ws->xd[j].aT, Bp, B, nullptr, "dec_hidden_relu", st)) return rc;
        in = ws->xd[j].a;
    }
    return 0;
}

int hlvae_encoder_fwd(const hlvae_plan* p, const hlvae_ws* ws, const float* eps, int sample, uint64_t rng_host_offset,
                      int B, hlvae_stream s) {
    CHECK_B();
    int rc;
    HL_REQUIRE(ws->splitk_enc >= 1, HLVAE_EINVAL, "splitk_enc");
    if ((rc = hl_check_extra(d, ws))) return rc;
    const bf16_t* in = ws->xn;                     // input of the last encoder Linear: Xn, or the output of the layers before it
    for (int i = 0; i < d.n_xe; ++i) {              // HLVAE.py:316-317: VAE_encoder_common_layers, all but its last Linear + ReLU
        const hlvae_layer& l = d.xe[i];
        if ((rc = hl_launch_gemm_act(0, in, l.n_in_p, ws->xe[i].w, l.n_in_p, Bp, l.n_out_p, l.n_in_p, ws->P + l.o_b, l.n_out, nullptr,
                                     ws->xe[i].a, l.n_out_p, ws->xe[i].aT, Bp, B, nullptr, "enc_hidden_relu", st))) return rc;
        in = ws->xe[i].a;
    }
    if (hl_mid_direct_fwd(d)) {     // narrow input: the fused middle computes Xn W1^T itself
        if ((rc = hl_launch_mid_fwd_fused(p, ws, eps, sample, rng_host_offset, B, Bp, st, in))) return rc;
        return hl_extra_decoder_fwd(p, ws, B, Bp, st);
    }
    // trunk product Xn W1^T as split-K slabs (HLVAE.py:316-317, evaluated once) ...
    if ((rc = hl_launch_gemm_splitk(in, d.K1p, ws->w1s, d.K1p, ws->slab, d.hep, Bp, d.hep, d.K1p, ws->splitk_enc, "enc1_splitk", st))) return rc;
    // ... then bias + ReLU, mean / log-var heads, clamp, reparameterisation AND the decoder trunk in one fused kernel
    if ((rc = hl_launch_mid_fwd_fused(p, ws, eps, sample, rng_host_offset, B, Bp, st))) return rc;
    return hl_extra_decoder_fwd(p, ws, B, Bp, st);
}

int hlvae_decoder_fwd(const hlvae_plan* p, const hlvae_ws* ws, const float* g_logpx, float g_scale, int want_grad,
                      int want_params, int trunk, int B, hlvae_stream s) {
    CHECK_B();
    int rc;
    if (trunk) {   // U = relu(z Wd^T + bd) from ws->zb (HLVAE.py:336): only when z was set by the caller (decode(z));
                   // after hlvae_encoder_fwd the trunk is already in ws->u
        if ((rc = hl_check_extra(d, ws))) return rc;
        // (no decoder hidden layer: the "trunk" is the identity on the latent -- bias-only epilogue)
        if ((rc = hl_launch_gemm_act(d.lin_d ? 2 : 0, ws->zb, d.Lp, ws->wds, d.Lp, Bp, d.hd0p, d.Lp, ws->P + d.o_bd, d.h_d0, nullptr, ws->u0,
                                     d.hd0p, ws->u0T, Bp, B, nullptr, "dec1_relu", st))) return rc;
        if ((rc = hl_extra_decoder_fwd(p, ws, B, Bp, st))) return rc;
    }
    if (d.conv) {   // y_layer as a plain Linear, then the two transposed convolutions (HLVAE.py:337-341)
        if ((rc = hl_launch_gemm_act(2, ws->u, d.hdp, ws->wys, d.hdp, Bp, d.NYlp, d.hdp, ws->P + d.o_by, d.NYl, nullptr, ws->yc,
                                     d.NYlp, nullptr, 0, B, nullptr, "y_layer_conv", st))) return rc;
        if ((rc = hl_launch_conv_dec_fwd(p, ws, B, st))) return rc;
    }
    if ((rc = hl_launch_y_heads(p, ws, g_logpx, g_scale, want_grad != 0, want_params, B, Bp, st))) return rc;
    if (want_grad != 2) return hl_launch_elbo_finalize(p, ws, B, Bp, st);
    // training step: nothing on the critical path reads the scalar reductions.  Only the dependency is recorded; the
    // one-workgroup kernel is queued on a side stream by the next hlvae_backward* / hlvae_join, which forks ONCE for all
    // its side work (a fork of its own here costs more in a HIP graph than the 7 us it saves: 0.229 vs 0.205 ms/step)
    if ((rc = hlvae_join(p, s))) return rc;       // earlier deferred work of this plan
    HL_CHECK(hipEventRecord(p->ev[5], st));
    p->pend_fin_ws = *ws;
    p->pend_fin_B = B;
    p->pend_flags |= HL_PEND_FINALIZE;
    return 0;
}

int hlvae_scale_dy(const hlvae_plan* p, const hlvae_ws* ws, const float* g_logpx, int B, hlvae_stream s) {
    CHECK_B();
    HL_REQUIRE(g_logpx, HLVAE_EINVAL, "scale_dy: null gradient");
    return hl_launch_scale_dy(p, ws, g_logpx, B, Bp, st);
}

static int hl_flush_deferred(const hlvae_plan* p, hipStream_t side, bool side_is_ordered, int mask = HL_PEND_DEFERRED, bool second = false,
                             bool feed_first = false);

int hlvae_step_metrics(const hlvae_plan* p, const hlvae_ws* ws, int B, float* err, hlvae_stream s) {
    CHECK_B();
    HL_REQUIRE(err, HLVAE_EINVAL, "step_metrics: null output");
    // metrics only READ what the decoder left behind: they run on a side stream beside whatever the caller queues next
    // (normally the backward pass).  Only the dependency is recorded here; the launches themselves are queued by the next
    // hlvae_backward* (after its critical path, see there) or hlvae_join, which also re-joins the caller's stream.
    if (p->pend_flags & (HL_PEND_METRICS | HL_PEND_RUNNING))       // an earlier, un-joined call
        if (int rc = hlvae_join(p, s)) return rc;
    HL_CHECK(hipEventRecord(p->ev[5], st));
    p->pend_ws = *ws;
    p->pend_B = B;
    p->pend_err = err;
    p->pend_flags |= HL_PEND_METRICS;
    return 0;
}

// queues the deferred launches on `side`.  side_is_ordered: the caller has just made `side` wait for a LATER point of
// its stream than ev[5] (one fork point for all the side work of the backward pass)
// mask: which of the pending pieces to queue now (the others stay pending).  second: this chain's end is ev[1] / HL_PEND_RUNNING2
// instead of ev[5] / HL_PEND_RUNNING -- two chains on two side streams (ELBO scalars + metrics on one, the next batch's input
// stage on the other) that hlvae_join waits for separately.
static int hl_flush_deferred(const hlvae_plan* p, hipStream_t side, bool side_is_ordered, int mask, bool second, bool feed_first) {
    const int todo = p->pend_flags & mask & HL_PEND_DEFERRED;
    if (!todo) return 0;
    if (!side_is_ordered) HL_CHECK(hipStreamWaitEvent(side, p->ev[5], 0));
    const bool both = (todo & HL_PEND_FINALIZE) && (todo & HL_PEND_METRICS);
    // feed_first: the next batch's input stage ahead of the ELBO scalars / metrics (fused-optimiser step: it then runs beside
    // dU_splitk and the fused middle instead of beside the two streaming launches, where its pack kernel took 52 us instead of 13)
    if (feed_first && (todo & HL_PEND_FEED)) {
        const int B = p->pend_feed_B;
        if (p->d.n_stat > 0)
            if (int rc = hl_launch_stats_compact(p, &p->pend_feed_ws, p->pend_feed_vals, p->pend_feed_mask, p->pend_feed_rows, B, side)) return rc;
        if (int rc = hl_launch_pack_compact(p, &p->pend_feed_ws, p->pend_feed_vals, p->pend_feed_mask, p->pend_feed_rows, B,
                                            (B + 127) / 128 * 128, side)) return rc;
    }
    if ((todo & HL_PEND_FINALIZE) && !both) {
        const int B = p->pend_fin_B;
        if (int rc = hl_launch_elbo_finalize(p, &p->pend_fin_ws, B, (B + 127) / 128 * 128, side)) return rc;
    }
    if (todo & HL_PEND_METRICS)       // (both pending: the ELBO bookkeeping rides in the metrics launch -- one link less)
        if (int rc = hl_launch_step_metrics(p, &p->pend_ws, p->pend_B, p->pend_err, side, both ? &p->pend_fin_ws : nullptr, p->pend_fin_B))
            return rc;
    if ((todo & HL_PEND_FEED) && !feed_first) {       // input stage of the NEXT batch into the other buffer set (data only, no weights)
        const int B = p->pend_feed_B;
        if (p->d.n_stat > 0)
            if (int rc = hl_launch_stats_compact(p, &p->pend_feed_ws, p->pend_feed_vals, p->pend_feed_mask, p->pend_feed_rows, B, side)) return rc;
        if (int rc = hl_launch_pack_compact(p, &p->pend_feed_ws, p->pend_feed_vals, p->pend_feed_mask, p->pend_feed_rows, B,
                                            (B + 127) / 128 * 128, side)) return rc;
    }
    HL_CHECK(hipEventRecord(p->ev[second ? 1 : 5], side));
    p->pend_flags = (p->pend_flags & ~todo) | (second ? HL_PEND_RUNNING2 : HL_PEND_RUNNING);
    return 0;
}

int hlvae_join(const hlvae_plan* p, hlvae_stream s) {
    HL_REQUIRE(p, HLVAE_EINVAL, "null plan");
    if (int rc = hl_flush_deferred(p, g_prof_on ? (hipStream_t)s : p->side[1], false)) return rc;
    if (p->pend_flags & HL_PEND_RUNNING) HL_CHECK(hipStreamWaitEvent((hipStream_t)s, p->ev[5], 0));
    if (p->pend_flags & HL_PEND_RUNNING2) HL_CHECK(hipStreamWaitEvent((hipStream_t)s, p->ev[1], 0));
    p->pend_flags &= ~(HL_PEND_RUNNING | HL_PEND_RUNNING2);
    return 0;
}

int hlvae_set_defer_join(const hlvae_plan* p, int on) {
    HL_REQUIRE(p, HLVAE_EINVAL, "null plan");
    p->defer_join = on == 2 ? 2 : (on ? 1 : 0);   // 2: the deferred side work is queued on the CALLER's stream at the end of the backward call
    return 0;
}

int hlvae_reset_pending(const hlvae_plan* p) {
    HL_REQUIRE(p, HLVAE_EINVAL, "null plan");
    p->pend_flags = 0;          // deferred side work recorded against a capture that failed: dropped
    return 0;
}

int hlvae_zero_grad(const hlvae_plan* p, const hlvae_ws* ws, hlvae_stream s) {
    HL_REQUIRE(p && ws, HLVAE_EINVAL, "null plan/workspace");
    HL_CHECK(hipMemsetAsync(ws->G, 0, sizeof(float) * p->d.atomic_region, (hipStream_t)s));
    return 0;
}

int hlvae_backward_wy(const hlvae_plan* p, const hlvae_ws* ws, int B, hlvae_stream s) {
    CHECK_B();
    // d Wy = dY^T U   [NY][h_d]: the largest gradient (55 % of the arena for D4) and the first one that is final, so a
    // data-parallel host can start its all-reduce while hlvae_backward(skip_wy = 1) is still running
    HL_REQUIRE(!d.conv, HLVAE_EINVAL, "backward_wy: not available for the convolutional model (y_layer's gradient is final "
               "only after the transposed convolutions' backward)");
    return hl_launch_gemm_f32(ws->dyT, Bp, ws->uT, Bp, ws->G + d.o_wy, d.h_d, d.NY, d.h_d, Bp, 0, 0, nullptr, "dWy", st, p->wy_rowsrc_dev);
}

// does hlvae_backward_adam apply the optimiser step in the epilogue of the weight-gradient GEMMs for this model and batch?
static bool hl_fused_optimiser(const hlvae_dims& d, int Bp) {
    return !d.conv && d.n_xe == 0 && d.n_xd == 0 && !d.lin_e && !d.lin_d && getenv("HL_NO_FUSED_ADAM") == nullptr && hl_gemm_adam_ok(d.NYl, d.h_d, Bp, false) &&
           hl_gemm_adam_ok(d.h_e, d.K1, Bp, false) && hl_gemm_adam_ok(d.h_d0, d.L, Bp, true) && hl_gemm_adam_ok(2 * d.Lp, d.h_e, Bp, true);
}

int hlvae_backward_adam_fused(const hlvae_plan* p, int B) {
    if (!p || B < 1) return 0;
    return hl_fused_optimiser(p->d, padded_batch(B)) ? 1 : 0;
}

struct HlAdamArgs {
    float *m1, *m2;
    int64_t* step_count;
    float lr, b1, b2, eps, gscale;
};

static int hl_backward_impl(const hlvae_plan* p, const hlvae_ws* ws, const float* g_mu, const float* g_lv, float kl_std_weight,
                            int skip_wy, int B, hlvae_stream s, const HlAdamArgs* opt) {
    CHECK_B();
    int rc;
    HL_REQUIRE(ws->splitk_dec >= 1, HLVAE_EINVAL, "splitk_dec");
    // Critical path, all on the caller's stream:
    //     dY -> dU -> d(mu, lv), dT -> {dW1, dWd, d[Wmu; Wlv]} (one grouped launch) -> [join] -> Adam of everything but Wy.
    // Leaves of the dependency graph run on two side streams:
    //     side 0:  d Wy = dY^T U (forks before dU_splitk), then -- behind dU_splitk, the last reader of its shadows --
    //              y_layer's Adam update + shadow refresh: the largest slice of the HBM-bound optimiser runs under the
    //              latency-bound middle of the backward pass instead of after it;
    //     side 1:  whatever hlvae_decoder_fwd(want_grad = 2) / hlvae_step_metrics deferred (ELBO scalars, row-M metrics),
    //              forked behind dU_splitk.
    // Why this shape (rocprofv3 kernel traces of the replayed HIP graph on MI355X, tools/timeline.py):
    //   * a node's FIRST-created child stays on its parent's hardware queue; every other child starts 6-20 us late
    //     (cross-queue signal) -> the whole critical path is queued first, the side work afterwards, behind events
    //     recorded along the way;
    //   * two HBM-bound kernels running concurrently take far longer than back to back (the two Adam launches: 71 us
    //     each together, 28 + 22 us apart) -> the second Adam launch waits for the join;
    //   * three small GEMMs as one grouped launch instead of a third queue.
    // Measured (ms/step, D4, batch 512): one stream 0.225; forks queued side-first 0.196; this order 0.170; also tried:
    // Adam split three ways and fully concurrent 0.234, dWy forked behind dU 0.192, deferred kernels behind y_layer's
    // Adam on side 0 0.186, deferred kernels forked before dU 0.183, Adam applied in the epilogue of the weight-gradient
    // GEMMs (no gradient round trip) 0.180-0.190.
    // (per-kernel timing on: everything on the caller's stream, so that every kernel is timed alone)
    hipStream_t s0 = g_prof_on ? st : p->side[0], s1 = g_prof_on ? st : p->side[1];
    if (d.conv)     // d y_grouped -> d a2 -> d (y_layer output), weight gradients of the transposed convolutions
        if ((rc = hl_launch_conv_dec_bwd(p, ws, B, Bp, st))) return rc;
    if (skip_wy)    // data-parallel host: the small region is all-reduced right after this call
        if ((rc = hl_launch_head_grad_reduce(p, ws, Bp, st))) return rc;
    const bf16_t* dyl = d.conv ? ws->dyc : ws->dy;          // gradient of y_layer's output, both layouts
    const bf16_t* dylT = d.conv ? ws->dycT : ws->dyT;
    HL_CHECK(hipEventRecord(p->ev[0], st));        // dY is final
    const bool fused_opt = opt != nullptr && !skip_wy && hl_fused_optimiser(d, Bp);
    AdamGemmGroup g_rest{}, g_wy{};
    unsigned tickets = 0;
    if (fused_opt) {
        g_rest.n = 3;
        g_rest.K = Bp;
        g_rest.p[0] = AdamGemmProb{ws->dtT, ws->xnT, nullptr, ws->w1s, nullptr, (long)d.o_w1, 0, Bp, Bp, d.h_e, d.K1, 0, 0, d.K1p, 0, 0, 0, 0};
        g_rest.p[1] = AdamGemmProb{ws->duT, ws->zbT, nullptr, ws->wds, ws->wdTs, (long)d.o_wd, 0, Bp, Bp, d.h_d0, d.L, 0, 0, d.Lp, d.hd0p, 0, 0, 0};
        g_rest.p[2] = AdamGemmProb{ws->dmlT, ws->tT, nullptr, ws->wmls, ws->wmlTs, (long)d.o_wmu, (long)d.o_wlv, Bp, Bp, 2 * d.Lp, d.h_e, d.Lp,
                                   d.L, d.hep, 2 * d.Lp, 0, 0, 0};
        HL_REQUIRE((ws->wys_next == nullptr) == (ws->wyTs_next == nullptr), HLVAE_EINVAL, "ws->wys_next / wyTs_next: both or none");
        bf16_t* wys_out = ws->wys_next != nullptr ? ws->wys_next : ws->wys;
        bf16_t* wyTs_out = ws->wyTs_next != nullptr ? ws->wyTs_next : ws->wyTs;
        g_wy.n = 1;
        g_wy.K = Bp;
        g_wy.p[0] = AdamGemmProb{dylT, ws->uT, p->wy_rowsrc_dev, wys_out, wyTs_out, (long)d.o_wy, 0, Bp, Bp, d.NYl, d.h_d, 0, 0, d.hdp,
                                 d.NYlp, 0, 0, 0};
        // (shard units of the three launches: common.h hl_take_ticket; slots 0 / 1 / 2 = the grouped launch, y_layer's, the small region's)
        tickets = hl_ticket_units(hl_gemm_adam_grid(g_rest)) + hl_ticket_units(hl_gemm_adam_grid(g_wy)) + hl_ticket_units(hl_adam_grid(p, ws, 0u, 1));
        // (y_layer's launch created BEFORE dU_splitk -- the head kernel's first child starts at once, the caller's chain pays the
        // cross-queue start instead: dU_splitk 21 us late and 25 us long beside it, 0.142 vs 0.137 ms/step)
    }
    // d U slabs = dY Wy (split-K), then the fused middle: dU -> dz -> d(mu, lv) -> dT, bias gradients
    const bool direct_bwd = hl_mid_direct_bwd(d);      // narrow y_layer: the fused middle computes dY Wy itself (and is then the last
    if (direct_bwd) {                                  // reader of y_layer's shadows)
    } else if (d.n_xd == 0) {
        if ((rc = hl_launch_gemm_splitk(dyl, d.NYlp, ws->wyTs, d.NYlp, ws->slab, d.hdp, Bp, d.hdp, d.NYlp, ws->splitk_dec, "dU_splitk", st))) return rc;
        HL_CHECK(hipEventRecord(p->ev[2], st));        // the last reader of y_layer's weight shadows is done
    } else {
        // deeper decoder: layer by layer down to the first one, whose pre-activation gradient the fused middle consumes.
        //   d_last = (dY Wy) .* (u > 0), bias gradient = its column sums
        if ((rc = hl_check_extra(d, ws))) return rc;
        const int jl = d.n_xd - 1;
        if ((rc = hl_launch_gemm_act(1, dyl, d.NYlp, ws->wyTs, d.NYlp, Bp, d.hdp, d.NYlp, nullptr, d.h_d, ws->xd[jl].a, ws->xd[jl].d, d.hdp,
                                     ws->xd[jl].dT, Bp, B, ws->G + d.xd[jl].o_b, "dU_hidden", st))) return rc;
        HL_CHECK(hipEventRecord(p->ev[2], st));
        for (int j = jl; j >= 0; --j) {
            const hlvae_layer& l = d.xd[j];
            //   d W_j = d_j^T a_(j-1)   [n_out][n_in]
            if ((rc = hl_launch_gemm_f32(ws->xd[j].dT, Bp, j > 0 ? ws->xd[j - 1].aT : ws->u0T, Bp, ws->G + l.o_w, l.n_in, l.n_out, l.n_in, Bp,
                                         0, 0, nullptr, "dW_dec_hidden", st))) return rc;
            if (j > 0) {    //   d_(j-1) = (d_j W_j) .* (a_(j-1) > 0)
                if ((rc = hl_launch_gemm_act(1, ws->xd[j].d, l.n_out_p, ws->xd[j].wT, l.n_out_p, Bp, l.n_in_p, l.n_out_p, nullptr, l.n_in,
                                             ws->xd[j - 1].a, ws->xd[j - 1].d, l.n_in_p, ws->xd[j - 1].dT, Bp, B, ws->G + d.xd[j - 1].o_b,
                                             "dU_hidden", st))) return rc;
            } else {        //   the first layer's dU as ONE fp32 slab for k_mid_bwd_fused
                if ((rc = hl_launch_gemm_splitk(ws->xd[0].d, l.n_out_p, ws->xd[0].wT, l.n_out_p, ws->slab, d.hd0p, Bp, d.hd0p, l.n_out_p, 1,
                                                "dU0", st))) return rc;
            }
        }
    }
    // split-K slices of the grouped weight-gradient launch add with atomics into [Wd | Wmu | Wlv | W1]'s gradients: cleared by the
    // fused middle (no memset node on the critical path)
    const bool fused_opt_ = opt != nullptr && !skip_wy && hl_fused_optimiser(d, Bp);
    const int wg_ksplit = fused_opt_ ? 1 : hl_wgrad_ksplit((long)((d.h_e + 63) / 64) * ((d.K1 + 63) / 64), Bp);
    // a small y_layer (64-feature models: 0.16 M parameters) and no extra layers: its weight gradient rides in the grouped launch
    // (a fourth problem) and ONE optimiser launch follows -- there is nothing to keep apart, and the side chain dWy -> Adam ->
    // gradient fold (dWy 37 us beside the fused middle) was what the final launch waited for
    const bool wy_in_group = opt != nullptr && !fused_opt_ && !d.conv && !skip_wy && d.n_xe == 0 && d.n_xd == 0 &&
                             (long)d.NYl * d.h_d <= 512l * 1024;
    const long clear_n = wy_in_group ? (long)(d.arena_size - d.o_wd) : (long)(d.o_xw - d.o_wd);
    const bool clear_in_mid = wg_ksplit > 1 && d.o_wd % 4 == 0 && clear_n % 4 == 0;
    if ((rc = hl_launch_mid_bwd_fused(p, ws, g_mu, g_lv, kl_std_weight, B, Bp, st, direct_bwd ? dyl : nullptr,
                                      clear_in_mid ? ws->G + d.o_wd : nullptr, clear_in_mid ? clear_n : 0))) return rc;
    if (direct_bwd) HL_CHECK(hipEventRecord(p->ev[2], st));
    // ---- single-process training step, MLP with one hidden layer per side: optimiser step in the weight gradients' epilogue ----
    //   caller:  dU -> fused middle -> {dW1, dWd, d[Wmu; Wlv]} + Adam + shadows (ONE launch) -> [join] -> next step
    //   side 0:  dWy + Adam + shadows (one launch; into the second shadow pair when the caller provides one, else behind dU_splitk,
    //            this step's last reader of y_layer's shadows) -> head-gradient fold -> Adam of the small region
    //   side 1:  deferred ELBO scalars / metrics / next batch's input stage
    // The three optimiser launches share one completion ticket (whichever workgroup finishes last commits the step number).
    // Before: gradients through HBM, then two HBM-bound k_adam_tiled launches that must not overlap (71 us each together, 28 +
    // 22 us apart), y_layer's on side 0 and the rest behind a cross-queue join: 44 us of optimiser on the critical path of a
    // 145 us step (rocprofv3 trace of the replayed graph: profiles/r2_*_step_timeline.txt).
    if (fused_opt) {
        // The four bias vectors whose gradients the fused middle accumulates (bd, bmu, blv, b1: the END of the small region, by
        // construction of the arena) take their Adam step in workgroup 0 of THIS launch, after its tile -- behind the middle kernel by
        // stream order.  The rest of the small region (head parameters, y_layer's bias: gradients from the head kernel) is then
        // independent of the backward pass: fold + Adam run at the head of side 0, and side 0 ends with y_layer's launch instead
        // of 17 us behind it.  (An event behind the middle kernel for the small-region launch to wait on costs the caller's queue
        // 6.6 us -- a second child of that kernel: 0.141 vs 0.134 ms; without any dependency the order held only because y_layer's
        // 30 us launch sat in between.)
        const long bias_lo = d.o_bd, bias_n = d.atomic_region - d.o_bd;
        HL_REQUIRE(d.o_bd % 4 == 0 && bias_n % 4 == 0 && d.o_bmu > d.o_bd && d.o_blv > d.o_bd && d.o_b1 > d.o_bd && d.o_by < d.o_bd,
                   HLVAE_EINVAL, "backward_adam: the arena must end its small region with [bd | bmu | blv | b1]");
        if ((rc = hl_launch_gemm_adam(g_rest, ws->P, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale,
                                      tickets, "dW1_dWd_dWmu_adam", st, ws->G, bias_lo, bias_n, p->tick_dev))) return rc;
        HL_CHECK(hipStreamWaitEvent(s0, p->ev[0], 0));
        if ((rc = hl_launch_head_grad_reduce(p, ws, Bp, s0))) return rc;
        if ((rc = hl_adam_part(p, ws, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale, 0u, 1, tickets,
                               "adam_small", s0, bias_lo, 2))) return rc;      // (behind y_layer's launch instead: 0.139 vs 0.137 ms)
        // y_layer's launch starts behind dU_splitk even when it writes the second shadow pair: the fused middle (64 workgroups of
        // 1024 threads) must be RESIDENT before the streaming launch takes every register of the chip -- started first, the
        // middle kernel waited for three of the four streaming workgroups of its CU to retire (33 us instead of 15; 0.138 ->
        // 0.135 ms/step, 0.134 with the input stage first on side 1).  Large batches: dU_splitk is long (62 us at 4096 rows) and
        // the middle kernel has 256 workgroups; there the launch starts with the head kernel's end when it may (0.485 vs 0.522 ms)
        const bool small_batch = Bp < 2048;
        if (g_wy.p[0].sh == ws->wys || small_batch) HL_CHECK(hipStreamWaitEvent(s0, p->ev[2], 0));
        if ((rc = hl_launch_gemm_adam(g_wy, ws->P, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale,
                                      tickets, "dWy_adam", s0, nullptr, 0, 0, p->tick_dev + HL_TICK_WORDS))) return rc;

        HL_CHECK(hipEventRecord(p->ev[3], s0));
        // (large batches, metrics behind side 0's chain and the input stage alone on side 1: its 16-workgroup statistics kernel
        //  starves beside the streaming launches -- 110 us instead of 12 -- and the step does not move, 0.490 vs 0.483 ms)
        const bool inline_deferred = (p->pend_flags & HL_PEND_DEFERRED) && p->defer_join == 2;
        if (inline_deferred) {
        } else if (p->pend_flags & HL_PEND_DEFERRED) {
            HL_CHECK(hipStreamWaitEvent(s1, p->ev[0], 0));   // forked at the head kernel like side 0: with the fork behind
            // dU_splitk the graph executor put both side chains on ONE hardware queue, y_layer's launch last (0.166 vs 0.144 ms/step)
            if ((rc = hl_flush_deferred(p, s1, true, HL_PEND_DEFERRED, false, small_batch))) return rc;
        }
        HL_CHECK(hipStreamWaitEvent(st, p->ev[3], 0));
        // the caller does not wait for anything else at the end of this step (GP prior with a deferred state update: its chains are
        // the step's critical path and share the hardware queues with side 1 -- the next batch's input stage sat 250 us behind
        // them, and the next step's encoder behind it): metrics and input stage in line, BEHIND both streaming launches (beside
        // y_layer's the pack kernel took 93 us instead of 24)
        if (inline_deferred)
            if ((rc = hl_flush_deferred(p, st, true, HL_PEND_DEFERRED, false, true))) return rc;
        return p->defer_join ? 0 : hlvae_join(p, s);
    }
    // d W1 below is the LAST encoder Linear's gradient; its input is Xn or the output of the layers before it
    const bf16_t* w1_inT = d.n_xe > 0 ? ws->xe[d.n_xe - 1].aT : ws->xnT;
    // d W1 = dT^T Xn [h_e][X] (no input gradient for layer 1);  d Wd = dU^T z [h_d][L];  d [Wmu; Wlv] = dml^T T 2 x [L][h_e]
    GemmGroup g{};
    g.n = 1;
    g.K = Bp;
    g.p[0] = GemmProb{ws->dtT, w1_inT, ws->G + d.o_w1, nullptr, Bp, Bp, d.K1, d.h_e, d.K1, 0, 0};
    // (dims without hidden layers: Wd / [Wmu; Wlv] are identities nobody trains -- their gradients are not formed and stay zero)
    if (!d.lin_d) g.p[g.n++] = GemmProb{ws->duT, ws->zbT, ws->G + d.o_wd, nullptr, Bp, Bp, d.L, d.h_d0, d.L, 0, 0};
    if (!d.lin_e) g.p[g.n++] = GemmProb{ws->dmlT, ws->tT, ws->G + d.o_wmu, ws->G + d.o_wlv, Bp, Bp, d.h_e, 2 * d.Lp, d.h_e, d.Lp, d.L};
    // few output tiles and a long batch axis (a 64-feature model at 4096 rows: 24 tiles x 64 k-steps): split-K with fp32 atomics
    // into the (cleared) gradient slices -- they are neighbours in the arena: [Wd | Wmu | Wlv | W1]
    if (wy_in_group) {      // (rows of dY^T are in the head kernel's variable order: stored through the row map)
        g.p[g.n] = GemmProb{dylT, ws->uT, ws->G + d.o_wy, nullptr, Bp, Bp, d.h_d, d.NYl, d.h_d, 0, 0};
        g.p[g.n++].rowmap = p->wy_rowsrc_dev;
    }
    g.ksplit = wg_ksplit;
    if (g.ksplit > 1 && !clear_in_mid) HL_CHECK(hipMemsetAsync(ws->G + d.o_wd, 0, sizeof(float) * (size_t)clear_n, st));
    if ((rc = hl_launch_gemm_f32_group(g, "dW1_dWd_dWmu", st))) return rc;
    if (d.n_xe > 0) {   // deeper encoder: d a = (dT W1) .* (a > 0) for the layer below the last, and so on down to the inputs
        const int il = d.n_xe - 1;
        if ((rc = hl_launch_gemm_act(1, ws->dt, d.hep, ws->w1Ts, d.hep, Bp, d.K1p, d.hep, nullptr, d.K1, ws->xe[il].a, ws->xe[il].d, d.K1p,
                                     ws->xe[il].dT, Bp, B, ws->G + d.xe[il].o_b, "dT_hidden", st))) return rc;
        for (int i = il; i >= 0; --i) {
            const hlvae_layer& l = d.xe[i];
            if ((rc = hl_launch_gemm_f32(ws->xe[i].dT, Bp, i > 0 ? ws->xe[i - 1].aT : ws->xnT, Bp, ws->G + l.o_w, l.n_in, l.n_out, l.n_in, Bp,
                                         0, 0, nullptr, "dW_enc_hidden", st))) return rc;
            if (i > 0)
                if ((rc = hl_launch_gemm_act(1, ws->xe[i].d, l.n_out_p, ws->xe[i].wT, l.n_out_p, Bp, l.n_in_p, l.n_out_p, nullptr, l.n_in,
                                             ws->xe[i - 1].a, ws->xe[i - 1].d, l.n_in_p, ws->xe[i - 1].dT, Bp, B, ws->G + d.xe[i - 1].o_b,
                                             "dT_hidden", st))) return rc;
        }
    }
    const bool conv_opt = d.conv && opt != nullptr && !skip_wy;
    // a small y_layer (64-feature models: 0.16 M parameters): ONE optimiser launch at the end instead of y_layer's own on side 0 --
    // nothing to keep apart, and the side chain (dWy -> Adam -> gradient fold) was what the final launch waited for
    const bool small_wy = opt != nullptr && !d.conv && !skip_wy && (long)d.NYl * d.h_d <= 512l * 1024;
    if (d.conv) {   // the convolutional features receive a gradient: d feat = dT W1, then conv2 / conv1 / representation layer
        if (d.n_xe > 0) {   // deeper encoder: the features feed the FIRST extra layer, whose pre-activation gradient the loop above left
            if ((rc = hl_launch_gemm_f32(ws->xe[0].d, d.xe[0].n_out_p, ws->xe[0].wT, d.xe[0].n_out_p, ws->dfeat, d.Xep, Bp, d.Xe, d.xe[0].n_out_p,
                                         0, 0, nullptr, "dfeat", st))) return rc;
        } else if ((rc = hl_launch_gemm_f32(ws->dt, d.hep, ws->w1Ts, d.hep, ws->dfeat, d.Xep, Bp, d.Xe, d.hep, 0, 0, nullptr, "dfeat", st))) return rc;   // (K1 = Xe)
        if (conv_opt) HL_CHECK(hipEventRecord(p->ev[1], st));      // dense gradients final, W1's transposed shadow read
        if ((rc = hl_launch_conv_enc_bwd(p, ws, B, st))) return rc;
    }
    if (!skip_wy) {
        HL_CHECK(hipStreamWaitEvent(s0, p->ev[0], 0));
        if (d.conv)     // the batch-contiguous copy of d(y_layer output) that the weight-gradient GEMM reads
            if ((rc = hl_launch_transpose_bf16(ws->dyc, d.NYlp, ws->dycT, Bp, Bp, d.NYl, "dyc_transpose", s0))) return rc;
        // d Wy = dY^T U  [NYl][h_d]
        // (rows of dY^T are in the head kernel's variable order: the store maps them back to the master's rows)
        if (!wy_in_group)
            if ((rc = hl_launch_gemm_f32(dylT, Bp, ws->uT, Bp, ws->G + d.o_wy, d.h_d, d.NYl, d.h_d, Bp, 0, 0, nullptr, "dWy", s0,
                                         d.conv ? nullptr : p->wy_rowsrc_dev))) return rc;
        if (opt != nullptr && !small_wy) {       // takes no completion ticket: the final launch below is ordered behind it by the join
            HL_REQUIRE(ws->wys_next == nullptr, HLVAE_EINVAL, "backward_adam: ws->wys_next is honoured by the fused-optimiser step only "
                       "(hlvae_backward_adam_fused() says when)");
            HL_CHECK(hipStreamWaitEvent(s0, p->ev[2], 0));
            if ((rc = hl_adam_part(p, ws, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale, 0x01, 0,
                                   0u, "adam_wy_early", s0))) return rc;
            if (conv_opt) {     // convolutional model: the other dense matrices too, under the 56 us of the encoder's backward
                HL_CHECK(hipStreamWaitEvent(s0, p->ev[1], 0));
                if ((rc = hl_adam_part(p, ws, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale,
                                       hl_all_matrices(p) & ~0x01u, 0, 0u, "adam_dense_early", s0))) return rc;      // (extra layers too)
            }
        }
        // head-parameter / y_layer-bias gradients: the head kernel left per-row-block partial sums; they are folded into the arena
        // at the END of side 0 (only the final Adam launch, which joins side 0, consumes them).  At the head of this chain the
        // 4 us kernel started 40 us late in the replayed graph and took dWy and y_layer's Adam launch with it (rocprofv3 trace).
        if ((rc = hl_launch_head_grad_reduce(p, ws, Bp, s0))) return rc;
        HL_CHECK(hipEventRecord(p->ev[3], s0));
        if (wy_in_group)        // side 0 is otherwise idle here: the ELBO scalars + metrics behind the fold, so that the next batch's
            // input stage -- what the end of the step waits for -- has side 1 to itself (colstats + pack ended 6 us after the
            // final optimiser launch, 132 -> 122 us traced)
            if ((rc = hl_flush_deferred(p, s0, true, HL_PEND_FINALIZE | HL_PEND_METRICS, true))) return rc;
    }
    if ((p->pend_flags & HL_PEND_DEFERRED) && p->defer_join != 2) {
        HL_CHECK(hipStreamWaitEvent(s1, p->ev[direct_bwd ? 0 : 2], 0));     // (narrow y_layer: ev[2] is behind the fused middle)
        if ((rc = hl_flush_deferred(p, s1, true))) return rc;
    }
    if (!skip_wy) HL_CHECK(hipStreamWaitEvent(st, p->ev[3], 0));
    if (opt != nullptr)     // Adam of the other matrices + the small flat region; commits the step number.  Behind side 0
        // (two concurrent Adam launches thrash HBM), but NOT behind side 1: its deferred kernels only have to be done
        // by the end of the step
        if ((rc = conv_opt ? hl_adam_part(p, ws, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale, 0u, 1,
                                          hl_ticket_units(hl_adam_grid(p, ws, 0u, 1)), "adam_small", st)
                           : hl_adam(p, ws, opt->m1, opt->m2, opt->step_count, opt->lr, opt->b1, opt->b2, opt->eps, opt->gscale, st,
                                     (skip_wy || small_wy) ? 0 : 1)))
            return rc;
    // skip_wy (data-parallel host): the deferred side chain (metrics, next batch's input stage) stays un-joined -- the host's
    // reduce-scatters and optimiser launches that follow do not need it (they were starting 17 us late behind the input stage);
    // it calls hlvae_join at the end of its step
    if ((p->pend_flags & HL_PEND_DEFERRED) && p->defer_join == 2)       // in line, behind the last optimiser launch (see the fused path)
        if ((rc = hl_flush_deferred(p, st, true, HL_PEND_DEFERRED, false, true))) return rc;
    if ((skip_wy && opt == nullptr) || p->defer_join) return 0;
    return hlvae_join(p, s);
}

int hlvae_backward(const hlvae_plan* p, const hlvae_ws* ws, const float* g_mu, const float* g_lv, float kl_std_weight,
                   int skip_wy, int B, hlvae_stream s) {
    return hl_backward_impl(p, ws, g_mu, g_lv, kl_std_weight, skip_wy, B, s, nullptr);
}

int hlvae_backward_adam(const hlvae_plan* p, const hlvae_ws* ws, const float* g_mu, const float* g_lv, float kl_std_weight,
                        int B, float* m1, float* m2, int64_t* step_count, float lr, float beta1, float beta2, float eps,
                        float grad_scale, hlvae_stream s) {
    HL_REQUIRE(p && ws && m1 && m2 && step_count, HLVAE_EINVAL, "backward_adam: null argument");
    const HlAdamArgs a{m1, m2, step_count, lr, beta1, beta2, eps, grad_scale};
    return hl_backward_impl(p, ws, g_mu, g_lv, kl_std_weight, 0, B, s, &a);     // both optimiser launches are queued inside
}

int hlvae_adam_step(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr,
                    float beta1, float beta2, float eps, float grad_scale, hlvae_stream s) {
    HL_REQUIRE(p && ws && m1 && m2 && step_count, HLVAE_EINVAL, "adam_step: null argument");
    // two launches back to back (y_layer's weight, then the rest + the flat region, which commits the step number): 46 us for
    // the D4 model against 80 us as ONE launch over all five matrices.  A streaming kernel whose workgroups are all resident
    // at once runs in lock-step (everyone reads, then everyone writes); past one wave of workgroups the phases interleave
    // and HBM pays read/write turnarounds (the same effect makes two concurrent Adam launches crawl, hl_backward_impl).
    if (int rc = hl_adam_part(p, ws, m1, m2, step_count, lr, beta1, beta2, eps, grad_scale, 0x01, 0, 0u, "adam_wy_early", (hipStream_t)s))
        return rc;
    return hl_adam(p, ws, m1, m2, step_count, lr, beta1, beta2, eps, grad_scale, (hipStream_t)s, 1);
}

int hlvae_adam_shard(const hlvae_plan* p, const hlvae_ws* ws, const float* grad_shard, float* m1, float* m2, uint16_t* pb16_shard,
                     const int64_t* step_count, int64_t lo, int64_t n, float lr, float beta1, float beta2, float eps,
                     float grad_scale, hlvae_stream s) {
    HL_REQUIRE(p && ws && grad_shard && m1 && m2 && pb16_shard && step_count, HLVAE_EINVAL, "adam_shard: null argument");
    // the kernel indexes master / m / v by arena element and the two compact buffers from 0
    return hl_adam_flat(p, ws, grad_shard, m1, m2, pb16_shard - lo, step_count, (long)lo, (long)n, lr, beta1, beta2, eps, grad_scale,
                        (hipStream_t)s);
}

int hlvae_adam_small(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr, float beta1,
                     float beta2, float eps, float grad_scale, hlvae_stream s) {
    HL_REQUIRE(p && ws && m1 && m2 && step_count, HLVAE_EINVAL, "adam_small: null argument");
    return hl_adam_part(p, ws, m1, m2, step_count, lr, beta1, beta2, eps, grad_scale, 0u, 1, hl_ticket_units(hl_adam_grid(p, ws, 0u, 1)),
                        "adam_small", (hipStream_t)s);
}

int hlvae_shadows_from_bf16(const hlvae_plan* p, const hlvae_ws* ws, const uint16_t* pb16, int64_t base, unsigned which,
                            hlvae_stream s) {
    HL_REQUIRE(p && ws && pb16, HLVAE_EINVAL, "shadows_from_bf16: null argument");
    HL_REQUIRE(which != 0 && (which & ~0x7ffu) == 0 && base % 4 == 0, HLVAE_EINVAL, "shadows_from_bf16: which=0x%x base=%ld", which,
               (long)base);
    return hl_shadows_from_bf16(p, ws, pb16 - base, which, which == 0x01 ? "shadows_wy" : "shadows_rest", (hipStream_t)s);
}

int hlvae_gemm_nt_f32(const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc, int M, int N, int K,
                      hlvae_stream s) {
    HL_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, HLVAE_EINVAL, "gemm: bad arguments");
    return hl_launch_gemm_f32(A, lda, B, ldb, C, ldc, M, N, K, 0, 0, nullptr, "gemm_nt_f32", (hipStream_t)s);
}

}  // extern "C"
