// Stand-alone micro-benchmark of the dense kernels (no Python): old register-staged core vs the LDS-DMA core, on the shapes of
// BASELINE configs[1] (batch 512).  Build + run on the GPU box:  bash tools/ubench/run.sh
// Every variant runs on the same inputs; outputs are compared; times are HIP-event medians over interleaved rounds with the
// optimiser state rotated through NSETS copies (> 256 MiB in total) so that every launch reads it from HBM as the real step does.
#include <algorithm>
#include <random>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../hl-vae_amd/csrc/dense.hip"

static char g_err[512];
void hl_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
void hl_prof_begin(const char*, hipStream_t) {}
void hl_prof_end(hipStream_t) {}
unsigned long long* hl_stamp_slot(int) { return nullptr; }
#ifdef UB_OLD_TREE
#define UB_TICK_NULL
#else
#define UB_TICK_NULL , nullptr
#endif
#ifdef UB_OLD_TREE          // built against the round-2 sources (.old/): one kernel, no switches
static const int NV = 1;
static const char* VN[5] = {"round-2 tree 64x64", "", "", "", ""};
static void set_variant(int) {}
static int g_hl_gemm_dma = 0;
#else
static const int NV = 3;
extern int g_hl_gemm_dma;
extern int g_hl_adam_stagger;
static const char* VN[5] = {"register-staged core", "LDS-DMA core", "LDS-DMA core, stagger", "", ""};
static void set_variant(int v) { g_hl_gemm_dma = v > 0; g_hl_adam_stagger = v >= 2; }
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class T> T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static std::mt19937 rng(1234);
static bf16_t* rand_bf16(size_t n, float scale) {
    std::vector<bf16_t> h(n);
    std::normal_distribution<float> nd(0.f, scale);
    for (auto& x : h) { float f = nd(rng); uint32_t u; memcpy(&u, &f, 4); x = (bf16_t)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
    bf16_t* d = dalloc<bf16_t>(n);
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static float* rand_f32(size_t n, float scale, bool positive = false) {
    std::vector<float> h(n);
    std::normal_distribution<float> nd(0.f, scale);
    for (auto& x : h) { x = nd(rng); if (positive) x = x * x; }
    float* d = dalloc<float>(n);
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}
template <class T> static std::vector<T> d2h(const T* d, size_t n) { std::vector<T> h(n); CK(hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost)); return h; }
static double maxdiff(const std::vector<float>& a, const std::vector<float>& b) { double m = 0; for (size_t i = 0; i < a.size(); ++i) m = std::max(m, (double)fabsf(a[i] - b[i])); return m; }
static size_t ndiff16(const std::vector<bf16_t>& a, const std::vector<bf16_t>& b) { size_t n = 0; for (size_t i = 0; i < a.size(); ++i) n += a[i] != b[i]; return n; }
static double med(std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F> float run(F f) { CK(hipEventRecord(a, 0)); f(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms * 1e3f; }
};

int main(int argc, char** argv) {
    const int Bp = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = 25;
    Timer T;
    // ---------------- y_layer weight gradient + Adam:  dYT [6480(6528)][Bp] x UT [500(512)][Bp] -> [6480][500] -----------------
    {
        const int M = 6480, N = 500, Mp = 6528, Np = 512;
        const int NSETS = 7;                                  // 7 x 39 MB of state + shadows
        bf16_t* dyT = rand_bf16((size_t)Mp * Bp, 0.05f);
        bf16_t* uT = rand_bf16((size_t)Np * Bp, 0.5f);
        const size_t arena = (size_t)M * N + 64;
        std::vector<float*> P(NSETS), M1(NSETS), M2(NSETS);
        std::vector<bf16_t*> sh(NSETS), shT(NSETS);
        for (int i = 0; i < NSETS; ++i) {
            P[i] = rand_f32(arena, 0.05f); M1[i] = rand_f32(arena, 0.01f); M2[i] = rand_f32(arena, 0.01f, true);
            sh[i] = dalloc<bf16_t>((size_t)Mp * Np); shT[i] = dalloc<bf16_t>((size_t)Np * Mp);
        }
        int64_t* step = dalloc<int64_t>(2);
        auto launch = [&](int set, int dma) {
            AdamGemmGroup g{};
            g.n = 1; g.K = Bp;
            g.p[0] = AdamGemmProb{dyT, uT, nullptr, sh[set], shT[set], 0, 0, Bp, Bp, M, N, 0, 0, Np, Mp, 0, 0, 0};
            set_variant(dma);
            if (hl_launch_gemm_adam(g, P[set], M1[set], M2[set], step, 1e-3f, 0.9f, 0.999f, 1e-8f, 1.f, 0u, "x", 0, nullptr, 0, 0 UB_TICK_NULL)) { printf("launch failed: %s\n", g_err); exit(1); }
        };
        // correctness: set 2 keeps a pristine copy of set 0's state; set 1 is reloaded from it for every variant
        CK(hipMemcpy(P[2], P[0], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M1[2], M1[0], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M2[2], M2[0], arena * 4, hipMemcpyDeviceToDevice));
        launch(0, 0);
        for (int v = 1; v < NV; ++v) {
            CK(hipMemcpy(P[1], P[2], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M1[1], M1[2], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M2[1], M2[2], arena * 4, hipMemcpyDeviceToDevice));
            launch(1, v); CK(hipDeviceSynchronize());
            printf("gemm_adam wy  old vs %s: max|dP| %.3g  max|dM1| %.3g  max|dM2| %.3g  shadow cells differing %zu  shadowT %zu\n", VN[v],
                   maxdiff(d2h(P[0], arena), d2h(P[1], arena)), maxdiff(d2h(M1[0], arena), d2h(M1[1], arena)), maxdiff(d2h(M2[0], arena), d2h(M2[1], arena)),
                   ndiff16(d2h(sh[0], (size_t)Mp * Np), d2h(sh[1], (size_t)Mp * Np)), ndiff16(d2h(shT[0], (size_t)Np * Mp), d2h(shT[1], (size_t)Np * Mp)));
        }
        std::vector<float> t[5];
        int set = 0;
        for (int r = 0; r < reps; ++r)
            for (int v = 0; v < NV; ++v) { set = (set + 1) % NSETS; t[v].push_back(T.run([&] { launch(set, v); })); }
        const double bytes = (double)(M + N) * Bp * 2 + 24.0 * M * N + 2.0 * 2 * M * N;
        for (int v = 0; v < NV; ++v) printf("gemm_adam wy  %s: median %.2f us  min %.2f  -> %.2f TB/s algorithmic (%.1f MB)\n", VN[v], med(t[v]), *std::min_element(t[v].begin(), t[v].end()), bytes / med(t[v]) * 1e-6, bytes * 1e-6);
        // ---- as bench.py launches it: master rows through a row map (the head kernel's variable order), completion tickets, state
        // NOT rotated (one set: what the Infinity Cache keeps between two steps) -- each switch on its own
        {
            std::vector<int32_t> hmap(Mp);
            for (int r = 0; r < Mp; ++r) hmap[r] = r;
            // D4-like grouping by kind: variables (5 rows each) in runs of 18 are pulled to the front
            { std::vector<int32_t> a, b; for (int d = 0; d < M / 5; ++d) (((d / 18) & 1) ? b : a).push_back(d);
              int r = 0; for (int d : a) for (int k = 0; k < 5; ++k) hmap[r++] = d * 5 + k; for (int d : b) for (int k = 0; k < 5; ++k) hmap[r++] = d * 5 + k; }
            int32_t* dmap = dalloc<int32_t>(Mp);
            CK(hipMemcpy(dmap, hmap.data(), Mp * 4, hipMemcpyHostToDevice));
#ifndef UB_OLD_TREE
            unsigned long long* shards = dalloc<unsigned long long>(HL_TICK_WORDS);
            const int NCFG = 12;
#else
            const int NCFG = 8;
#endif
            for (int cfg = 0; cfg < NCFG; ++cfg) {
                const bool two_level = cfg >= 8;
                const bool use_map = cfg & 1, use_tick = two_level || (cfg & 2), rotate = two_level ? !(cfg & 2) : !(cfg & 4);
                for (int v = 0; v < NV; ++v) {
                    std::vector<float> tt;
                    for (int r = 0; r < reps; ++r) {
                        set = rotate ? (set + 1) % NSETS : 3;
                        AdamGemmGroup g{};
                        g.n = 1; g.K = Bp;
                        g.p[0] = AdamGemmProb{dyT, uT, use_map ? dmap : nullptr, sh[set], shT[set], 0, 0, Bp, Bp, M, N, 0, 0, Np, Mp, 0, 0, 0};
                        set_variant(v);
#ifndef UB_OLD_TREE
                        const unsigned tick = use_tick ? (two_level ? hl_ticket_units(hl_gemm_adam_grid(g)) : (unsigned)hl_gemm_adam_grid(g)) : 0u;
                        tt.push_back(T.run([&] { if (hl_launch_gemm_adam(g, P[set], M1[set], M2[set], step, 1e-3f, 0.9f, 0.999f, 1e-8f, 1.f, tick, "x", 0, nullptr, 0, 0, two_level ? shards : nullptr)) { printf("launch failed: %s\n", g_err); exit(1); } }));
#else
                        const unsigned tick = use_tick ? (unsigned)hl_gemm_adam_grid(g) : 0u;
                        tt.push_back(T.run([&] { if (hl_launch_gemm_adam(g, P[set], M1[set], M2[set], step, 1e-3f, 0.9f, 0.999f, 1e-8f, 1.f, tick, "x", 0, nullptr, 0, 0)) { printf("launch failed: %s\n", g_err); exit(1); } }));
#endif
                    }
                    printf("gemm_adam wy  %-18s rowmap %d tickets %s state %s: median %.2f us (min %.2f)\n", VN[v], (int)use_map, two_level ? "2-level" : (use_tick ? "1-level" : "off    "), rotate ? "rotated (HBM)" : "one set (warm) ", med(tt), *std::min_element(tt.begin(), tt.end()));
                }
            }
        }
        // the other three in one launch: dW1 [500][5184], dWd [500][32], d[Wmu;Wlv] [64][500]
        {
            const int h = 500, X = 5184, L = 32;
            bf16_t* dtT = rand_bf16((size_t)512 * Bp, 0.05f);
            bf16_t* xnT = rand_bf16((size_t)X * Bp, 0.5f);
            bf16_t* duT = rand_bf16((size_t)512 * Bp, 0.05f);
            bf16_t* zbT = rand_bf16((size_t)64 * Bp, 0.5f);
            bf16_t* dmlT = rand_bf16((size_t)64 * Bp, 0.05f);
            bf16_t* tT = rand_bf16((size_t)512 * Bp, 0.5f);
            bf16_t* w1s = dalloc<bf16_t>((size_t)512 * X); bf16_t* wds = dalloc<bf16_t>(512 * 32); bf16_t* wdTs = dalloc<bf16_t>(32 * 512);
            bf16_t* wmls = dalloc<bf16_t>(64 * 512); bf16_t* wmlTs = dalloc<bf16_t>(512 * 64);
            const long o_w1 = 0, o_wd = (long)h * X, o_wmu = o_wd + h * L, o_wlv = o_wmu + L * h;      // all inside the wy arena (3.24 M floats)
            auto launch3 = [&](int set, int dma) {
                AdamGemmGroup g{};
                g.n = 3; g.K = Bp;
                g.p[0] = AdamGemmProb{dtT, xnT, nullptr, w1s, nullptr, o_w1, 0, Bp, Bp, h, X, 0, 0, X, 0, 0, 0, 0};
                g.p[1] = AdamGemmProb{duT, zbT, nullptr, wds, wdTs, o_wd, 0, Bp, Bp, h, L, 0, 0, 32, 512, 0, 0, 0};
                g.p[2] = AdamGemmProb{dmlT, tT, nullptr, wmls, wmlTs, o_wmu, o_wlv, Bp, Bp, 64, h, 32, L, 512, 64, 0, 0, 0};
                set_variant(dma);
                if (hl_launch_gemm_adam(g, P[set], M1[set], M2[set], step, 1e-3f, 0.9f, 0.999f, 1e-8f, 1.f, 0u, "x", 0, nullptr, 0, 0 UB_TICK_NULL)) { printf("launch failed: %s\n", g_err); exit(1); }
            };
            CK(hipMemcpy(P[2], P[0], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M1[2], M1[0], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M2[2], M2[0], arena * 4, hipMemcpyDeviceToDevice));
            launch3(0, 0); auto w_old = d2h(w1s, (size_t)512 * X);
            for (int v = 1; v < NV; ++v) {
                CK(hipMemcpy(P[1], P[2], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M1[1], M1[2], arena * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(M2[1], M2[2], arena * 4, hipMemcpyDeviceToDevice));
                launch3(1, v); CK(hipDeviceSynchronize());
                printf("gemm_adam rest old vs %s: max|dP| %.3g  max|dM1| %.3g  shadow cells differing %zu\n", VN[v], maxdiff(d2h(P[0], arena), d2h(P[1], arena)),
                       maxdiff(d2h(M1[0], arena), d2h(M1[1], arena)), ndiff16(w_old, d2h(w1s, (size_t)512 * X)));
            }
            std::vector<float> t3[5];
            for (int r = 0; r < reps; ++r)
                for (int v = 0; v < NV; ++v) { set = (set + 1) % NSETS; t3[v].push_back(T.run([&] { launch3(set, v); })); }
            const double b3 = (double)(h + X) * Bp * 2 + 26.0 * h * X + (double)(h + L) * Bp * 2 + (double)(64 + h) * Bp * 2 + 28.0 * (h * L + 2 * L * h);
            for (int v = 0; v < NV; ++v) printf("gemm_adam rest %s: median %.2f us  min %.2f  -> %.2f TB/s algorithmic (%.1f MB)\n", VN[v], med(t3[v]), *std::min_element(t3[v].begin(), t3[v].end()), b3 / med(t3[v]) * 1e-6, b3 * 1e-6);
        }
    }
    // ---------------- split-K products ------------------------------------------------------------------------------------------
    for (int which = 0; which < 2; ++which) {
        const int M = Bp, N = 512, K = which ? 6528 : 5184, S = 8;
        bf16_t* A = rand_bf16((size_t)M * K, 0.5f);
        bf16_t* B = rand_bf16((size_t)N * K, 0.05f);
        float* slab[2] = {dalloc<float>((size_t)S * M * N), dalloc<float>((size_t)S * M * N)};
        bf16_t* flush = dalloc<bf16_t>((size_t)160 << 20);
        auto launch = [&](int dma) {
            g_hl_gemm_dma = dma;
            if (hl_launch_gemm_splitk(A, K, B, K, slab[dma], N, M, N, K, S, "x", 0)) { printf("launch failed: %s\n", g_err); exit(1); }
        };
        launch(0); if (NV > 1) launch(1); CK(hipDeviceSynchronize());
        auto s0 = d2h(slab[0], (size_t)S * M * N), s1 = d2h(slab[NV > 1 ? 1 : 0], (size_t)S * M * N);
        double mx = 0; for (auto x : s0) mx = std::max(mx, (double)fabsf(x));
        printf("splitk %s old vs dma: max|d| %.3g (max |value| %.3g)\n", which ? "dU " : "enc1", maxdiff(s0, s1), mx);
        std::vector<float> t[2], tw[2];
        for (int r = 0; r < reps; ++r)
            for (int v = 0; v < (NV > 1 ? 2 : 1); ++v) {
                CK(hipMemsetAsync(flush, r, (size_t)320 << 20, 0));          // operands and slabs out of L2 / Infinity Cache
                t[v].push_back(T.run([&] { launch(v); }));
                tw[v].push_back(T.run([&] { launch(v); }));                   // and once more, warm
            }
        for (int v = 0; v < (NV > 1 ? 2 : 1); ++v) printf("splitk %s %s: cold median %.2f us (min %.2f)   warm median %.2f us (min %.2f)\n", which ? "dU " : "enc1", v ? "dma" : "old", med(t[v]),
                                           *std::min_element(t[v].begin(), t[v].end()), med(tw[v]), *std::min_element(tw[v].begin(), tw[v].end()));
    }
    return 0;
}
