/*
 * hlvae_hip.h -- C ABI of libhlvae_hip.so: the MI355X (gfx950) HIP kernels of the HL-VAE ELBO
 * training hot path.  Plain pointers and sizes only; no torch types.
 *
 * The reference (MineOgre/HL-VAE) is 19 Python files with NO native/FFI boundary,
 * so there is no existing FFI to bind.  Each entry point below replaces the Python-level stage of
 * the reference that is cited next to it (paths relative to the reference repo); the host side
 * that calls them mirrors the reference's class/function names (hl-vae_amd/HLVAE.py etc.).
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller; nothing here allocates or frees caller
 *     memory.  The only library-owned device memory is the small per-plan tables.
 *   - every call is asynchronous on the given hipStream_t (pass torch's current stream);
 *     no call synchronises the device, so all of them can be captured into a hipGraph.
 *   - return value 0 = success, otherwise a hipError_t (>0) or a negative HLVAE_E* code;
 *     hlvae_last_error() returns a human-readable message for the calling thread.
 *   - bf16 buffers are raw uint16_t.  "p" suffixed sizes are padded sizes, see hlvae_dims.
 */
#ifndef HLVAE_HIP_H
#define HLVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HLVAE_ABI_VERSION 35
#define HLVAE_STAT_CHUNKS 16
/* accumulators per variable in ws->hgpart for the largest head instance: (y_dim + 1) (K - 1) + y_dim with K <= 16, y_dim = 5 */
#define HLVAE_HEAD_ACC 95

#define HLVAE_EINVAL (-1)   /* bad argument / unsupported configuration */
#define HLVAE_ESHAPE (-2)   /* operand shapes do not match what the kernel grid assumes */

/* variable kinds (column plan, hl-vae_amd/layout.py; reference HL_VAE/loglik.py function names) */
enum { HLVAE_REAL = 0, HLVAE_POS = 1, HLVAE_COUNT = 2, HLVAE_CAT = 3, HLVAE_ORDINAL = 4 };

/* one row per variable d (reference types_info, HL_VAE/read_functions.py:142-198) */
typedef struct {
    int32_t kind;    /* HLVAE_REAL ...                                              */
    int32_t ncls;    /* K for cat / ordinal, 1 otherwise                             */
    int32_t xoff;    /* first column in the expanded data / parameter matrix [B, X]  */
    int32_t sidx;    /* row in the batch-statistic vectors (reals first, then pos)   */
    int32_t w_off;   /* arena offset of the head weight   [y_dim][a]  (HLVAE.py:14,37,57,74) */
    int32_t b_off;   /* arena offset of the head bias     [a]                        */
    int32_t e_off;   /* arena offset of _log_vy_{real,pos}[i] or ordinal thresholds[K-1], -1 if none */
    int32_t r_off;   /* conv only: arena offset of representation_layer weight[d][K] (HLVAE.py:94), -1 otherwise */
    int32_t rb_off;  /* conv only: arena offset of representation_layer bias[d], -1 otherwise                     */
    int32_t poff;    /* first column in the likelihood-parameter matrix [B, Theta] (= xoff unless logvar_network) */
    int32_t poff2;   /* logvar_network, real / pos: column of the variance parameter, -1 otherwise                  */
    int32_t w2_off;  /* logvar_network, real / pos: arena offset of weight_logvar [y_dim] (HLVAE.py:31-37), -1 otherwise */
    int32_t b2_off;  /* logvar_network, real / pos: arena offset of bias_logvar, -1 otherwise                      */
    int32_t pad;
} hlvae_var;

/* one extra hidden Linear + ReLU of a deeper trunk (dims[1] / dims[3] with more than one entry, HLVAE.py:125-137, 232-242) */
#define HLVAE_MAX_EXTRA 3
typedef struct {
    int32_t n_in, n_out;         /* nn.Linear(n_in, n_out): weight [n_out][n_in]                                        */
    int32_t n_in_p, n_out_p;     /* derived: padded to multiples of 64                                                  */
    int64_t o_w, o_b;            /* arena offsets: the weight in the dense region [o_xw, o_wy), the bias in the atomic one */
} hlvae_layer;

/* model geometry.  Padded sizes are derived by hlvae_dims_fill(). */
typedef struct {
    int32_t D, X, y_dim, h_e, h_d, L;          /* reference dims = [X, [.., h_e], L, reversed [h_d0, .., h_d], y_dim], D = n_variables:
                                                  h_e = width of the LAST encoder layer (it feeds mean_layer / log_var_layer),
                                                  h_d = width of the LAST decoder layer (y_layer's input) */
    int32_t n_real, n_pos;
    int32_t conv;                              /* 1 = convolutional front / back end (HLVAE.py:139-152, 253-259): D must be 36 * 36 */
    int32_t Theta;                             /* columns of the likelihood-parameter matrix: X, or X + n_real + n_pos under logvar_network */
    /* derived */
    int32_t Xp, hep, hdp, Lp, NY, NYp, n_stat;
    int32_t Xe, Xep;                           /* width of the first encoder Linear's input: X, or 32*9*9 under conv            */
    int32_t NYl, NYlp;                         /* width of y_layer's output: D * y_dim, or 32*9*9 under conv                     */
    /* arena offsets (floats) of the dense layers, reference names in comments */
    int64_t o_w1, o_b1;          /* VAE_encoder_common_layers.0.{weight [h_e,X], bias}   (HLVAE.py:131) */
    int64_t o_wmu, o_bmu;        /* mean_layer.0        [L,h_e]                           (HLVAE.py:168) */
    int64_t o_wlv, o_blv;        /* log_var_layer.0     [L,h_e]                           (HLVAE.py:174) */
    int64_t o_wd, o_bd;          /* hidden.0 / d_layers.0 [h_d,L]                         (HLVAE.py:236) */
    int64_t o_wy, o_by;          /* y_layer.0           [NYl,h_d]                         (HLVAE.py:246-248) */
    int64_t o_c1w, o_c1b;        /* conv1  [16,1,3,3]   (HLVAE.py:147)  -- conv only, inside the atomic region */
    int64_t o_c2w, o_c2b;        /* conv2  [32,16,3,3]  (HLVAE.py:151)                                         */
    int64_t o_t1w, o_t1b;        /* deconv_layer.0  ConvTranspose2d [32,16,4,4]  (HLVAE.py:255)                */
    int64_t o_t2w, o_t2b;        /* deconv_layer.2  ConvTranspose2d [16,y_dim,4,4]  (HLVAE.py:257-258)         */
    int64_t o_cv_lo, cv_n;       /* conv only: arena range [o_cv_lo, o_cv_lo + cv_n) holding the representation layers, the four
                                    convolution layers and y_layer's bias -- gradients produced by csrc/conv.hip */
    int64_t arena_size;          /* floats */
    int64_t atomic_region;       /* grads in [0, atomic_region) are accumulated with atomics and
                                    must be zero when a backward pass starts (hlvae_backward zeroes them) */
    int64_t frozen_lo, frozen_hi; /* arena range inside [0, atomic_region) that the optimiser leaves alone (vy_fixed = True:
                                    _log_vy_real / _log_vy_pos without requires_grad, HLVAE.py:209-216; torch.optim.Adam skips
                                    parameters without a gradient); frozen_lo == frozen_hi: none */
    /* deeper trunks.  The fused kernels keep their single-hidden-layer shape: "W1" (o_w1) is the LAST encoder layer, with
     * input width K1 (X for one layer), "Wd" (o_wd) the FIRST decoder layer [h_d0][L]; the layers in between run as plain
     * GEMM + ReLU launches.  n_xe = n_xd = 0 and h_d0 = 0 describe the reference configuration ([500] / [500]). */
    int32_t n_xe, n_xd;          /* encoder layers before the last one, decoder layers after the first one (<= HLVAE_MAX_EXTRA) */
    int32_t h_d0;                /* width of the first decoder layer; 0 = h_d                                            */
    int32_t K1, K1p, hd0p;       /* derived: input width of the last encoder layer (Xe, or xe[n_xe-1].n_out); padded sizes */
    hlvae_layer xe[HLVAE_MAX_EXTRA];   /* VAE_encoder_common_layers.{0, 2, ..}: X -> .. (all but the last Linear)         */
    hlvae_layer xd[HLVAE_MAX_EXTRA];   /* d_layers.{2, 4, ..}: h_d0 -> .. -> h_d                                          */
    int64_t o_xw;                /* start of the extra layers' weights in the arena (they sit between W1 and Wy); o_wy when none */
    /* dims without hidden layers (reference HLVAE.py:128, 233: h_dim = [] / 0).  lin_e: "W1" [h_e = 2L][X] IS [mean_layer.weight;
     * log_var_layer.weight] (b1 the two biases), its activation is linear, and [Wmu; Wlv] / bmu / blv hold an identity / zeros
     * that the library never trains (their gradients are not formed).  lin_d: "Wd" [h_d0 = L][L] is an identity, bd zeros, the
     * activation linear: y_layer reads the latent.  The fused-optimiser step is not used with either. */
    int32_t lin_e, lin_d;
} hlvae_dims;

void hlvae_dims_fill(hlvae_dims* d);   /* fills the derived fields from D,X,y_dim,h_e,h_d,L,n_real,n_pos */

/* buffers of one extra hidden layer */
typedef struct {
    uint16_t* w;  uint16_t* wT;      /* bf16 shadows of the weight [n_out_p][n_in_p] and its transpose [n_in_p][n_out_p]   */
    uint16_t* a;  uint16_t* aT;      /* ReLU output [Bp][n_out_p], [n_out_p][Bp]                                           */
    uint16_t* d;  uint16_t* dT;      /* gradient with respect to the pre-activation, same shapes                           */
} hlvae_layer_ws;

/* Workspace: every device buffer one training / inference step touches.  Allocated by the caller
 * (hl-vae_amd/HLVAE.py) for a maximum padded batch Bp_max (multiple of 128).  Padding rows/columns
 * are kept zero by the kernels that produce a buffer. */
typedef struct {
    int32_t Bp_max;
    int32_t splitk_enc, splitk_dec;   /* split-K factors of the two long-K GEMMs */
    /* parameters */
    float* P;            /* fp32 master parameters, reference shapes, flat arena           */
    float* G;            /* fp32 gradients, same layout                                     */
    uint16_t* w1s;       /* bf16 shadow  [hep][Xp]   of W1                                  */
    uint16_t* wmls;      /* bf16 shadow  [2Lp][hep]  rows 0..L-1 = Wmu, Lp..Lp+L-1 = Wlv    */
    uint16_t* wmlTs;     /* its transpose [hep][2Lp]                                        */
    uint16_t* wds;       /* bf16 shadow  [hdp][Lp]   of Wd                                  */
    uint16_t* wdTs;      /* [Lp][hdp]                                                       */
    uint16_t* wys;       /* bf16 shadow  [NY][hdp]   of Wy                                  */
    uint16_t* wyTs;      /* [hdp][NYp]                                                      */
    /* batch statistics (row A) */
    double* sums;        /* [HLVAE_STAT_CHUNKS][3][n_stat] per-row-chunk partial sums of m, x m, x^2 m (x = d or log1p d) */
    float* norm;         /* [2][n_stat]  mean, var (var of pos already clamped to [1e-6,1e20]) */
    /* packed inputs */
    uint16_t* xn;        /* [Bp][Xp]  normalised encoder input, bf16                        */
    uint16_t* xnT;       /* [Xp][Bp]                                                        */
    float* xt;           /* [Bp][D]   likelihood target: raw x | log1p x | class index      */
    uint8_t* m8;         /* [Bp][D]   1 = observed                                          */
    /* activations */
    float* slab;         /* [max(splitk)][Bp][max(hep,hdp)] fp32 split-K partials           */
    uint16_t* t;  uint16_t* tT;      /* encoder trunk  [Bp][hep], [hep][Bp]                 */
    float* mu; float* lv; float* z;  /* [Bp][L] fp32 (row stride L)                         */
    uint16_t* zb; uint16_t* zbT;     /* [Bp][Lp], [Lp][Bp]                                  */
    uint16_t* u;  uint16_t* uT;      /* decoder trunk  [Bp][hdp], [hdp][Bp]                 */
    uint16_t* dy; uint16_t* dyT;     /* d loss / d Y   [Bp][NYp], [NY][Bp]                  */
    float* log_p_x; float* log_p_x_missing;   /* [Bp][D]                                    */
    float* rowpart;      /* [ceil(D/16)][Bp] partial row sums of log_p_x                    */
    float* hgpart;       /* [Bp/64][ceil(D/16)*16][HLVAE_HEAD_ACC] per-row-block partial sums of the head-parameter and y_layer-bias
                            gradients (k_y_heads); folded into ws->G by the backward pass                             */
    float* nll;          /* [Bp]  -sum_d log_p_x                                            */
    double* scal;        /* [8]: 0 = sum_b nll, 1 = KL(q || N(0,I)) of the batch (extension), 2.. reserved */
    double* klpart;      /* [Bp/4] KL partial sums, one per 4 rows (k_mid_fwd_fused)            */
    float* eps;          /* [Bp][L] reparameterisation noise actually used (kept for backward) */
    uint64_t* rng;       /* [2]: Philox seed, offset (advanced by one per step on device)    */
    float* pfull;        /* [Bp][Theta]  likelihood parameters concatenated by key (row M), optional */
    float* xhat;         /* [Bp][D]  per-variable imputed value (statistics mean), optional */
    float* metpart;      /* [16][6][D] partials of the row-M metrics kernel                  */
    /* backward activations */
    uint16_t* du; uint16_t* duT;     /* [Bp][hdp], [hdp][Bp]                                */
    float* dz;                       /* [Bp][Lp]                                            */
    uint16_t* dml; uint16_t* dmlT;   /* [Bp][2Lp], [2Lp][Bp]                                */
    uint16_t* dt; uint16_t* dtT;     /* [Bp][hep], [hep][Bp]                                */
    /* convolutional front / back end (NULL otherwise) */
    uint16_t* w1Ts;      /* bf16 shadow [K1p][hep] of W1^T (input gradient of that Linear: conv, or n_xe > 0) */
    uint16_t* cpack;     /* packed bf16 convolution weights (csrc/conv.hip, CP_TOTAL elements)          */
    float* img;          /* [Bp][1296] the one-number-per-variable image the encoder convolves          */
    uint16_t* yc;        /* [Bp][NYlp] y_layer output (bf16), viewed as [32][9][9]                      */
    uint16_t* a2;        /* [Bp][18*18][16] ReLU(deconv 1), channel-last                                */
    float* yv;           /* [Bp][NY]  deconv 2 output = y_grouped[b][pixel][c], fp32                    */
    uint16_t* da2;       /* [Bp][18*18][16]                                                             */
    uint16_t* dyc; uint16_t* dycT;   /* d yc [Bp][NYlp], [NYl][Bp]                                      */
    float* dfeat;        /* [Bp][Xep] gradient of the 2592 convolutional features                       */
    float* dimg;         /* [Bp][1296] gradient of img                                                  */
    float* cvpart;       /* [512][cv_n] per-workgroup partial gradients of the arena range above; padding entries must be
                            zero at allocation (the kernels write every real entry each step, never the padding) */
    /* deeper trunks (NULL for one hidden layer per side, except u0 / u0T) */
    uint16_t* u0; uint16_t* u0T;     /* output of the FIRST decoder layer [Bp][hd0p], [hd0p][Bp]; == u, uT when n_xd == 0  */
    hlvae_layer_ws xe[HLVAE_MAX_EXTRA];
    hlvae_layer_ws xd[HLVAE_MAX_EXTRA];   /* xd[n_xd-1].a / .aT must be u / uT (y_layer's input)                          */
    /* optional second pair of y_layer shadows: hlvae_backward_adam writes the UPDATED shadows there instead of in place, so
     * that its y_layer launch need not wait for this step's last reader of wys / wyTs; the caller then passes them as wys /
     * wyTs of the next step (and these two as that step's *_next).  NULL: in place.  (Below 2048 rows the launch starts behind that
     * reader anyway, so that the fused middle kernel is resident first: the second pair then changes nothing.) */
    uint16_t* wys_next; uint16_t* wyTs_next;
} hlvae_ws;

typedef struct hlvae_plan hlvae_plan;
typedef void* hlvae_stream;      /* hipStream_t */

int  hlvae_abi_version(void);
const char* hlvae_last_error(void);
/* sizeof(hlvae_dims), sizeof(hlvae_var), sizeof(hlvae_ws): lets a foreign-language binding verify its struct layout */
void hlvae_struct_sizes(int32_t* dims_bytes, int32_t* var_bytes, int32_t* ws_bytes);

/* replaces: building types_info index vectors per step (HL_VAE/utils.py:94-96, HLVAE.py:387-410).
 * var_order (host int32 [D], may be NULL = the variables' own order): the order in which the HEAD KERNEL walks the variables,
 * var_order[j] = variable at kernel position j.  The reference fixes the external order (read_functions.py:142-198); the
 * kernel-facing one is free, and grouping the variables by kind makes the 16-variable tiles of the head kernel homogeneous
 * (one likelihood body per wavefront instead of up to five: 38.8 -> 24.3 us on the interleaved 64-feature mix at 4096 rows).
 * The bf16 shadows of y_layer's weight and dY are laid out in that order (ws->wys, wyTs, dy, dyT); masters, gradients and
 * every [B, D] buffer keep the variables' own order. */
int  hlvae_plan_create(hlvae_plan** out, const hlvae_dims* dims, const hlvae_var* vars /* host, [D] */, const int32_t* var_order);
void hlvae_plan_destroy(hlvae_plan* p);

/* The same input stage from the COMPACT device-resident dataset (csrc/feed.hip; replaces the per-row pandas gather of
 * dataset_def.py:67-92 and the 93 kB/row fp64 transfer): values fp32 [N][D] = raw value (real / pos / count), class index
 * (cat, -1 = none) or level - 1 (ordinal); mask8 u8 [N][D], 1 = observed; rows int32 [B] = dataset rows of this batch.
 * Outputs are those of hlvae_normalize_stats / hlvae_normalize_pack on the expanded fp64 form of the same rows. */
int  hlvae_feed_stats(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                      int B, hlvae_stream s);
int  hlvae_feed_pack(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                     int B, hlvae_stream s);

/* statistics + pack in one call for single-process training (no statistics all-reduce between the passes); skips the
 * statistics pass when no column needs it (convolutional model without pos variables). */
int  hlvae_normalize_fused(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, int B, hlvae_stream s);
int  hlvae_feed_fused(const hlvae_plan* p, const hlvae_ws* ws, const float* values, const uint8_t* mask8, const int32_t* rows,
                      int B, hlvae_stream s);

/* bf16 shadows of the dense weights from the fp32 arena (no reference counterpart: the reference
 * computes in fp64; BASELINE.json config 2 asks for bf16 encoder/decoder) */
int hlvae_refresh_shadows(const hlvae_plan* p, const hlvae_ws* ws, hlvae_stream s);

/* row A -- HL_VAE/utils.py:88-143 batch_normalization.
 * stats: masked column sums (fp64 atomics).  For data-parallel runs all-reduce ws->sums between the two calls. */
int hlvae_normalize_stats(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask,
                          int B, hlvae_stream s);
int hlvae_normalize_pack(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask,
                         int B, hlvae_stream s);

/* The input stage of the NEXT batch, into another set of input-stage buffers (ws_next: a copy of the workspace whose xn,
 * xnT, xt, m8, sums, norm point at the second set).  It depends on the data only, so it can run beside the backward pass of
 * the current batch: deferred like hlvae_step_metrics -- queued on a library-owned side stream by the next hlvae_backward* /
 * hlvae_join on this plan (all pointers must stay valid until then).  Not for the convolutional model (its input stage
 * runs conv1 / conv2, i.e. reads the weights). */
int  hlvae_feed_prefetch(const hlvae_plan* p, const hlvae_ws* ws_next, const float* values, const uint8_t* mask8,
                         const int32_t* rows, int B, hlvae_stream s);

/* rows B + C -- HLVAE.encode MLP branch (HLVAE.py:311-324) + sample_latent (:351-362).
 * noise: eps != NULL -> that [B][L] fp32 tensor;  eps == NULL && sample != 0 -> Philox4x32-10 normals generated in
 * the kernel from ws->rng (+ rng_host_offset);  sample == 0 -> z = mu (get_test_samples, HLVAE.py:472).
 * The noise used is kept in ws->eps for hlvae_backward.  Also writes the per-block partials of KL(q || N(0,I)). */
int hlvae_encoder_fwd(const hlvae_plan* p, const hlvae_ws* ws, const float* eps, int sample, uint64_t rng_host_offset,
                      int B, hlvae_stream s);

/* rows D-J -- HLVAE.decode (HLVAE.py:326-349): decoder trunk, y_layer, theta_estimation (:416-453),
 * loglik_{real,pos,count,cat,ordinal} (HL_VAE/loglik.py) and the scatter/ELBO row sums (HLVAE.py:377-414).
 * g_logpx: upstream gradient of log_p_x, per element [B][D] fp32, or NULL -> the scalar g_scale.
 * Writes ws->dy = g * d log_p_x / d Y (zero where unobserved: stop-gradient of HLVAE.py:435-452) and
 * accumulates the head-parameter gradients into ws->G when want_grad != 0.
 * want_grad = 2 (training step): as 1, and the scalar reductions (ws->nll, ws->scal, RNG offset advance) are DEFERRED like
 * hlvae_step_metrics: queued on a library-owned side stream by the next hlvae_backward* / hlvae_join on this plan, so
 * that the backward pass forks once for all its side work.  ws must stay valid until then.
 * want_params = 1 additionally fills ws->pfull and ws->xhat (p_params / row M); 2 = ws->xhat only (training metrics).
 * trunk != 0 recomputes the decoder trunk U = relu(z Wd^T + bd) from ws->zb first (decode(z) with a caller-set z);
 * hlvae_encoder_fwd already leaves U in the workspace (it is fused with the reparameterisation). */
int hlvae_decoder_fwd(const hlvae_plan* p, const hlvae_ws* ws, const float* g_logpx, float g_scale,
                      int want_grad, int want_params, int trunk, int B, hlvae_stream s);

/* row M -- per-step reconstruction metrics (training.py:84-101 -> read_functions.py:342-412, true_miss_mask = 1):
 * err [3][D] fp32 = per-variable error over observed / missing / all rows (0/1 mismatch for cat, |dx|/K for ordinal,
 * range-normalised RMSE otherwise) from ws->xhat (decoder_fwd with want_params != 0), ws->xt and ws->m8. */
int hlvae_step_metrics(const hlvae_plan* p, const hlvae_ws* ws, int B, float* err, hlvae_stream s);
/* hlvae_step_metrics only records its dependency on s; the kernels are queued on a library-owned side stream by the next
 * hlvae_backward* / hlvae_join on this plan (ws and err must stay valid until then), and `err` is ordered before later
 * work on a stream only after that stream called hlvae_backward* or hlvae_join. */
int hlvae_join(const hlvae_plan* p, hlvae_stream s);
/* on != 0: hlvae_backward / hlvae_backward_adam return WITHOUT joining the deferred side chain (metrics, next batch's input stage);
 * the host queues what does not depend on it (the GP prior's state update) and calls hlvae_join itself.  Un-joined work is
 * joined by the plan's next forward call at the latest.
 * on == 2: as 1, and the deferred work is queued on the CALLER's stream behind the backward pass's last launch of its own instead
 * of on the library's side stream (a host whose own side streams own the hardware queues: the GP prior with a deferred update). */
int hlvae_set_defer_join(const hlvae_plan* p, int on);

/* rescale ws->dy by a per-element upstream gradient after the fact (autograd path) */
int hlvae_scale_dy(const hlvae_plan* p, const hlvae_ws* ws, const float* g_logpx, int B, hlvae_stream s);

/* backward of rows B-D: all dense-layer gradients into ws->G (reads the noise from ws->eps).
 * g_mu, g_lv: upstream gradients of mu / log_var from the KL term ([B][L] fp32, may be NULL);
 * kl_std_weight != 0 adds the gradient of kl_std_weight * KL(q(z|x) || N(0,I)) in the same kernel
 * (closed form; NOT in the reference).
 * skip_wy = 1 (data-parallel host): d Wy is not computed (hlvae_backward_wy did) and the call returns WITHOUT joining the deferred
 * side work (metrics, next batch's input stage): the host's collectives and optimiser calls that follow do not depend on it;
 * call hlvae_join at the end of the step. */
int hlvae_backward(const hlvae_plan* p, const hlvae_ws* ws, const float* g_mu, const float* g_lv,
                   float kl_std_weight, int skip_wy, int B, hlvae_stream s);
/* only d Wy = dY^T U, on the given stream.  Data-parallel hosts call this first, start the all-reduce of that arena
 * slice, then hlvae_backward(..., skip_wy = 1): the collective overlaps the rest of the backward pass. */
int hlvae_backward_wy(const hlvae_plan* p, const hlvae_ws* ws, int B, hlvae_stream s);
/* zero the atomically accumulated gradient region; call before hlvae_decoder_fwd(want_grad=1) */
int hlvae_zero_grad(const hlvae_plan* p, const hlvae_ws* ws, hlvae_stream s);

/* torch.optim.Adam(lr) step on the flat arena (HLVAE_main.py:277-278) fused with the bf16 shadow refresh.
 * step_count: device int64[2] advanced by the kernels (graph-capture safe). grad_scale multiplies G first.
 * The small-parameter gradient region [0, atomic_region) is zeroed after it is consumed. */
int hlvae_adam_step(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count,
                    float lr, float beta1, float beta2, float eps, float grad_scale, hlvae_stream s);

/* hlvae_backward + hlvae_adam_step as ONE call (single-process training): identical result, but the Adam update of
 * y_layer's weight (the largest slice of the HBM-bound optimiser) is queued on the side stream as soon as its gradient and
 * its last reader of the step are done, so it runs under the latency-bound remainder of the backward pass.
 * MLP model with one hidden layer per side, widths that are multiples of 4 and enough output tiles that the batch axis of the
 * weight gradients is not sliced: the optimiser step is applied in the EPILOGUE of the weight-gradient GEMMs (csrc/dense.hip:
 * k_gemm_adam) -- the gradients of the dense matrices never reach ws->G, and the step is three launches ({W1, Wd, Wmu, Wlv} on
 * the caller's stream; y_layer, then the small region on a side stream) that share one completion ticket.  Same arithmetic
 * as hlvae_backward + hlvae_adam_step (tests/test_gpu_configs.py: test_fused_optimiser_epilogue_matches_separate_launches). */
int hlvae_backward_adam(const hlvae_plan* p, const hlvae_ws* ws, const float* g_mu, const float* g_lv, float kl_std_weight,
                        int B, float* m1, float* m2, int64_t* step_count, float lr, float beta1, float beta2, float eps,
                        float grad_scale, hlvae_stream s);
/* 1 when hlvae_backward_adam takes that fused form for this plan and batch size (ws->wys_next may then be set), else 0 */
int hlvae_backward_adam_fused(const hlvae_plan* p, int B);

/* ---- sharded optimiser step (data parallel: hl-vae_amd/parallel.py; the reference has no distributed code, the semantics
 * to keep are ONE optimiser step on the sum of the ranks' gradients, training.py:121-128) ------------------------------------
 * The dense part of the arena [atomic_region, arena_size) is cut into equal contiguous slices, one per rank.  After the
 * reduce-scatter of the gradient arena a rank holds the SUMMED gradients of its slice in a compact buffer grad_shard[n]
 * (element i of the arena at grad_shard[i - lo]); hlvae_adam_shard updates master / m / v of [lo, lo + n) only (m1, m2 are
 * indexed like the arena) and writes the bf16 copy of the updated values to pb16_shard[i - lo].  The caller all-gathers the
 * bf16 slices; hlvae_shadows_from_bf16 then rebuilds the padded row-major + transposed shadows of the matrices `which`
 * (bit 0 y_layer, 1 encoder Linear, 2 d_layers, 3 mean_layer, 4 log_var_layer) from that flat bf16 copy, where arena
 * element i lives at pb16[i - base].  hlvae_adam_small is the replicated part: Adam on [0, atomic_region) (head parameters,
 * biases, convolution weights: all-reduce their gradients first), zeroes the consumed gradients and commits the step number
 * -- call it AFTER every hlvae_adam_shard of the step, on the same stream. */
int hlvae_adam_shard(const hlvae_plan* p, const hlvae_ws* ws, const float* grad_shard, float* m1, float* m2, uint16_t* pb16_shard,
                     const int64_t* step_count, int64_t lo, int64_t n, float lr, float beta1, float beta2, float eps,
                     float grad_scale, hlvae_stream s);
int hlvae_adam_small(const hlvae_plan* p, const hlvae_ws* ws, float* m1, float* m2, int64_t* step_count, float lr, float beta1,
                     float beta2, float eps, float grad_scale, hlvae_stream s);
int hlvae_shadows_from_bf16(const hlvae_plan* p, const hlvae_ws* ws, const uint16_t* pb16, int64_t base, unsigned which,
                            hlvae_stream s);

/* ---- GP-prior KL (row K): reference elbo_functions.py:196-285 with the kernels of GP_model.py:27-116, fp64 ----------
 * An additive kernel = sum over terms of  scale_t[l] * prod_f factor_f(x[dim], x'[dim]);  factors: categorical equality,
 * binary AND, RBF with its own lengthscale per latent dimension (at most 2 RBF factors per term).  Hyper-parameters are
 * one array of RAW values raw [n_slots][L]; the kernels read the transformed planes hyp [3][n_slots][L] that
 * hlvae_gp_transform writes once per step: pos = exp(-16 + softplus(raw + 16)) (GP_model.py:57,85),
 * dpos = sigmoid(raw + 16) (d pos / d raw = pos * dpos), il2 = 1 / pos^2.  Gradients come back w.r.t. the RAW values. */
#define HLVAE_GP_MAX_TERMS 8
#define HLVAE_GP_MAX_FACTORS 4
enum { HLVAE_GP_CAT = 0, HLVAE_GP_BIN = 1, HLVAE_GP_RBF = 2 };
typedef struct {
    int32_t n_terms;
    int32_t scale_slot[HLVAE_GP_MAX_TERMS];                        /* row of raw/hyp with the scale               */
    int32_t n_factors[HLVAE_GP_MAX_TERMS];
    int32_t kind[HLVAE_GP_MAX_TERMS][HLVAE_GP_MAX_FACTORS];
    int32_t dim[HLVAE_GP_MAX_TERMS][HLVAE_GP_MAX_FACTORS];         /* covariate column                            */
    int32_t ls_slot[HLVAE_GP_MAX_TERMS][HLVAE_GP_MAX_FACTORS];     /* row of raw/hyp with the lengthscale, or -1  */
} hlvae_gp_kernel;

int hlvae_gp_transform(const double* raw, int n_slots, int L, double* hyp, hlvae_stream s);
/* out [L][n1][n2] = K(x1_i, x2_j) (+ jitter on i == j).  x1 / x2: [n][Q] shared by all latents, or [L][n][Q] when the
 * per_latent flag is set (inducing points zt_list).  Replaces covar_module(x1, x2).evaluate() (elbo_functions.py:222-223). */
int hlvae_gp_kernel_matrix(const hlvae_gp_kernel* k, const double* hyp, int n_slots, int L, int Q, const double* x1, int n1,
                           int per_latent1, const double* x2, int n2, int per_latent2, double jitter, double* out,
                           hlvae_stream s);
/* batched SPD inverse + log-determinant, N <= 128, one workgroup per matrix, Gauss-Jordan with the matrix resident in
 * registers (replaces torch.cholesky + cholesky_solve(eye), elbo_functions.py:225-228, training.py:131-135).
 * *fail (device int, may be NULL) is set to 1 if a pivot is not positive. */
int hlvae_gp_chol_inv(const double* A, int n, int N, double* inv, double* logdet, int* fail, hlvae_stream s);
/* the per-subject loop of elbo_functions.py:243-266 as one workgroup per (subject, latent).  idx [S][T]: batch row of the
 * t-th observation of subject s, -1 = padding (T <= 32).  resid [L][B] = K0xz iK0zz m - mu^T.  Outputs: iB, K0s [S][L][T][T];
 * V [L][B][M] = iB_s K0xz_s (row-indexed by batch row); v [L][B] = iB_s resid_s; part [S][L][4] = {A, B, C, sum(iB*K0)}
 * contributions; g_mu, g_lv [B][L] fp32 = d(KL bound)/d(mu, log_var) with c = P / P_batch.
 * iKm [L][M] = iK0zz m (or NULL): the residual is then computed inside from the staged rows of K0xz and mu [B][L] (resid unused).
 * u_acc, p1_acc [L][M] (both or NULL; need mu): += this batch's K0xz^T v and V^T mu (elbo_functions.py:262-266), per-subject
 * fp64 atomics -- the caller zeroes them. */
int hlvae_gp_subject_fwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* hyp, int n_slots, int L, int Q,
                         const double* x, const double* noise, const int32_t* idx, int S, int T, const double* Kxz, int B,
                         int M, const double* resid, const float* lv, double c, double* iB, double* K0s, double* V,
                         double* v, double* part, float* g_mu, float* g_lv, const double* iKm, const float* mu, double* u_acc,
                         double* p1_acc, hlvae_stream s);
/* gradients of the bound w.r.t. B_st and K0_st chained into the hyper-parameters (accumulates into gprm [n_slots][L]).
 * Y [L][B][M] = V (iK0zz - iK0zz H iK0zz). */
int hlvae_gp_subject_bwd(const hlvae_gp_kernel* k0, const hlvae_gp_kernel* k1, const double* hyp, int n_slots, int L, int Q,
                         const double* x, const int32_t* idx, int S, int T, int B, int M, const double* iB, const double* K0s,
                         const double* V, const double* v, const double* Y, const float* lv, double c, double* gprm,
                         hlvae_stream s);
/* chain rule from G [L][n1][n2] = dL/dK(x1_i, x2_j) into gprm [n_slots][L] and gx2 [L][n2][Q] (points of the second
 * argument, always per latent).  both_args != 0: x1 == x2 (K0zz) and G must arrive symmetrised (G + G^T): the points
 * sit in both argument positions.
 * v, w != NULL (v [L][n1], w [L][n2]): G holds Y and the gradient is formed on the fly, G_ij := c (v_i w_j - Y_ij)
 * (the gradient of the bound w.r.t. K0xz, what hlvae_gp_gkxz writes out as a matrix). */
int hlvae_gp_param_grad(const hlvae_gp_kernel* k, const double* hyp, int n_slots, int L, int Q, const double* x1, int n1,
                        int per_latent1, const double* x2, int n2, int both_args, const double* G, double* gprm, double* gx2,
                        const double* v, const double* w, double c, hlvae_stream s);
/* hlvae_gp_chol_inv that also writes -logdet of the first n_neg matrices to logdet_neg[0 .. n_neg): the end-of-step inversion of
 * [iH_new | K0zz] yields [H_new | iK] and needs log det H_new = -log det iH_new (training.py:131-135) */
int hlvae_gp_spd_inv2(const double* A, int n, int N, double* inv, double* logdet, int n_neg, double* logdet_neg, int* fail,
                      hlvae_stream s);
/* batched fp64 product on the matrix cores: C[l] (M x N, ldc) = alpha op(A[l]) B[l] + beta D[l]; op(A) = A ([M][K] rows, lda)
 * or, transA != 0, A^T with A stored [K][M]; B stored [K][N] (ldb); strides in elements between batch items; D may be NULL.
 * The rectangular products of the bound: W = Kxz^T V (sum_s Ks^T iB Ks, elbo_functions.py:256-261) and Y = V (iK - iK H iK). */
int hlvae_gp_gemm(const double* A, int lda, int64_t strideA, int transA, const double* B, int ldb, int64_t strideB, const double* D,
                  int ldd, int64_t strideD, double* C, int ldc, int64_t strideC, int M, int N, int K, int batch, double alpha,
                  double beta, hlvae_stream s);
/* C[l] += alpha op(A[l]) B[l]: as hlvae_gp_gemm without D, adding into C as the caller left it (cleared beforehand: no memset in
 * front of the launch that heads the GP step's critical chain) */
int hlvae_gp_gemm_acc(const double* A, int lda, int64_t strideA, int transA, const double* B, int ldb, int64_t strideB, double* C,
                      int ldc, int64_t strideC, int M, int N, int K, int batch, double alpha, hlvae_stream s);
/* out[l] = alpha A[l] x[l] + beta y[l]  (A [batch][N][N] row-major; x, y, out [batch][N]; y may be NULL or alias out) */
int hlvae_gp_bmv(const double* A, const double* x, const double* y, double* out, int N, int batch, double alpha, double beta,
                 hlvae_stream s);
/* resid[l][b] = sum_m Kxz[l][b][m] w[l][m] - mu[b][l]  (A_part of elbo_functions.py:230; mu fp32 [B][L] = the VAE's encoder means) */
int hlvae_gp_resid(const double* Kxz, const double* w, const float* mu, int L, int B, int M, double* out, hlvae_stream s);
/* hlvae_gp_gemv_t with an fp32 vector operand (V^T mu straight from the VAE's fp32 means) */
int hlvae_gp_gemv_t_f32(const double* A, const float* x, long x_stride_l, long x_stride_b, double* out, int L, int B, int M,
                        hlvae_stream s);
/* natural-gradient terms (elbo_functions.py:279-283) and the right-hand side of the (m, H) update (training.py:130-137):
 *   grad_m = -(iK P1) + Bm m;  grad_H = (Bm - iH) / 2;  tmp = iH m - lr (grad_m - 2 grad_H m)
 * with Bm = iK W iK + iK [batch][N][N]; m, P1, grad_m, tmp [batch][N].  hlvae_gp_natgrad_apply then updates the precision in
 * place, iH <- iH + lr (grad_H + grad_H^T) (training.py:131-133): the input of the end-of-step inversion, after which
 * m_new = H_new tmp (hlvae_gp_bmv). */
int hlvae_gp_natgrad(const double* Bm, const double* iK, const double* iH, const double* m, const double* P1, double lr, int N,
                     int batch, double* grad_m, double* grad_H, double* tmp, hlvae_stream s);
int hlvae_gp_natgrad_apply(const double* grad_H, double* iH, double lr, int N, int batch, hlvae_stream s);
/* all scalar reductions of the bound (elbo_functions.py:268-285) into *out (device double):
 *   c/2 [sum(part) - sum(N1 o W) - sum(log_var)] + 1/2 [sum(iK o H) + m.iKm - L M + sum ldK - sum ldH] - L N/2
 * W = sum_s Ks^T iB Ks, N1 = iK - iK H iK (passed as Qm), all [L][M][M]; m, iKm [L][M]; ldK, ldH [L]; lv fp32 [B][L].
 * rep scales the second bracket and the constant: 1 in a single process; 1 / world under data parallelism, where every
 * rank passes its local part / W / lv and the bound of the global batch is the SUM of the ranks' results. */
int hlvae_gp_bound(const double* part, int S, const double* W, const double* iK, const double* Qm, const double* H,
                   const double* m, const double* iKm, const double* ldK, const double* ldH, const float* lv, int B, int L,
                   int M, double c, double n_total, double rep, double* out, hlvae_stream s);
/* batched dense N x N fp64 products  C[l] = alpha A[l] B[l] + beta D[l]  (row-major [batch][N][N], N <= 128, D may be NULL or
 * alias C): the M x M algebra of the bound and of the natural gradient (elbo_functions.py:268-283) on the fp64 matrix cores. */
int hlvae_gp_bmm(const double* A, const double* B, const double* D, double* C, int N, int batch, double alpha, double beta,
                 hlvae_stream s);
/* out[l][m] = sum_b A[l][b][m] x[l][b]  (A: [L][B][M] dense, x element (l, b) at x[l * x_stride_l + b * x_stride_b], M <= 128):
 * the matrix^T-vector products Kxz^T v and V^T mu of the bound as one streaming pass per latent */
int hlvae_gp_gemv_t(const double* A, const double* x, long x_stride_l, long x_stride_b, double* out, int L, int B, int M,
                    hlvae_stream s);
/* G[l][b][m] = c (v[l][b] w[l][m] - Y[l][b][m])   (Y, G: [L][B][M]; v: [L][B]; w: [L][M]): gradient w.r.t. K0xz in one pass */
int hlvae_gp_gkxz(const double* Y, const double* v, const double* w, double c, int L, int B, int M, double* G, hlvae_stream s);
/* out[l] = c (u m^T + m u^T - W + X + X^T) + H + m m^T  per latent (u, m: [batch][N]; W, X, H, out: [batch][N][N]): the
 * symmetrised gradient term of K0zz in one pass */
int hlvae_gp_rsym(const double* u, const double* m, const double* W, const double* X, const double* H, double c, int N, int batch,
                  double* out, hlvae_stream s);
/* The N x N algebra of a GP-prior step behind W (reference elbo_functions.py:279-283: natural-gradient terms; the gradient of the
 * bound w.r.t. K0zz) as ONE launch, two independent chains per latent (N % 4 == 0, N <= 128; all matrices [batch][N][N], vectors
 * [batch][N]; the intermediates T1, Bm, HiKW, Rs, T1b are distinct caller buffers):
 *   T1 = iK W;  Bm = T1 iK + iK;  grad_m = Bm m - iK P1;  grad_H = (Bm - iH) / 2;  tmp = iH m - lr (grad_m - 2 grad_H m)
 *   HiKW = HiK W;  Rs = c (u m^T + m u^T - W + HiKW + HiKW^T) + H + m m^T;  T1b = iK Rs;  G = g_alpha (T1b iK) + g_beta iK
 * replaces hlvae_gp_bmm x 5, hlvae_gp_natgrad and hlvae_gp_rsym on that path (same arithmetic, fp64 matrix cores). */
int hlvae_gp_chain(const double* iK, const double* W, const double* HiK, const double* H, const double* iH, const double* m,
                   const double* P1, const double* u, double lr, double c, double g_alpha, double g_beta, int N, int batch,
                   double* T1, double* Bm, double* grad_m, double* grad_H, double* tmp, double* HiKW, double* Rs, double* T1b,
                   double* G, hlvae_stream s);
/* The same outputs (grad_m, grad_H, tmp, G; Rs as a caller buffer) by 32-row blocks in two launches of grid (batch, ceil(N / 32)):
 * a row block of T1, Bm, H iK W and W (H iK)^T needs no other block; the second launch multiplies the completed Rs from both
 * sides.  No workgroup waits for another. */
int hlvae_gp_chain_rb(const double* iK, const double* W, const double* HiK, const double* H, const double* iH, const double* m,
                      const double* P1, const double* u, double lr, double c, double g_alpha, double g_beta, int N, int batch,
                      double* grad_m, double* grad_H, double* tmp, double* Rs, double* G, hlvae_stream s);
/* torch.optim.Adam step (HLVAE_main.py:277-278) on a flat fp64 arena (hyper-parameters + inducing points, n <= ~1e5);
 * step: device int64[2] = {completed steps, 0}, advanced by the kernel; the consumed gradients are zeroed. */
int hlvae_gp_adam(double* p, double* g, double* m1, double* m2, int n, int64_t* step, double lr, double b1, double b2,
                  double eps, hlvae_stream s);
/* hlvae_gp_adam on the arena [raw hyper-parameters n_slots x L | inducing points], hlvae_gp_transform of the updated
 * hyper-parameters into hyp, and hlvae_gp_natgrad_apply (iH += ng_lr (grad_H + grad_H^T), [batch][N][N]) as ONE launch: the head of
 * the prior's state update (reference HLVAE_main.py:277-278, training.py:130-133). */
int hlvae_gp_state_head(double* p, double* g, double* m1, double* m2, int n, int64_t* step, double lr, double b1, double b2,
                        double eps, int n_slots, int L, double* hyp, const double* grad_H, double* iH, double ng_lr, int N, int batch,
                        hlvae_stream s);

/* Per-kernel HIP-event timing (bench.py's roofline leg): while enabled every kernel launch of this library is
 * bracketed by hipEventRecord on its stream, and the work the library normally forks onto its side streams is queued on
 * the caller's stream instead, so that every kernel is timed ALONE.  hlvae_prof_report synchronises the device and writes one line
 * "<kernel-label> <launches> <total_ms>" per label into buf.  Do not enable during hipGraph capture. */
void hlvae_prof_enable(int on);
int  hlvae_prof_report(char* buf, int buflen);

/* forget the side work that hlvae_step_metrics / hlvae_decoder_fwd(want_grad = 2) / hlvae_feed_prefetch deferred and that no
 * hlvae_backward* / hlvae_join has queued yet (host bookkeeping only; used after a HIP-graph capture that failed mid-step) */
int  hlvae_reset_pending(const hlvae_plan* p);

/* In-graph kernel stamps (bench.py "roofline.in_step"; no counterpart in the reference): buf = device uint64 [hlvae_stamp_words()],
 * or NULL to switch the feature off (the default).  Set BEFORE the step is launched or captured: the pointer is a kernel argument.
 * Kernel k (0 enc1 split-K, 1 fused middle fwd, 2 head kernel, 3 dU split-K, 4 fused middle bwd, 5 grouped weight gradient + Adam,
 * 6 y_layer weight gradient + Adam) owns 32 sub-slots of 8 words at buf[256 k]: word 0 = earliest workgroup start, word 1 = latest
 * workgroup end, in 10 ns ticks of s_memrealtime (a workgroup uses sub-slot block-id mod 32; fold them on the host).  Arm: word 0
 * of every sub-slot = ~0, word 1 = 0.  Disarm: word 0 of sub-slot 0 = 0 (nothing is recorded). */
int  hlvae_stamp_slots(void);
int  hlvae_stamp_words(void);
void hlvae_stamp_buffer(uint64_t* buf);

/* generic bf16 NT GEMM  C[M][N] (fp32, ldc) = A[M][K] * B[N][K]^T, exposed for unit tests */
int hlvae_gemm_nt_f32(const uint16_t* A, int lda, const uint16_t* B, int ldb, float* C, int ldc,
                      int M, int N, int K, hlvae_stream s);

#ifdef __cplusplus
}
#endif
#endif /* HLVAE_HIP_H */
