// Convolutional front / back end of HL-VAE (reference HLVAE.py:139-152 conv1/conv2,
// :253-259 deconv_layer, :293-308 encode, :338-341 decode): what the shipped configuration selects
// (config/hlvae_config_file.txt:51).  Geometry is fixed by the reference: 1296 variables = one 36 x 36 image,
//     encoder  image -> conv 3x3 (1->16) + ReLU + maxpool 2 -> conv 3x3 (16->32) + ReLU + maxpool 2 -> 2592 features
//     decoder  2592 -> [32, 9, 9] -> ConvTranspose 4x4 s2 p1 (32->16) + ReLU -> ConvTranspose 4x4 s2 p1 (16->y_dim=5)
//              -> y[b, pixel, 0..4] for the per-variable heads.
//
// One workgroup (4 waves) owns one image; every layer is an implicit GEMM on the MFMA units whose A operand is
// gathered from an activation tile held in LDS in channel-last order (NHWC with a zero halo): the 8 consecutive k of
// one lane's fragment are 8 consecutive channels of one tap = ONE aligned 16-byte LDS read.  Weights are pre-packed
// (k_conv_pack, once per optimiser step) into K-contiguous bf16 rows so that a B fragment is one 16-byte global load.
// Pooling costs nothing: the GEMM row order puts the 4 pixels of a 2 x 2 window into the 4 accumulator rows of one lane.
// Weight gradients are MFMA products over the pixel axis (fragments gathered with stride = channel count) whose
// accumulators stay in registers while a workgroup walks its images; one atomic flush per workgroup.
#include "common.h"

#define CV_W 36
#define CV_D (CV_W * CV_W)
#define CV_C1 16
#define CV_C2 32
#define CV_H1 18
#define CV_H2 9
#define CV_FEAT (CV_C2 * CV_H2 * CV_H2)
#define CV_PART_ROWS 512              // workgroups of the backward kernels = rows of the partial-gradient buffer

// packed bf16 weights inside ws->cpack (elements)
#define CP_C2F 0                          // conv2 forward        [32 co][160]   k = tap * 16 + ci        (144 used)
#define CP_C2D (CP_C2F + 32 * 160)        // conv2 data gradient  [16 ci][288]   k = tap' * 32 + co
#define CP_T1F (CP_C2D + 16 * 288)        // deconv1 forward      [4 parity][16 co][128]  k = (a*2+b) * 32 + ci
#define CP_T1D (CP_T1F + 4 * 16 * 128)    // deconv1 data grad    [32 ci][256]   k = (ky*4+kx) * 16 + co
#define CP_T2F (CP_T1D + 32 * 256)        // deconv2 forward      [4 parity][16 co (5 used)][64]  k = (a*2+b) * 16 + ci
#define CP_T2D (CP_T2F + 4 * 16 * 64)     // deconv2 data grad    [16 ci][128]   k = (ky*4+kx) * 8 + co (5 used)
#define CP_TOTAL (CP_T2D + 16 * 128)

// ConvTranspose2d(k = 4, s = 2, p = 1): output row o = 2 i - 1 + k.  For output parity p the two contributing taps are
//   p = 0: (input offset 0, k = 1), (input offset -1, k = 3);   p = 1: (offset +1, k = 0), (offset 0, k = 2)
__host__ __device__ __forceinline__ int ct_off(int p, int a) { return p == 0 ? (a == 0 ? 0 : -1) : (a == 0 ? 1 : 0); }
__host__ __device__ __forceinline__ int ct_k(int p, int a) { return p == 0 ? (a == 0 ? 1 : 3) : (a == 0 ? 0 : 2); }

__device__ __forceinline__ bf16x8_t ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8_t*>(p); }
__device__ __forceinline__ bf16x8_t gather8(const bf16_t* base, const int (&off)[8]) {
    bf16x8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = __builtin_bit_cast(__bf16, base[off[e]]);
    return v;
}
__device__ __forceinline__ f32x4_t mfma16(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
#define F4Z f32x4_t{0.f, 0.f, 0.f, 0.f}

__global__ __launch_bounds__(256) void k_conv_pack(const float* __restrict__ P, hlvae_dims d, bf16_t* __restrict__ cp) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < CP_TOTAL; e += gridDim.x * 256) {
        float v = 0.f;
        if (e < CP_C2D) {                                  // conv2.weight [co][ci][ky][kx]
            const int co = e / 160, k = e % 160;
            if (k < 144) v = P[d.o_c2w + (co * 16 + (k & 15)) * 9 + (k >> 4)];
        } else if (e < CP_T1F) {                           // reads dz2 at (y + ty - 1, x + tx - 1): ky = 2 - ty
            const int r = e - CP_C2D, ci = r / 288, k = r % 288, tp = k >> 5, co = k & 31;
            v = P[d.o_c2w + (co * 16 + ci) * 9 + (2 - tp / 3) * 3 + (2 - tp % 3)];
        } else if (e < CP_T1D) {                           // deconv_layer.0.weight [ci 32][co 16][ky][kx]
            const int r = e - CP_T1F, par = r / (16 * 128), co = (r / 128) & 15, k = r & 127, t = k >> 5, ci = k & 31;
            v = P[d.o_t1w + (ci * 16 + co) * 16 + ct_k(par >> 1, t >> 1) * 4 + ct_k(par & 1, t & 1)];
        } else if (e < CP_T2F) {
            const int r = e - CP_T1D, ci = r >> 8, k = r & 255;
            v = P[d.o_t1w + (ci * 16 + (k & 15)) * 16 + (k >> 4)];
        } else if (e < CP_T2D) {                           // deconv_layer.2.weight [ci 16][co 5][ky][kx]
            const int r = e - CP_T2F, par = r / (16 * 64), co = (r / 64) & 15, k = r & 63, t = k >> 4, ci = k & 15;
            if (co < 5) v = P[d.o_t2w + (ci * 5 + co) * 16 + ct_k(par >> 1, t >> 1) * 4 + ct_k(par & 1, t & 1)];
        } else {
            const int r = e - CP_T2D, ci = r >> 7, k = r & 127, co = k & 7;
            if (co < 5) v = P[d.o_t2w + (ci * 5 + co) * 16 + (k >> 3)];
        }
        cp[e] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------------------
// encoder front end, shared by the forward and the backward kernel
// ------------------------------------------------------------------------------------------------------------
#define IMG_LD 38                    // 36 + halo
#define A1_LD 20                     // 18 + halo

// conv1 3x3 (1 -> 16) + bias + ReLU + maxpool 2 from the fp32 image tile (halo = 0) into the NHWC bf16 tile a1;
// am1 (optional): which of the 4 window positions won (first maximum, scan order), 4 = none (all <= 0)
__device__ __forceinline__ void conv1_pool(const float* img, const float* w1s, bf16_t* a1, uint8_t* am1, int tid) {
    const int co = tid & 15;                                   // 256 % 16 == 0: a thread keeps its output channel
    float w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = w1s[co * 9 + t];
    const float bias = w1s[144 + co];
    // 324 windows / 16 per pass = 21 passes (the last one partial); unrolled by 3 so that the LDS reads of the next windows
    // are in flight under the 36 FMAs of the current one (two waves per SIMD do not hide them: 13.5 k clocks before)
    static_assert(IMG_LD % 2 == 0, "8-byte patch reads");
#pragma unroll 3
    for (int pp = tid >> 4; pp < CV_H1 * CV_H1; pp += 16) {
        const int py = pp / CV_H1, px = pp % CV_H1;
        float pch[4][4];                                       // the 4 x 4 input patch of one pooling window (16-lane broadcast)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 lo = *reinterpret_cast<const float2*>(img + (2 * py + i) * IMG_LD + 2 * px);
            const float2 hi = *reinterpret_cast<const float2*>(img + (2 * py + i) * IMG_LD + 2 * px + 2);
            pch[i][0] = lo.x; pch[i][1] = lo.y; pch[i][2] = hi.x; pch[i][3] = hi.y;
        }
        float best = 0.f;
        int sel = 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float acc = bias;
#pragma unroll
            for (int t = 0; t < 9; ++t) acc += pch[(r >> 1) + t / 3][(r & 1) + t % 3] * w[t];
            if (acc > best) { best = acc; sel = r; }
        }
        a1[((py + 1) * A1_LD + px + 1) * CV_C1 + co] = f2bf(best);
        if (am1 != nullptr) am1[pp * CV_C1 + co] = (uint8_t)sel;
    }
}

// input stage of the convolutional encoder (HLVAE.py:293-304 + HL_VAE/utils.py:88-143 with types_info['conv']):
// one number per variable -- real: x / 255; pos: normalised log1p; count: log x; cat / ordinal: learned
// representation sum_k x_k w[d][k] + bias[d] (Representation_One_Hot, HLVAE.py:91-102) -- times the mask.
// Also packs the likelihood targets exactly like k_normalize_pack.
__device__ __forceinline__ float conv_input_value(const hlvae_var& var, const double* __restrict__ px, const float* __restrict__ P,
                                                  const double* __restrict__ sums, int n_stat, float& tv) {
    const float x0 = (float)px[0];
    switch (var.kind) {
        case HLVAE_REAL:
            tv = x0;
            return x0 / 255.f;                                                // utils.py:99-102
        case HLVAE_COUNT:
            tv = x0;
            return __logf(x0);                                                // :118
        case HLVAE_POS: {
            double n = 0, s1 = 0, s2 = 0;
            for (int ch = 0; ch < HL_STAT_CHUNKS; ++ch) {
                n += sums[((size_t)ch * 3 + 0) * n_stat + var.sidx];
                s1 += sums[((size_t)ch * 3 + 1) * n_stat + var.sidx];
                s2 += sums[((size_t)ch * 3 + 2) * n_stat + var.sidx];
            }
            const double mu = s1 / n;
            double vv = (s2 - 2.0 * mu * s1 + mu * mu * n) / n;
            vv = fmin(fmax(vv, 1e-6), 1e20);                                  // :128
            const float lg = log1pf(x0);
            tv = lg;
            return (float)((lg - mu) / sqrt(vv + 1e-5));                      // :129
        }
        default: {
            float rep = P[var.rb_off];
            int sum = 0, cls = -1;
            float best = 0.f;
            for (int k = 0; k < var.ncls; ++k) {
                const float xk = (float)px[k];
                rep += xk * P[var.r_off + k];
                sum += (int)xk;
                if (xk > best) { best = xk; cls = k; }
            }
            tv = var.kind == HLVAE_CAT ? (float)cls : (float)(sum - 1);       // as k_normalize_pack (loglik.py:172)
            return rep;
        }
    }
}

// conv2 3x3 (16 -> 32) on the MFMA units.  GEMM row m = 4 * window + (dy * 2 + dx): the four rows a lane holds in its
// accumulator fragment are the four pixels of ONE pooling window.  EPI(window, co, c[4]) is called once per
// (window < 81, output channel) with the four pre-activation values (bias added).
template <typename EPI>
__device__ __forceinline__ void conv2_mfma(const bf16_t* a1, const bf16_t* __restrict__ cp, const float* __restrict__ bias2,
                                           int wave, int lane, EPI epi) {
    const int q = lane >> 4, r16 = lane & 15;
    bf16x8_t bw[2][5];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 5; ++s) bw[nt][s] = ld8(cp + CP_C2F + (nt * 16 + r16) * 160 + s * 32 + q * 8);
    const float b0 = bias2[r16], b1 = bias2[16 + r16];
    for (int mt = wave; mt < 21; mt += 4) {
        const int wa = min(mt * 4 + (r16 >> 2), 80), sub = r16 & 3;
        const int y = 2 * (wa / CV_H2) + (sub >> 1), x = 2 * (wa % CV_H2) + (sub & 1);
        f32x4_t acc0 = F4Z, acc1 = F4Z;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const int tap = min(2 * s + (q >> 1), 8);                   // k >= 144 meets zero weights
            const bf16x8_t a = ld8(a1 + ((y + tap / 3) * A1_LD + x + tap % 3) * CV_C1 + (q & 1) * 8);
            acc0 = mfma16(a, bw[0][s], acc0);
            acc1 = mfma16(a, bw[1][s], acc1);
        }
        const int wi = mt * 4 + q;
        if (wi < 81) {
            float c0[4] = {acc0[0] + b0, acc0[1] + b0, acc0[2] + b0, acc0[3] + b0};
            float c1[4] = {acc1[0] + b1, acc1[1] + b1, acc1[2] + b1, acc1[3] + b1};
            epi(wi, r16, c0);
            epi(wi, 16 + r16, c1);
        }
    }
}

// value of one variable from the compact feed (csrc/feed.hip): cv = raw value | class index | level - 1
__device__ __forceinline__ float conv_input_value_compact(const hlvae_var& var, float cv, const float* __restrict__ P,
                                                          const double* __restrict__ sums, int n_stat) {
    if (var.kind == HLVAE_CAT || var.kind == HLVAE_ORDINAL) {
        float rep = P[var.rb_off];
        const int cls = (int)cv;
        if (var.kind == HLVAE_CAT) {                      // one-hot: exactly one weight (none for an all-zero row, cls = -1)
            if (cls >= 0 && cls < var.ncls) rep += P[var.r_off + cls];
            return rep;
        }
        for (int k = 0; k < var.ncls; ++k)
            if (k <= cls) rep += P[var.r_off + k];
        return rep;
    }
    double x = (double)cv;
    float tv;
    return conv_input_value(var, &x, P, sums, n_stat, tv);
}

__global__ __launch_bounds__(256) void k_conv_enc_fwd(
    const double* __restrict__ data, const double* __restrict__ mask, const float* __restrict__ cvals,
    const uint8_t* __restrict__ cmask, const int32_t* __restrict__ crows, const hlvae_var* __restrict__ vars,
    const float* __restrict__ P, hlvae_dims d, const double* __restrict__ sums, float* __restrict__ norm,
    const bf16_t* __restrict__ cp, bf16_t* __restrict__ xn, float* __restrict__ xt,
    uint8_t* __restrict__ m8, float* __restrict__ img_out, int B, int Bp) {
    __shared__ __attribute__((aligned(16))) float img[IMG_LD * IMG_LD];
    __shared__ __attribute__((aligned(16))) bf16_t a1[A1_LD * A1_LD * CV_C1];
    __shared__ float w1s[160];
    __shared__ __attribute__((aligned(16))) bf16_t feat[CV_FEAT];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < IMG_LD * IMG_LD; i += 256) img[i] = 0.f;
    for (int i = tid; i < A1_LD * A1_LD * CV_C1 / 2; i += 256) reinterpret_cast<uint32_t*>(a1)[i] = 0u;
    if (tid < 160) w1s[tid] = tid < 144 ? P[d.o_c1w + tid] : P[d.o_c1b + tid - 144];
    __syncthreads();
    // fully unrolled: the dependent chains (variable record -> value -> representation weights) of the six variables a
    // thread owns overlap instead of running one after the other
#pragma unroll
    for (int kk = 0; kk < (CV_D + 255) / 256; ++kk) {
        const int dd = tid + 256 * kk;
        if (dd >= CV_D) break;
        const hlvae_var var = vars[dd];
        bool ob;
        float tv, rep;
        if (cvals != nullptr) {                                               // compact device-resident feed
            const size_t o = (size_t)crows[b] * d.D + dd;
            ob = cmask[o] != 0;
            tv = cvals[o];
            rep = conv_input_value_compact(var, tv, P, sums, d.n_stat);
            if (var.kind == HLVAE_POS) tv = log1pf(tv);
        } else {
            ob = mask[(size_t)b * d.D + dd] != 0.0;
            rep = conv_input_value(var, data + (size_t)b * d.X + var.xoff, P, sums, d.n_stat, tv);
        }
        const float val = ob ? rep : 0.f;                                     // HLVAE.py:304 (representation * mask)
        img[(dd / CV_W + 1) * IMG_LD + dd % CV_W + 1] = val;
        img_out[(size_t)b * CV_D + dd] = val;
        xt[(size_t)b * d.D + dd] = tv;
        m8[(size_t)b * d.D + dd] = ob ? 1 : 0;
        if (b == 0 && (var.kind == HLVAE_REAL || var.kind == HLVAE_POS)) {   // statistics the head kernel reads
            float mean = 0.f, var_ = 1.f;                                     // real under conv: none (loglik.py:40-41)
            if (var.kind == HLVAE_POS) {
                double n = 0, s1 = 0, s2 = 0;
                for (int ch = 0; ch < HL_STAT_CHUNKS; ++ch) {
                    n += sums[((size_t)ch * 3 + 0) * d.n_stat + var.sidx];
                    s1 += sums[((size_t)ch * 3 + 1) * d.n_stat + var.sidx];
                    s2 += sums[((size_t)ch * 3 + 2) * d.n_stat + var.sidx];
                }
                const double mu = s1 / n;
                mean = (float)mu;
                var_ = (float)fmin(fmax((s2 - 2.0 * mu * s1 + mu * mu * n) / n, 1e-6), 1e20);
            }
            norm[var.sidx] = mean;
            norm[d.n_stat + var.sidx] = var_;
        }
    }
    __syncthreads();
    conv1_pool(img, w1s, a1, nullptr, tid);
    __syncthreads();
    conv2_mfma(a1, cp, P + d.o_c2b, tid >> 6, tid & 63, [&](int wi, int co, const float (&c)[4]) {
        const float m = fmaxf(fmaxf(c[0], c[1]), fmaxf(c[2], c[3]));
        feat[co * 81 + wi] = f2bf(fmaxf(m, 0.f));                             // HLVAE.py:305-308: view(-1, 32*9*9)
    });
    __syncthreads();
    for (int i = tid; i < CV_FEAT / 2; i += 256)
        reinterpret_cast<uint32_t*>(xn + (size_t)b * d.Xep)[i] = reinterpret_cast<const uint32_t*>(feat)[i];
    // (the transposed copy is made by k_transpose_bf16: from here it would be 2592 two-byte stores per image, each to a
    // line of its own -- 28 MB of HBM write transactions for 2.6 MB of data, PMC)
}

// ------------------------------------------------------------------------------------------------------------
// decoder: ConvTranspose 32 -> 16 (+ ReLU) and 16 -> 5
// ------------------------------------------------------------------------------------------------------------
#define T1_LD 11                     // 9 + halo
__global__ __launch_bounds__(256) void k_convT1_fwd(const bf16_t* __restrict__ yc, int ldy, const bf16_t* __restrict__ cp,
                                                    const float* __restrict__ bias, bf16_t* __restrict__ a2) {
    __shared__ __attribute__((aligned(16))) bf16_t xin[T1_LD * T1_LD * 32];
    __shared__ __attribute__((aligned(16))) bf16_t outs[CV_H1 * CV_H1 * 16];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    // the image's global loads are issued first and land while the tile is being zeroed (16-byte pieces: 8 bf16)
    constexpr int NYC = CV_FEAT / 8;
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    u16x8_t yv[(NYC + 255) / 256];
#pragma unroll
    for (int k = 0; k < (NYC + 255) / 256; ++k) {
        const int i8 = tid + 256 * k;
        yv[k] = i8 < NYC ? reinterpret_cast<const u16x8_t*>(yc + (size_t)b * ldy)[i8] : u16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
    for (int i = tid; i < T1_LD * T1_LD * 32 / 2; i += 256) reinterpret_cast<uint32_t*>(xin)[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (NYC + 255) / 256; ++k) {                             // y.view(-1, 32, 9, 9)  (HLVAE.py:339)
        const int i8 = tid + 256 * k;
        if (i8 < NYC) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * i8 + e8, ci = e / 81, pix = e % 81;
                xin[((pix / 9 + 1) * T1_LD + pix % 9 + 1) * 32 + ci] = yv[k][e8];
            }
        }
    }
    __syncthreads();
    const float bv = bias[r16];
    for (int t = wave; t < 24; t += 4) {
        const int par = t / 6, mt = t % 6, py = par >> 1, px = par & 1;
        const int pos = min(mt * 16 + r16, 80), i = pos / 9, j = pos % 9;
        f32x4_t acc = F4Z;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                                         // one tap (a, b) per k-step, 32 channels
            const bf16x8_t bw = ld8(cp + CP_T1F + (par * 16 + r16) * 128 + s * 32 + q * 8);
            const int iy = i + 1 + ct_off(py, s >> 1), ix = j + 1 + ct_off(px, s & 1);
            acc = mfma16(ld8(xin + (iy * T1_LD + ix) * 32 + q * 8), bw, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mt * 16 + 4 * q + r;
            if (m < 81) {
                const int oy = 2 * (m / 9) + py, ox = 2 * (m % 9) + px;
                outs[(oy * CV_H1 + ox) * 16 + r16] = f2bf(fmaxf(acc[r] + bv, 0.f));
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < CV_H1 * CV_H1 * 16 / 8; i += 256)
        reinterpret_cast<uint4*>(a2 + (size_t)b * CV_H1 * CV_H1 * 16)[i] = reinterpret_cast<const uint4*>(outs)[i];
}

__global__ __launch_bounds__(256) void k_convT2_fwd(const bf16_t* __restrict__ a2, const bf16_t* __restrict__ cp,
                                                    const float* __restrict__ bias, float* __restrict__ yv, int ldv) {
    __shared__ __attribute__((aligned(16))) bf16_t ain[A1_LD * A1_LD * 16];
    __shared__ __attribute__((aligned(16))) float ys[CV_D * 5];
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    constexpr int NA2 = CV_H1 * CV_H1 * 2;                                    // 16 channels = two 16-byte pieces per pixel
    uint4 av[(NA2 + 255) / 256];                                              // in flight while the tile is being zeroed
#pragma unroll
    for (int k = 0; k < (NA2 + 255) / 256; ++k) {
        const int i = tid + 256 * k;
        av[k] = i < NA2 ? reinterpret_cast<const uint4*>(a2 + (size_t)b * CV_H1 * CV_H1 * 16)[i] : make_uint4(0u, 0u, 0u, 0u);
    }
    for (int i = tid; i < A1_LD * A1_LD * 16 / 2; i += 256) reinterpret_cast<uint32_t*>(ain)[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < (NA2 + 255) / 256; ++k) {
        const int i = tid + 256 * k, pix = i >> 1, h = i & 1;
        if (i < NA2) *reinterpret_cast<uint4*>(ain + ((pix / CV_H1 + 1) * A1_LD + pix % CV_H1 + 1) * 16 + h * 8) = av[k];
    }
    __syncthreads();
    const float bv = r16 < 5 ? bias[r16] : 0.f;
    for (int t = wave; t < 84; t += 4) {
        const int par = t / 21, mt = t % 21, py = par >> 1, px = par & 1;
        const int pos = min(mt * 16 + r16, CV_H1 * CV_H1 - 1), i = pos / CV_H1, j = pos % CV_H1;
        f32x4_t acc = F4Z;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                                         // two taps per k-step, 16 channels each
            const bf16x8_t bw = ld8(cp + CP_T2F + (par * 16 + r16) * 64 + s * 32 + q * 8);
            const int tp = 2 * s + (q >> 1);
            const int iy = i + 1 + ct_off(py, tp >> 1), ix = j + 1 + ct_off(px, tp & 1);
            acc = mfma16(ld8(ain + (iy * A1_LD + ix) * 16 + (q & 1) * 8), bw, acc);
        }
        if (r16 < 5) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * q + r;
                if (m < CV_H1 * CV_H1) {
                    const int pix = (2 * (m / CV_H1) + py) * CV_W + 2 * (m % CV_H1) + px;
                    ys[pix * 5 + r16] = acc[r] + bv;                          // y_grouped[b, pixel, c]  (HLVAE.py:341)
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < CV_D * 5 / 4; i += 256)
        reinterpret_cast<float4*>(yv + (size_t)b * ldv)[i] = reinterpret_cast<const float4*>(ys)[i];
}

// ------------------------------------------------------------------------------------------------------------
// backward of the decoder convolutions
// ------------------------------------------------------------------------------------------------------------
#define DY_LD 38                     // 36 + halo, 8 channels (5 used)
// d a2 = ConvTranspose^T(dY) * ReLU'(a2);  d W_t2 += a2 (x) dY;  d b_t2 += sum dY
__global__ __launch_bounds__(256) void k_convT2_bwd(const bf16_t* __restrict__ dy, int lddy, const bf16_t* __restrict__ a2,
                                                    const bf16_t* __restrict__ cp, bf16_t* __restrict__ da2,
                                                    float* __restrict__ part, long part_stride, long part_lo, long o_w,
                                                    long o_b, int B) {
    // gw / gb: this workgroup's row of the partial-gradient buffer (plain stores, every entry written: no atomics, no memset)
    float* __restrict__ gw = part + (size_t)blockIdx.x * part_stride - part_lo + o_w;
    float* __restrict__ gb = part + (size_t)blockIdx.x * part_stride - part_lo + o_b;
    __shared__ __attribute__((aligned(16))) bf16_t dys[DY_LD * DY_LD * 8];
    __shared__ __attribute__((aligned(16))) bf16_t a2s[(CV_H1 * CV_H1 + 1) * 16];        // + one zero row
    __shared__ __attribute__((aligned(16))) bf16_t outs[CV_H1 * CV_H1 * 16];
    __shared__ float bred[5];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    for (int i = tid; i < DY_LD * DY_LD * 8 / 2; i += 256) reinterpret_cast<uint32_t*>(dys)[i] = 0u;
    if (tid < 8) reinterpret_cast<uint32_t*>(a2s + CV_H1 * CV_H1 * 16)[tid] = 0u;
    if (tid < 5) bred[tid] = 0.f;
    f32x4_t wacc[2] = {F4Z, F4Z};                                                 // n-tiles wave, wave + 4 of d W_t2
    float bacc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    bf16x8_t bw[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bw[s] = ld8(cp + CP_T2D + r16 * 128 + s * 32 + q * 8);
    __syncthreads();
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // every global load of the image in flight before the first LDS store (16-byte pieces: 8 bf16)
        constexpr int NDY = CV_D * 5 / 8, NA2 = CV_H1 * CV_H1 * 2;
        static_assert(CV_D * 5 % 8 == 0, "dY row in 16-byte pieces");
        typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
        u16x8_t dv[(NDY + 255) / 256];
        uint4 av[(NA2 + 255) / 256];
#pragma unroll
        for (int k = 0; k < (NDY + 255) / 256; ++k) {
            const int i8 = tid + 256 * k;
            dv[k] = i8 < NDY ? reinterpret_cast<const u16x8_t*>(dy + (size_t)b * lddy)[i8] : u16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < (NA2 + 255) / 256; ++k) {
            const int i = tid + 256 * k;
            av[k] = i < NA2 ? reinterpret_cast<const uint4*>(a2 + (size_t)b * CV_H1 * CV_H1 * 16)[i] : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < (NDY + 255) / 256; ++k) {
            const int i8 = tid + 256 * k;
            if (i8 < NDY) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = 8 * i8 + e8, pix = e / 5, c = e - pix * 5;
                    const bf16_t v = dv[k][e8];
                    dys[((pix / CV_W + 1) * DY_LD + pix % CV_W + 1) * 8 + c] = v;
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) bacc[kk] += (c == kk) ? bf2f(v) : 0.f;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < (NA2 + 255) / 256; ++k) {
            const int i = tid + 256 * k;
            if (i < NA2) reinterpret_cast<uint4*>(a2s)[i] = av[k];
        }
        __syncthreads();
        // data gradient: d in[ci][y][x] = sum_{ky,kx,co} dY[co][2y-1+ky][2x-1+kx] w[ci][co][ky][kx]
        for (int mt = wave; mt < 21; mt += 4) {
            const int pos = min(mt * 16 + r16, CV_H1 * CV_H1 - 1), y = pos / CV_H1, x = pos % CV_H1;
            f32x4_t acc = F4Z;
#pragma unroll
            for (int s = 0; s < 4; ++s) {                                     // four taps per k-step, 8 channels each
                const int tap = 4 * s + q;
                acc = mfma16(ld8(dys + ((2 * y + (tap >> 2)) * DY_LD + 2 * x + (tap & 3)) * 8), bw[s], acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * q + r;
                if (m < CV_H1 * CV_H1) {
                    const bool on = bf2f(a2s[m * 16 + r16]) > 0.f;            // ReLU between the two deconvolutions
                    outs[m * 16 + r16] = f2bf(on ? acc[r] : 0.f);
                }
            }
        }
        // weight gradient: rows ci, columns (tap, co), reduction over the 324 input pixels
        for (int s = 0; s < 11; ++s) {
            int offA[8], offB[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = min(s * 32 + q * 8 + e, CV_H1 * CV_H1);         // 324 = the zero row of a2s
                const int pc = min(p, CV_H1 * CV_H1 - 1);
                offA[e] = p * 16 + r16;
                offB[e] = ((2 * (pc / CV_H1)) * DY_LD + 2 * (pc % CV_H1)) * 8;
            }
            const bf16x8_t a = gather8(a2s, offA);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = (wave + 4 * j) * 16 + r16, tap = n >> 3, co = n & 7;
                const int sh = ((tap >> 2) * DY_LD + (tap & 3)) * 8 + co;
                int ob[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = offB[e] + sh;
                wacc[j] = mfma16(a, gather8(dys, ob), wacc[j]);
            }
        }
        __syncthreads();
        for (int i = tid; i < CV_H1 * CV_H1 * 2; i += 256)
            reinterpret_cast<uint4*>(da2 + (size_t)b * CV_H1 * CV_H1 * 16)[i] = reinterpret_cast<const uint4*>(outs)[i];
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = (wave + 4 * j) * 16 + r16, tap = n >> 3, co = n & 7;
        if (co < 5)
#pragma unroll
            for (int r = 0; r < 4; ++r) gw[((4 * q + r) * 5 + co) * 16 + tap] = wacc[j][r];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float v = wave_sum(bacc[k]);
        if (lane == 0) atomicAdd(&bred[k], v);
    }
    __syncthreads();
    if (tid < 5) gb[tid] = bred[tid];
}

// d yc = ConvTranspose^T(d a2);  d W_t1 += yc (x) d a2;  d b_t1 += sum d a2
__global__ __launch_bounds__(256) void k_convT1_bwd(const bf16_t* __restrict__ da2, const bf16_t* __restrict__ yc, int ldy,
                                                    const bf16_t* __restrict__ cp, bf16_t* __restrict__ dyc,
                                                    float* __restrict__ part,
                                                    long part_stride, long part_lo, long o_w, long o_b, long o_by, int B) {
    float* __restrict__ prow = part + (size_t)blockIdx.x * part_stride - part_lo;
    float* __restrict__ gw = prow + o_w;
    float* __restrict__ gb = prow + o_b;
    float* __restrict__ gby = prow + o_by;
    __shared__ __attribute__((aligned(16))) bf16_t das[A1_LD * A1_LD * 16];
    __shared__ __attribute__((aligned(16))) bf16_t ycs[82 * 32];                  // NHWC, row 81 = zeros
    __shared__ __attribute__((aligned(16))) bf16_t outs[CV_FEAT];
    __shared__ __attribute__((aligned(16))) float wstage[32 * 16 * 16];         // d W_t1 on its way out
    __shared__ float bred[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    for (int i = tid; i < A1_LD * A1_LD * 16 / 2; i += 256) reinterpret_cast<uint32_t*>(das)[i] = 0u;
    if (tid < 16) { reinterpret_cast<uint32_t*>(ycs + 81 * 32)[tid] = 0u; bred[tid] = 0.f; }
    f32x4_t wacc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) wacc[i][j] = F4Z;
    float bacc = 0.f;
    float yacc[(CV_FEAT + 255) / 256];                                            // d y_layer.bias[tid + 256 k]
#pragma unroll
    for (int k = 0; k < (CV_FEAT + 255) / 256; ++k) yacc[k] = 0.f;
    __syncthreads();
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        // every global load of the image in flight before the first LDS store
        constexpr int NDA = CV_H1 * CV_H1 * 2, NYC = CV_FEAT / 8;
        static_assert(CV_FEAT % 8 == 0, "yc row in 16-byte pieces");
        typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
        uint4 dv[(NDA + 255) / 256];
        u16x8_t yv[(NYC + 255) / 256];
#pragma unroll
        for (int k = 0; k < (NDA + 255) / 256; ++k) {
            const int i = tid + 256 * k;
            dv[k] = i < NDA ? *reinterpret_cast<const uint4*>(da2 + ((size_t)b * CV_H1 * CV_H1 + (i >> 1)) * 16 + (i & 1) * 8)
                            : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int k = 0; k < (NYC + 255) / 256; ++k) {
            const int i8 = tid + 256 * k;
            yv[k] = i8 < NYC ? reinterpret_cast<const u16x8_t*>(yc + (size_t)b * ldy)[i8] : u16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int k = 0; k < (NDA + 255) / 256; ++k) {
            const int i = tid + 256 * k, pix = i >> 1, h = i & 1;
            if (i < NDA) *reinterpret_cast<uint4*>(das + ((pix / CV_H1 + 1) * A1_LD + pix % CV_H1 + 1) * 16 + h * 8) = dv[k];
        }
#pragma unroll
        for (int k = 0; k < (NYC + 255) / 256; ++k) {
            const int i8 = tid + 256 * k;
            if (i8 < NYC) {
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = 8 * i8 + e8;
                    ycs[(e % 81) * 32 + e / 81] = yv[k][e8];
                }
            }
        }
        __syncthreads();
        for (int pix = tid >> 4; pix < CV_H1 * CV_H1; pix += 16)
            bacc += bf2f(das[((pix / CV_H1 + 1) * A1_LD + pix % CV_H1 + 1) * 16 + r16]);          // channel = tid & 15
        // data gradient: d in[ci][i][j] = sum_{ky,kx,co} d a2[co][2i-1+ky][2j-1+kx] w[ci][co][ky][kx]
        for (int t = wave; t < 12; t += 4) {
            const int nt = t / 6, mt = t % 6;
            const int pos = min(mt * 16 + r16, 80), i = pos / 9, j = pos % 9;
            f32x4_t acc = F4Z;
#pragma unroll
            for (int s = 0; s < 8; ++s) {                                     // two taps per k-step, 16 channels each
                const bf16x8_t bw = ld8(cp + CP_T1D + (nt * 16 + r16) * 256 + s * 32 + q * 8);
                const int tap = 2 * s + (q >> 1);
                acc = mfma16(ld8(das + ((2 * i + (tap >> 2)) * A1_LD + 2 * j + (tap & 3)) * 16 + (q & 1) * 8), bw, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * q + r;
                if (m < 81) outs[(nt * 16 + r16) * 81 + m] = f2bf(acc[r]);
            }
        }
        // weight gradient: rows ci (2 tiles), columns (tap, co) (16 tiles: wave, wave + 4, ...), reduction over 81 positions
        for (int s = 0; s < 3; ++s) {
            int offA[8], offB[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = min(s * 32 + q * 8 + e, 81), pc = min(p, 80);
                offA[e] = p * 32;
                offB[e] = ((2 * (pc / 9)) * A1_LD + 2 * (pc % 9)) * 16;
            }
            bf16x8_t a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int oa[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) oa[e] = offA[e] + i * 16 + r16;
                a[i] = gather8(ycs, oa);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tap = wave + 4 * j;                                  // n-tile = tap, column = co
                const int sh = ((tap >> 2) * A1_LD + (tap & 3)) * 16 + r16;
                int ob[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) ob[e] = offB[e] + sh;
                const bf16x8_t bfr = gather8(das, ob);
                wacc[0][j] = mfma16(a[0], bfr, wacc[0][j]);
                wacc[1][j] = mfma16(a[1], bfr, wacc[1][j]);
            }
        }
        __syncthreads();
        for (int i = tid; i < CV_FEAT / 2; i += 256)
            reinterpret_cast<uint32_t*>(dyc + (size_t)b * ldy)[i] = reinterpret_cast<const uint32_t*>(outs)[i];
#pragma unroll
        for (int k = 0; k < (CV_FEAT + 255) / 256; ++k)
            if (tid + 256 * k < CV_FEAT) yacc[k] += bf2f(outs[tid + 256 * k]);
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < (CV_FEAT + 255) / 256; ++k)
        if (tid + 256 * k < CV_FEAT) gby[tid + 256 * k] = yacc[k];
    // d W_t1 in arena order [ci][co][tap] through LDS (the accumulator layout would be 32 scattered 4-byte stores per lane)
    __syncthreads();
    float* stage = wstage;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int tap = wave + 4 * j;
#pragma unroll
            for (int r = 0; r < 4; ++r) stage[((i * 16 + 4 * q + r) * 16 + r16) * 16 + tap] = wacc[i][j][r];
        }
    {   // d b_t1[co = r16]: the four lane groups of a wave by shuffle, the four waves by LDS atomics
        float v = bacc;
        v = xor32_sum(xor16_sum(v));
        if (q == 0) atomicAdd(&bred[r16], v);
    }
    __syncthreads();
    for (int i4 = tid; i4 < 32 * 16 * 16 / 4; i4 += 256)
        reinterpret_cast<float4*>(gw)[i4] = reinterpret_cast<const float4*>(stage)[i4];
    if (tid < 16) gb[tid] = bred[tid];
}

// ------------------------------------------------------------------------------------------------------------
// backward of the convolutional encoder front end.  The forward activations are recomputed from the stored image
// (one number per variable), which is cheaper than keeping [B, 16, 36, 36] + [B, 32, 18, 18] around.
// ------------------------------------------------------------------------------------------------------------
struct EncBwdSmem {                                          // one per image group; two groups share a workgroup's LDS
    float img[IMG_LD * IMG_LD];
    float dimg[CV_D];
    float dfeat[CV_FEAT];
    float w1s[160];
    float red[16 * 10];
    float red2[32];
    __attribute__((aligned(16))) bf16_t a1[A1_LD * A1_LD * CV_C1];
    __attribute__((aligned(16))) bf16_t dz2[A1_LD * A1_LD * CV_C2];
    __attribute__((aligned(16))) bf16_t da1[CV_H1 * CV_H1 * CV_C1];
    __attribute__((aligned(16))) uint8_t am1[CV_H1 * CV_H1 * CV_C1];
};

// blockDim = 256 * ngroups (1 or 2): a group of four waves owns one image at a time (its own LDS tiles); with two groups
// the workgroup still produces ONE partial-gradient row (group 0 stores, group 1 adds).
__global__ __launch_bounds__(512) void k_conv_enc_bwd(const float* __restrict__ img_in, const float* __restrict__ dfeat,
                                                      int ldf, const hlvae_var* __restrict__ vars,
                                                      const float* __restrict__ P, hlvae_dims d,
                                                      const bf16_t* __restrict__ cp, float* __restrict__ dimg_out,
                                                      float* __restrict__ Gpart, long part_stride, long part_lo, int B) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int gi = threadIdx.x >> 8, tid = threadIdx.x & 255;
    EncBwdSmem& sm = reinterpret_cast<EncBwdSmem*>(dsm)[gi];
    // G = this workgroup's row of the partial-gradient buffer, addressed with ARENA offsets (row - part_lo)
    float* __restrict__ G = Gpart + (size_t)blockIdx.x * part_stride - part_lo;
    const int wave = tid >> 6, lane = tid & 63, q = lane >> 4, r16 = lane & 15;
    for (int i = tid; i < IMG_LD * IMG_LD; i += 256) sm.img[i] = 0.f;
    for (int i = tid; i < A1_LD * A1_LD * CV_C1 / 2; i += 256) reinterpret_cast<uint32_t*>(sm.a1)[i] = 0u;
    for (int i = tid; i < A1_LD * A1_LD * CV_C2 / 2; i += 256) reinterpret_cast<uint32_t*>(sm.dz2)[i] = 0u;
    if (tid < 160) { sm.w1s[tid] = tid < 144 ? P[d.o_c1w + tid] : P[d.o_c1b + tid - 144]; sm.red[tid] = 0.f; }
    if (tid < 32) sm.red2[tid] = 0.f;
    bf16x8_t bwd[9];
#pragma unroll
    for (int s = 0; s < 9; ++s) bwd[s] = ld8(cp + CP_C2D + r16 * 288 + s * 32 + q * 8);
    f32x4_t wacc[2][3];                                       // d conv2.weight: rows co, n-tiles (taps) wave, wave + 4, wave + 8
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) wacc[i][j] = F4Z;
    float b2acc[2] = {0.f, 0.f};                              // d conv2.bias for co = r16, 16 + r16
    float w1acc[10];                                          // d conv1.weight[co = tid & 15][9], d conv1.bias
#pragma unroll
    for (int t = 0; t < 10; ++t) w1acc[t] = 0.f;
    __syncthreads();
    const int ngroups = blockDim.x >> 8;                                         // 1 or 2 image groups per workgroup
    const int n_iter = (B + ngroups * (int)gridDim.x - 1) / (ngroups * (int)gridDim.x);   // same trip count for all groups
    for (int it = 0; it < n_iter; ++it) {
        const int b = (it * gridDim.x + blockIdx.x) * ngroups + gi;
        const bool live = b < B;
        const int bb = live ? b : 0;
        // all global loads of the image in flight before the first LDS store (a load-then-store loop waits for every load
        // in turn: 15 k -> 9 k clocks for this stage, clock64() phase timing)
        static_assert(CV_D % 4 == 0 && CV_FEAT % 4 == 0, "float4 staging");
        float4 iv[(CV_D / 4 + 255) / 256], fv[(CV_FEAT / 4 + 255) / 256];
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < (CV_D / 4 + 255) / 256; ++k) {
            const int i4 = tid + 256 * k;
            iv[k] = i4 < CV_D / 4 ? reinterpret_cast<const float4*>(img_in + (size_t)bb * CV_D)[i4] : z4;
        }
#pragma unroll
        for (int k = 0; k < (CV_FEAT / 4 + 255) / 256; ++k) {
            const int i4 = tid + 256 * k;
            fv[k] = (live && i4 < CV_FEAT / 4) ? reinterpret_cast<const float4*>(dfeat + (size_t)bb * ldf)[i4] : z4;
        }
        if (it > 0) {                                   // a1 / dz2 were reused as scratch: restore their zero halos
            for (int i = tid; i < A1_LD * A1_LD * CV_C1 / 2; i += 256) reinterpret_cast<uint32_t*>(sm.a1)[i] = 0u;
            for (int i = tid; i < A1_LD * A1_LD * CV_C2 / 2; i += 256) reinterpret_cast<uint32_t*>(sm.dz2)[i] = 0u;
        }
#pragma unroll
        for (int k = 0; k < (CV_D / 4 + 255) / 256; ++k) {
            const int dd = 4 * (tid + 256 * k);
            if (dd < CV_D) {                             // CV_W % 4 == 0: the four pixels are in one image row
                float* dst = sm.img + (dd / CV_W + 1) * IMG_LD + dd % CV_W + 1;
                dst[0] = iv[k].x; dst[1] = iv[k].y; dst[2] = iv[k].z; dst[3] = iv[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < (CV_FEAT / 4 + 255) / 256; ++k) {
            const int i4 = tid + 256 * k;
            if (i4 < CV_FEAT / 4) reinterpret_cast<float4*>(sm.dfeat)[i4] = fv[k];
        }
        __syncthreads();
        conv1_pool(sm.img, sm.w1s, sm.a1, sm.am1, tid);
        __syncthreads();
        // conv2 again; its epilogue routes d feat to the winning pixel of each pooling window (ReLU gate included)
        conv2_mfma(sm.a1, cp, P + d.o_c2b, wave, lane, [&](int wi, int co, const float (&c)[4]) {
            int sel = 0;
            float best = fmaxf(c[0], 0.f);
#pragma unroll
            for (int r = 1; r < 4; ++r) {
                const float v = fmaxf(c[r], 0.f);
                if (v > best) { best = v; sel = r; }
            }
            const float g = best > 0.f ? sm.dfeat[co * 81 + wi] : 0.f;
            b2acc[co >> 4] += g;
            const int y0 = 2 * (wi / CV_H2), x0 = 2 * (wi % CV_H2);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                sm.dz2[((y0 + (r >> 1) + 1) * A1_LD + x0 + (r & 1) + 1) * CV_C2 + co] = f2bf(r == sel ? g : 0.f);
        });
        __syncthreads();
        // data gradient of conv2 -> d a1 (gated by ReLU(a1) > 0)
        for (int mt = wave; mt < 21; mt += 4) {
            const int pos = min(mt * 16 + r16, CV_H1 * CV_H1 - 1), y = pos / CV_H1, x = pos % CV_H1;
            f32x4_t acc = F4Z;
#pragma unroll
            for (int s = 0; s < 9; ++s)
                acc = mfma16(ld8(sm.dz2 + ((y + s / 3) * A1_LD + x + s % 3) * CV_C2 + q * 8), bwd[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + 4 * q + r;
                if (m < CV_H1 * CV_H1) {
                    const bool on = bf2f(sm.a1[((m / CV_H1 + 1) * A1_LD + m % CV_H1 + 1) * CV_C1 + r16]) > 0.f;
                    sm.da1[m * CV_C1 + r16] = f2bf(on ? acc[r] : 0.f);
                }
            }
        }
        // weight gradient of conv2: rows co, columns (tap, ci), reduction over the 324 pixels
        for (int s = 0; s < 11; ++s) {
            int offA[8], offB[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = s * 32 + q * 8 + e;
                const int pc = min(p, CV_H1 * CV_H1 - 1), y = pc / CV_H1, x = pc % CV_H1;
                offA[e] = p < CV_H1 * CV_H1 ? ((y + 1) * A1_LD + x + 1) * CV_C2 : 0;   // cell (0, 0) is halo = zero
                offB[e] = (y * A1_LD + x) * CV_C1;
            }
            bf16x8_t a[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int oa[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) oa[e] = offA[e] + i * 16 + r16;
                a[i] = gather8(sm.dz2, oa);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int tap = wave + 4 * j;
                if (tap < 9) {                                                 // wave-uniform
                    const int sh = ((tap / 3) * A1_LD + tap % 3) * CV_C1 + r16;
                    int ob[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) ob[e] = offB[e] + sh;
                    const bf16x8_t bfr = gather8(sm.a1, ob);
                    wacc[0][j] = mfma16(a[0], bfr, wacc[0][j]);
                    wacc[1][j] = mfma16(a[1], bfr, wacc[1][j]);
                }
            }
        }
        __syncthreads();
        // conv1: weight gradient and image gradient from the sparse d a1 (only the winner of each pooling window)
        {
            const int co = tid & 15;
            float w[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) w[t] = sm.w1s[co * 9 + t];
            for (int pp = tid >> 4; pp < CV_H1 * CV_H1; pp += 16) {
                const float g = bf2f(sm.da1[pp * CV_C1 + co]);
                const int sel = sm.am1[pp * CV_C1 + co];
                if (g != 0.f && sel < 4) {
                    const int y = 2 * (pp / CV_H1) + (sel >> 1), x = 2 * (pp % CV_H1) + (sel & 1);
                    w1acc[9] += g;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int yy = y + t / 3, xx = x + t % 3;                  // halo coordinates
                        w1acc[t] += g * sm.img[yy * IMG_LD + xx];
                    }
                }
            }
        }
        // image gradient on the MFMA units: S[pos][tap] = sum_co dz1[pos][co] w1[co][tap] (M = 1296 conv1 output positions,
        // N = 9 taps, K = 16 channels), where dz1 is expanded on the fly from the pooled gradient and the winner index (no
        // dense [36][36][16] tile).  S goes to LDS (the a1 / dz2 tiles are free now) in two passes of 19 image rows, and
        // every pixel then GATHERS its nine S values: LDS float atomics run at ~1 lane per 2.5 clocks on this part and the
        // scatter formulations spent 55-90 k clocks here.
        {
            float* S = reinterpret_cast<float*>(sm.a1);                             // [19 * 36][9] floats <= a1 + dz2 (38.4 KB)
            bf16x8_t bw1;                                                            // B[n = tap][k = co], k >= 16 is zero
#pragma unroll
            for (int e = 0; e < 8; ++e)
                bw1[e] = (__bf16)((q < 2 && r16 < 9) ? sm.w1s[(q * 8 + e) * 9 + r16] : 0.f);
            for (int pass = 0; pass < 2; ++pass) {
                const int row0 = pass * 17, pos0 = row0 * CV_W;                     // S rows [row0, row0 + 19)
                for (int mt = wave + (pos0 >> 4); mt * 16 < pos0 + 19 * CV_W; mt += 4) {
                    const int pos = mt * 16 + r16, Y = pos / CV_W, X = pos % CV_W;
                    const int pp = (Y >> 1) * CV_H1 + (X >> 1);
                    const uint32_t r = (uint32_t)((Y & 1) * 2 + (X & 1));
                    bf16x8_t a;
#pragma unroll
                    for (int e = 0; e < 8; ++e) a[e] = (__bf16)0.f;
                    if (q < 2) {
                        const uint2 am = *reinterpret_cast<const uint2*>(sm.am1 + pp * CV_C1 + q * 8);
                        const uint4 g = *reinterpret_cast<const uint4*>(sm.da1 + pp * CV_C1 + q * 8);
                        const uint32_t amw[2] = {am.x, am.y};
                        const uint32_t gw[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const uint32_t sel = (amw[e >> 2] >> (8 * (e & 3))) & 0xffu;
                            const uint16_t gv = (uint16_t)((e & 1) ? (gw[e >> 1] >> 16) : (gw[e >> 1] & 0xffffu));
                            a[e] = __builtin_bit_cast(__bf16, (uint16_t)(sel == r ? gv : 0));
                        }
                    }
                    const f32x4_t acc = mfma16(a, bw1, F4Z);
                    if (r16 < 9) {
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) {
                            const int po = mt * 16 + 4 * q + rr - pos0;
                            if (po >= 0 && po < 19 * CV_W) S[po * 9 + r16] = acc[rr];
                        }
                    }
                }
                __syncthreads();
                for (int pix = tid + pass * 18 * CV_W; pix < (pass + 1) * 18 * CV_W; pix += 256) {   // pixel rows [18 pass, +18)
                    const int y = pix / CV_W, x = pix % CV_W;
                    float v = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) {
                        const int Y = y - t / 3 + 1, X = x - t % 3 + 1;
                        if (Y >= 0 && Y < CV_W && X >= 0 && X < CV_W) v += S[((Y - row0) * CV_W + X) * 9 + t];
                    }
                    sm.dimg[pix] = v;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        // gradient of the image (one number per variable): the representation layer's reduction over the batch is a
        // separate kernel (k_conv_rep_grad)
        if (live)
            for (int dd = tid; dd < CV_D; dd += 256) dimg_out[(size_t)b * CV_D + dd] = sm.dimg[dd];
        __syncthreads();
    }
    // per-group reductions in LDS, then group 0 stores its partial row and group 1 adds to it
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float v = b2acc[i];
        v = xor32_sum(xor16_sum(v));
        if (q == 0) atomicAdd(&sm.red2[i * 16 + r16], v);          // the four waves own different pooling windows
    }
    // d conv1.{weight, bias}: the 4 lane groups of a wave share co = tid & 15 -> shuffles, then one LDS slot per wave
    // (256 x 10 same-address LDS float atomics cost ~6 k clocks here)
    float* w1red = sm.dfeat;                                         // [4 waves][160], free at this point
#pragma unroll
    for (int t = 0; t < 10; ++t) {
        float v = w1acc[t];
        v = xor32_sum(xor16_sum(v));
        if (q == 0) w1red[wave * 160 + r16 * 10 + t] = v;
    }
    // d conv2.weight in arena order [co][ci][tap] through LDS: the accumulator layout would be 24 scattered 4-byte stores per lane
    float* stage = reinterpret_cast<float*>(sm.a1);                 // 4608 floats <= a1 + dz2
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int tap = wave + 4 * j;
            if (tap < 9)
#pragma unroll
                for (int r = 0; r < 4; ++r) stage[((i * 16 + 4 * q + r) * 16 + r16) * 9 + tap] = wacc[i][j][r];
        }
    __syncthreads();
    if (tid < 160) sm.red[tid] = w1red[tid] + w1red[160 + tid] + w1red[320 + tid] + w1red[480 + tid];
    __syncthreads();
    for (int phase = 0; phase < ngroups; ++phase) {
        if (gi == phase) {
            const bool add = phase == 1;
            static_assert((CV_C2 * CV_C1 * 9) % 4 == 0, "float4 rows");
            float4* dst4 = reinterpret_cast<float4*>(G + d.o_c2w);       // (o_c2w and the partial rows are 16-byte aligned: checked on the host)
            for (int i4 = tid; i4 < CV_C2 * CV_C1 * 9 / 4; i4 += 256) {
                float4 v = reinterpret_cast<const float4*>(stage)[i4];
                if (add) { const float4 o = dst4[i4]; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                dst4[i4] = v;
            }
            if (tid < 160) {
                float* dst = G + (tid % 10 < 9 ? d.o_c1w + (tid / 10) * 9 + tid % 10 : d.o_c1b + tid / 10);
                *dst = add ? *dst + sm.red[tid] : sm.red[tid];
            }
            if (tid < 32) {
                float* dst = G + d.o_c2b + tid;
                *dst = add ? *dst + sm.red2[tid] : sm.red2[tid];
            }
        }
        __threadfence_block();            // workgroup scope is enough (same CU, same L1); a device-scope fence writes back L2
        __syncthreads();
    }
}

// representation layer (HLVAE.py:91-102): d w[d][k] = sum_b g[b][d] x_k[b][d], d bias[d] = sum_b g[b][d] over the observed
// rows, x = one-hot (cat) or thermometer (ordinal) of the packed class index.  grid (ceil(D / 256), row chunks)
__device__ __forceinline__ void conv_rep_grad(int bx, int by, int ny, const float* __restrict__ dimg, const float* __restrict__ xt,
                                              const uint8_t* __restrict__ m8, const hlvae_var* __restrict__ vars, int D,
                                              int B, float* __restrict__ G) {
    const int dd = bx * 256 + threadIdx.x;
    if (dd >= D) return;
    const hlvae_var var = vars[dd];
    if (var.kind != HLVAE_CAT && var.kind != HLVAE_ORDINAL) return;
    const int rows = (B + ny - 1) / ny, b0 = by * rows, b1 = min(B, b0 + rows);
    float acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.f;
#pragma unroll 4
    for (int b = b0; b < b1; ++b) {
        const size_t o = (size_t)b * D + dd;
        const float g = m8[o] ? dimg[o] : 0.f;                   // three independent loads per row, no branch around them
        const int cls = (int)xt[o];
        acc[8] += g;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (var.kind == HLVAE_CAT ? k == cls : k <= cls) ? g : 0.f;
    }
    if (acc[8] != 0.f) atomicAdd(G + var.rb_off, acc[8]);
    for (int k = 0; k < var.ncls; ++k)
        if (acc[k] != 0.f) atomicAdd(G + var.r_off + k, acc[k]);
}

// G[lo + i] += sum over the workgroups' partial rows (coalesced across i): replaces one atomic per weight and workgroup,
// which on MI355X serialises in the fabric when 256 workgroups on 8 XCDs hit the same address
__device__ __forceinline__ void conv_wgrad_reduce(int bx, int by, const float* __restrict__ part, int nrows, long n,
                                                  float* __restrict__ G) {
    // (ceil(n / 256), 8) workgroups: workgroup (x, y) sums rows y, y + 8, ... of its 256 columns, then 8-way atomics
    const long i = (long)bx * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = by;
    for (; r + 24 < nrows; r += 32) {
        s0 += part[(size_t)r * n + i];
        s1 += part[(size_t)(r + 8) * n + i];
        s2 += part[(size_t)(r + 16) * n + i];
        s3 += part[(size_t)(r + 24) * n + i];
    }
    for (; r < nrows; r += 8) s0 += part[(size_t)r * n + i];
    const float v = (s0 + s1) + (s2 + s3);
    if (v != 0.f) atomicAdd(G + i, v);
}

// the two independent tails of the convolutional backward pass in ONE launch (as launches of their own they are two
// dependent ~10 us kernels): workgroups [0, 8 nx) fold the partial rows, the rest reduce the representation layer
#define CV_REP_CHUNKS 64
__global__ __launch_bounds__(256) void k_conv_grad_finish(const float* __restrict__ part, int nrows, long n, float* __restrict__ Gcv,
                                                          const float* __restrict__ dimg, const float* __restrict__ xt,
                                                          const uint8_t* __restrict__ m8, const hlvae_var* __restrict__ vars, int D,
                                                          int B, float* __restrict__ G) {
    const int nx = (int)((n + 255) / 256), nred = 8 * nx;
    if ((int)blockIdx.x < nred) {
        conv_wgrad_reduce(blockIdx.x % nx, blockIdx.x / nx, part, nrows, n, Gcv);
    } else {
        const int id = blockIdx.x - nred, nbx = (D + 255) / 256;
        conv_rep_grad(id % nbx, id / nbx, CV_REP_CHUNKS, dimg, xt, m8, vars, D, B, G);
    }
}

// out[c][r] = in[r][c] for a [rows][cols] bf16 matrix (row strides ld_in / ld_out), 64 x 64 tiles through LDS: 16-byte
// loads along the input rows, 16-byte stores along the output rows.  rows, cols need not be multiples of 64; ld_in, ld_out
// multiples of 8.
__global__ __launch_bounds__(256) void k_transpose_bf16(const bf16_t* __restrict__ in, int ld_in, bf16_t* __restrict__ out, int ld_out,
                                                        int rows, int cols) {
    __shared__ bf16_t tile[64][66];
    typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, r = idx >> 3, c8 = (idx & 7) * 8;
        u16x8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
        // (columns up to ru(cols, 8) <= ld_in are the caller's padding)
        if (r0 + r < rows && c0 + c8 < cols) v = *reinterpret_cast<const u16x8_t*>(in + (size_t)(r0 + r) * ld_in + c0 + c8);
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][c8 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, c = idx >> 3, r8 = (idx & 7) * 8;
        if (c0 + c < cols && r0 + r8 < rows) {
            u16x8_t v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tile[r8 + e][c];
            *reinterpret_cast<u16x8_t*>(out + (size_t)(c0 + c) * ld_out + r0 + r8) = v;
        }
    }
}

int hl_launch_transpose_bf16(const bf16_t* in, int ld_in, bf16_t* out, int ld_out, int rows, int cols, const char* label, hipStream_t s) {
    HL_REQUIRE(ld_in % 8 == 0 && ld_out % 8 == 0 && rows % 8 == 0, HLVAE_ESHAPE, "transpose: ld_in=%d ld_out=%d rows=%d", ld_in, ld_out, rows);
    HL_PROF(label, s);
    k_transpose_bf16<<<dim3((cols + 63) / 64, (rows + 63) / 64), 256, 0, s>>>(in, ld_in, out, ld_out, rows, cols);
    HL_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
int hl_conv_pack_weights(const hlvae_plan* p, const hlvae_ws* ws, hipStream_t s) {
    HL_PROF("conv_pack_weights", s);
    k_conv_pack<<<32, 256, 0, s>>>(ws->P, p->d, ws->cpack);
    HL_LAUNCH_CHECK();
    return 0;
}

int hl_launch_conv_enc_fwd(const hlvae_plan* p, const hlvae_ws* ws, const double* data, const double* mask, const float* cvals,
                           const uint8_t* cmask, const int32_t* crows, int B, int Bp, hipStream_t s) {
    HL_PROF("conv_enc_fwd", s);
    k_conv_enc_fwd<<<B, 256, 0, s>>>(data, mask, cvals, cmask, crows, p->vars_dev, ws->P, p->d, ws->sums, ws->norm, ws->cpack, ws->xn,
                                     ws->xt, ws->m8, ws->img, B, Bp);
    HL_LAUNCH_CHECK();
    return hl_launch_transpose_bf16(ws->xn, p->d.Xep, ws->xnT, Bp, Bp, CV_FEAT, "xn_transpose", s);
}

int hl_launch_conv_dec_fwd(const hlvae_plan* p, const hlvae_ws* ws, int B, hipStream_t s) {
    const hlvae_dims& d = p->d;
    {
        HL_PROF("convT1_fwd", s);
        k_convT1_fwd<<<B, 256, 0, s>>>(ws->yc, d.NYlp, ws->cpack, ws->P + d.o_t1b, ws->a2);
        HL_LAUNCH_CHECK();
    }
    {
        HL_PROF("convT2_fwd", s);
        k_convT2_fwd<<<B, 256, 0, s>>>(ws->a2, ws->cpack, ws->P + d.o_t2b, ws->yv, d.NY);
        HL_LAUNCH_CHECK();
    }
    return 0;
}

int hl_launch_conv_dec_bwd(const hlvae_plan* p, const hlvae_ws* ws, int B, int Bp, hipStream_t s) {
    const hlvae_dims& d = p->d;
    const int grid = B < CV_PART_ROWS ? B : CV_PART_ROWS;
    {
        HL_PROF("convT2_bwd", s);
        k_convT2_bwd<<<grid, 256, 0, s>>>(ws->dy, d.NYp, ws->a2, ws->cpack, ws->da2, ws->cvpart, d.cv_n, d.o_cv_lo, d.o_t2w,
                                          d.o_t2b, B);
        HL_LAUNCH_CHECK();
    }
    {
        HL_PROF("convT1_bwd", s);
        k_convT1_bwd<<<grid, 256, 0, s>>>(ws->da2, ws->yc, d.NYlp, ws->cpack, ws->dyc, ws->cvpart, d.cv_n,
                                          d.o_cv_lo, d.o_t1w, d.o_t1b, d.o_by, B);
        HL_LAUNCH_CHECK();
    }
    return 0;
}

int hl_launch_conv_enc_bwd(const hlvae_plan* p, const hlvae_ws* ws, int B, hipStream_t s) {
    const hlvae_dims& d = p->d;
    const int grid = B < CV_PART_ROWS ? B : CV_PART_ROWS;
    static bool attr_set = false;
    if (!attr_set) {
        HL_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_enc_bwd), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(2 * sizeof(EncBwdSmem))));
        attr_set = true;
    }
    {
        HL_PROF("conv_enc_bwd", s);
        // same number of workgroups (= partial rows) as the decoder kernels; a workgroup without images stores zeros
        // one image group (256 threads) per workgroup, two workgroups per CU (77 KB of LDS each); the kernel also runs
        // with 512 threads = two groups sharing one partial row (measured equal within 7 %)
        k_conv_enc_bwd<<<grid, 256, sizeof(EncBwdSmem), s>>>(ws->img, ws->dfeat, d.Xep, p->vars_dev, ws->P, d, ws->cpack,
                                                             ws->dimg, ws->cvpart, d.cv_n, d.o_cv_lo, B);
        HL_LAUNCH_CHECK();
    }
    {   // all three backward kernels have filled their columns of the partial rows: fold them into the gradient arena;
        // beside it, the representation layer's reduction over the batch
        HL_PROF("conv_grad_finish", s);
        const int nx = (int)((d.cv_n + 255) / 256);
        k_conv_grad_finish<<<8 * nx + ((CV_D + 255) / 256) * CV_REP_CHUNKS, 256, 0, s>>>(ws->cvpart, grid, d.cv_n, ws->G + d.o_cv_lo, ws->dimg,
                                                                                      ws->xt, ws->m8, p->vars_dev, d.D, B, ws->G);
        HL_LAUNCH_CHECK();
    }
    return 0;
}
