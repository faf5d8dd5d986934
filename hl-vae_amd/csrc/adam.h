// Adam arithmetic shared by the optimiser kernels (optim.hip) and the weight-gradient GEMMs that apply the update in their
// epilogue (dense.hip): torch.optim.Adam semantics (reference HLVAE_main.py:277-278), fp32 state.
#pragma once
#include "common.h"

struct AdamScalars {
    float step_size, rs_bc2, b1, b2, eps, gscale;
};

// step_count[0] = completed steps; every Adam kernel of a step uses step_count[0] + 1.  The step number is committed
// through a ticket in step_count[1]: every workgroup of every launch that belongs to the step takes one when it is done,
// and the one that takes the last of `ticket_total` commits.  The launches of a step may therefore run concurrently on
// different streams; no separate "increment" launch is needed.
__device__ __forceinline__ AdamScalars adam_scalars(float t, float lr, float b1, float b2, float eps, float gscale) {
    AdamScalars a;
    a.step_size = lr / (1.f - powf(b1, t));
    a.rs_bc2 = rsqrtf(1.f - powf(b2, t));
    a.b1 = b1; a.b2 = b2; a.eps = eps; a.gscale = gscale;
    return a;
}

__device__ __forceinline__ float adam_one(float p, float g, float& m, float& v, const AdamScalars& a) {
    g *= a.gscale;
    m = a.b1 * m + (1.f - a.b1) * g;
    v = a.b2 * v + (1.f - a.b2) * g * g;
    return p - a.step_size * m / (sqrtf(v) * a.rs_bc2 + a.eps);
}

