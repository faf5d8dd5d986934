// Issue rate of v_mfma_f64_16x16x4_f64 on gfx950: N dependent-free MFMAs per wave (8 accumulators round-robin), 1 / 2 / 4 waves per
// SIMD on every CU.  Prints TFLOP/s and clocks per instruction per SIMD (shader clock from the run time and the device's clock rate).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) double f64x4_t;
__global__ void k(double* out, int iters, double a, double b) {
    f64x4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    double x = a + threadIdx.x * 1e-9, y = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}
__global__ void kv(double* out, int iters, double a, double b) {      // vector fp64 FMA, 16 independent chains per lane
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-9 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678) out[0] = s;
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    double* out;
    hipMalloc(&out, 8);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps, grid = p.multiProcessorCount * 2;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<grid, threads>>>(out, 10, 1.0, 2.0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<grid, threads>>>(out, iters, 1.0, 2.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)grid * (threads / 64) * iters * 8, flops = n_mfma * 16 * 16 * 4 * 2;
        const double per_simd = n_mfma / (p.multiProcessorCount * 4.0);
        printf("%d wave(s)/SIMD x 2 workgroups/CU: %.3f ms  %.1f TFLOP/s  %.1f clocks per MFMA per SIMD at %.0f MHz\n", wps, ms, flops / ms * 1e-9,
               ms * 1e-3 * p.clockRate * 1e3 / per_simd, p.clockRate * 1e-3);
    }
    for (int wps : {1, 2, 4}) {
        const int threads = 64 * 4 * wps, grid = p.multiProcessorCount * 2;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        kv<<<grid, threads>>>(out, 10, 1.0000001, 1e-9);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kv<<<grid, threads>>>(out, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n_fma = (double)grid * (threads / 64) * iters * 16, flops = n_fma * 64 * 2;
        const double per_simd = n_fma / (p.multiProcessorCount * 4.0);
        printf("v_fma_f64, %d wave(s)/SIMD x 2 workgroups/CU: %.3f ms  %.1f TFLOP/s  %.1f clocks per wave instruction per SIMD at %.0f MHz\n", wps, ms,
               flops / ms * 1e-9, ms * 1e-3 * p.clockRate * 1e3 / per_simd, p.clockRate * 1e-3);
    }
    return 0;
}
