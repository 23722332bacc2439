// v_mfma_f32_32x32x2_f32 issue rate (dev tool): dependent chain vs 2 / 4 independent accumulators, 1..4 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = threadIdx.x * 1e-3f + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(float* out, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-9f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 1e-9f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * 4 * iters * 8 * CHAINS;
    printf("chains %d, %d waves/SIMD: %.1f TFLOP/s, %.1f cycles per MFMA per SIMD (2.4 GHz)\n", CHAINS, waves_per_simd,
           n_mfma * 4096 / (ms * 1e-3) / 1e12, ms * 1e-3 * 2.4e9 / (n_mfma / 1024));
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int w : {1, 2, 3, 4}) { run<1>(out, w); run<2>(out, w); run<4>(out, w); }
    return 0;
}
