// Do fp32 MFMAs of one wave overlap with VALU work of another wave on the same SIMD?  (dev tool)
// mode 0: every wave runs MFMA chains; mode 1: every wave runs VALU FMAs / transcendentals;
// mode 2: even workgroups MFMA, odd workgroups VALU (2 waves per SIMD, one of each).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float* out, int iters, int mode, int trans, float a, float b) {
    const bool do_mfma = mode == 0 || (mode == 2 && (blockIdx.x & 1) == 0);
    float s = 0;
    if (do_mfma) {
        f32x16 acc0, acc1;
        for (int r = 0; r < 16; ++r) { acc0[r] = threadIdx.x; acc1[r] = r; }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    } else {
        float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (trans) {
                    x0 = __builtin_amdgcn_exp2f(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_exp2f(x2); x3 = __builtin_amdgcn_rcpf(x3);
                } else {
                    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                }
            }
        }
        s = x0 + x1 + x2 + x3;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
float run(float* out, int blocks, int mode, int trans) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode, trans, 1.0f, 1e-9f);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, mode, trans, 1.0f, 1e-9f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    for (int trans : {0, 1}) {
        const float a = run(out, 256, 0, trans), b = run(out, 256, 1, trans), c = run(out, 512, 2, trans);
        printf("%s: MFMA only (1 wave/SIMD) %.3f ms, VALU only (1 wave/SIMD) %.3f ms, one of each per SIMD %.3f ms -> %s\n",
               trans ? "transcendental VALU" : "plain FMA VALU", a, b, c, c < 0.5f * (a + b) + 0.5f * (a > b ? a : b) ? "they overlap" : "they serialise");
    }
    return 0;
}
