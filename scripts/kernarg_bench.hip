// kernarg_bench.hip -- dev tool (GPU box): what does the SIZE of r6_step's by-value kernel arguments cost per launch?
// (round-3 review, Next 2c: "kernarg diet": R6Params travels by value, ~110 dwords; park it in a device buffer and pass a pointer?)
//   hipcc -O3 --offload-arch=gfx950 -o scripts/kernarg_bench scripts/kernarg_bench.hip && scripts/kernarg_bench
// Three kernels with the step kernel's launch shape (1280 x 256 threads), each streaming `bytes` of HBM traffic (copy) so that a
// launch looks like a step to the dispatcher, back to back on one stream:
//   val   : a 656-byte struct by value (R6Params 440 B + R6Buffers 160 B + scalars), 48 dwords of it read by every wave (s_load
//           from the kernarg segment)
//   ptr   : the same struct in device memory, a pointer in the kernarg segment, the same 48 dwords read through it (s_load)
//   small : 24 bytes of kernel arguments, nothing else read
// Reported: microseconds per launch (HIP events around 2000 launches), for bytes = 0 (launch floor), 16 MB and 64 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

struct Big { float f[164]; };   // 656 bytes

__device__ __forceinline__ float sum48(const __attribute__((address_space(4))) float* k) {
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += k[(i * 3) % 164];
    return s;
}
__device__ __forceinline__ void body(const float4* src, float4* dst, long n4, float bias) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = src[i];
        v.x += bias;
        dst[i] = v;
    }
}
__global__ __launch_bounds__(256) void k_val(const Big p, const float4* src, float4* dst, long n4) {
    const __attribute__((address_space(4))) float* k = (const __attribute__((address_space(4))) float*)__builtin_amdgcn_kernarg_segment_ptr();
    body(src, dst, n4, sum48(k) * 1e-30f);
}
__global__ __launch_bounds__(256) void k_ptr(const Big* p, const float4* src, float4* dst, long n4) {
    body(src, dst, n4, sum48((const __attribute__((address_space(4))) float*)p) * 1e-30f);
}
__global__ __launch_bounds__(256) void k_small(const float4* src, float4* dst, long n4) { body(src, dst, n4, 0.0f); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

int main() {
    const long maxb = 64l << 20;
    float4 *src, *dst;
    Big* dp;
    CK(hipMalloc(&src, maxb));
    CK(hipMalloc(&dst, maxb));
    CK(hipMalloc(&dp, sizeof(Big)));
    CK(hipMemset(src, 0, maxb));
    Big hb;
    for (int i = 0; i < 164; ++i) hb.f[i] = (float)i;
    CK(hipMemcpy(dp, &hb, sizeof(Big), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int launches = 2000, rounds = 9;
    for (long bytes : {0l, 16l << 20, 64l << 20}) {
        const long n4 = bytes / 16;
        std::vector<float> t[3];
        for (int r = 0; r < rounds; ++r) {
            for (int which = 0; which < 3; ++which) {
                const int w = (r & 1) ? 2 - which : which;
                for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < (pass ? launches : 100); ++i) {
                        if (w == 0) hipLaunchKernelGGL(k_val, dim3(1280), dim3(256), 0, 0, hb, src, dst, n4);
                        else if (w == 1) hipLaunchKernelGGL(k_ptr, dim3(1280), dim3(256), 0, 0, dp, src, dst, n4);
                        else hipLaunchKernelGGL(k_small, dim3(1280), dim3(256), 0, 0, src, dst, n4);
                    }
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (pass) t[w].push_back(ms * 1e3f / launches);
                }
            }
        }
        const char* names[3] = {"val (656 B by value, 48 dwords read)", "ptr (pointer to device copy, 48 dwords read)", "small (24 B, nothing read)"};
        for (int w = 0; w < 3; ++w) {
            std::sort(t[w].begin(), t[w].end());
            std::printf("bytes/launch %3ld MB  %-48s us/launch median %.3f min %.3f max %.3f\n", 2 * bytes >> 20, names[w], t[w][rounds / 2], t[w][0], t[w][rounds - 1]);
        }
    }
    return 0;
}
