// Issue cost of packed-fp32 instructions against plain ones, all sources in vector registers (dev microbenchmark, GPU box;
// round-4 review, Next 3: "hand-packed fp32 inside one env"):
//   hipcc -O3 --offload-arch=gfx950 scripts/pk_rate_bench.hip -o scripts/pk_rate_bench && scripts/pk_rate_bench
// Per mode: ns per wave64 INSTRUCTION and SIMD at 1 / 2 / 4 / 5 resident waves per SIMD.  A packed instruction does two
// float operations per lane: it pays only if it costs less than two plain instructions (2 x the `fma v,v,v` column).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: v_fma_f32 v,v,v   1: v_pk_fma_f32 v2,v2,v2   2: v_pk_mul_f32 v2,v2   3: v_pk_add_f32 v2,v2
//      4: v_pk_fma_f32 with a scalar-pair source   5: v_fma_f32 v,v,s   6: two independent plain fmas on the halves of the pairs (same registers as mode 1)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    const int lane = threadIdx.x & 63;
    f2 x[8], y[8], z[8];
    f2 sp = {a, b};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] = f2{(float)(lane + j), (float)(lane - j)};
        y[j] = f2{1.0f + 1e-6f * (lane + j), 1.0f - 1e-6f * (lane + j)};
        z[j] = f2{0.25f * j, 0.125f * j};
    }
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].x) : "v"(y[j].x), "v"(z[j].x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(z[j]));
                if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(y[j]));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[j]) : "v"(z[j]));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "s"(sp), "v"(z[j]));
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].x) : "v"(y[j].x), "s"(b));
                if (MODE == 6) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].x) : "v"(y[j].x), "v"(z[j].x));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j].y) : "v"(y[j].y), "v"(z[j].y));
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j].x + x[j].y + y[j].x + z[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
float run(float* d, int w) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, 64, 1.0001f, 0.5f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, 2048, 1.0001f, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    const float insts = (MODE == 6) ? 2.0f : 1.0f;
    return ms * 1e6f / (2048.0f * 16 * 8 * w * insts);   // ns per instruction and SIMD
}

int main() {
    float* d; (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int w : {1, 2, 4, 5}) {
        std::printf("waves/SIMD %d: ns per wave64 instruction and SIMD: fma v,v,v %.2f | pk_fma v2,v2,v2 %.2f | pk_mul %.2f | pk_add %.2f | pk_fma v2,s2,v2 %.2f | fma v,v,s %.2f | plain fma on pair halves %.2f\n",
                    w, run<0>(d, w), run<1>(d, w), run<2>(d, w), run<3>(d, w), run<4>(d, w), run<5>(d, w), run<6>(d, w));
    }
    return 0;
}
