// What a wave64 VALU instruction costs by the number of vector-register source operands (dev microbenchmark, GPU box):
//   hipcc -O3 --offload-arch=gfx950 scripts/valu_operand_bench.hip -o scripts/valu_operand_bench && scripts/valu_operand_bench
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: v_fma d, v, s, const   1: v_fma d, v, v, s   2: v_fma d, v, v, v   3: v_mul d, v, v   4: v_add d, v, s  5: v_fmac d, v, v (d also source)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    const int lane = threadIdx.x & 63;
    float x[8], y[8], z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = lane + j; y[j] = 1.0f + 1e-6f * (lane + j); z[j] = 0.25f * j; }
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(x[j]) : "s"(a));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "s"(b));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(z[j]));
                if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(y[j]));
                if (MODE == 4) asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[j]) : "s"(b));
                if (MODE == 5) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(z[j]));
                if (MODE == 6) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(x[j]));
                if (MODE == 7) asm volatile("v_mul_f32 %0, 0x3f7fff58, %0" : "+v"(x[j]));
                if (MODE == 8) asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x[j]) : "v"(y[j]));
                if (MODE == 9) asm volatile("v_exp_f32 %0, %0" : "+v"(x[j]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + y[j] + z[j];
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
    return ms * 1e6f / (2048.0f * 16 * 8 * w);   // ns per instruction and SIMD
}

int main() {
    float* d; (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int w : {1, 2, 4, 5}) {
        std::printf("waves/SIMD %d: ns per wave64 instruction and SIMD: fma v,s,const %.2f | fma v,v,s %.2f | fma v,v,v %.2f | mul v,v %.2f | add v,s %.2f | fmac v,v (+d) %.2f | mul v,inline const %.2f | mul v,literal %.2f | fma v,v,inline const %.2f | exp %.2f\n",
                    w, run<0>(d, w), run<1>(d, w), run<2>(d, w), run<3>(d, w), run<4>(d, w), run<5>(d, w), run<6>(d, w), run<7>(d, w), run<8>(d, w), run<9>(d, w));
    }
    return 0;
}
