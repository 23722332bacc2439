// Dependent-instruction latency of the vector ALU as seen by the issue rate (dev microbenchmark, GPU box): a wave that issues C
// independent chains of dependent v_fma_f32 (all sources vector registers), at 1 / 2 / 4 / 5 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 scripts/dep_chain_bench.hip -o scripts/dep_chain_bench && scripts/dep_chain_bench
// ns per instruction and SIMD; where it stays above the 1.2 ns of the fully interleaved case the SIMD waits for results, not for issue slots.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int C, int TRANS>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float x[8], y[8], z[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x[j] = lane + j; y[j] = 1.0f + 1e-6f * (lane + j); z[j] = 0.25f * j; }
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 128 / C; ++u) {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                if (TRANS && (u % 8) == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[j]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(y[j]), "v"(z[j]));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + y[j] + z[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int C, int TRANS>
float run(float* d, int w) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, TRANS>), dim3(256 * w), dim3(256), 0, 0, d, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<C, TRANS>), dim3(256 * w), dim3(256), 0, 0, d, 2048);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (2048.0f * 128 * w);
}

int main() {
    float* d; (void)hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    for (int w : {1, 2, 3, 4, 5}) {
        std::printf("waves/SIMD %d: ns per wave64 v_fma and SIMD by independent chains per wave: 1 chain %.2f | 2 chains %.2f | 4 chains %.2f | 8 chains %.2f | 1 chain, every 8th a v_rcp %.2f | 4 chains, every 8th a v_rcp %.2f\n",
                    w, run<1, 0>(d, w), run<2, 0>(d, w), run<4, 0>(d, w), run<8, 0>(d, w), run<1, 1>(d, w), run<4, 1>(d, w));
    }
    return 0;
}
