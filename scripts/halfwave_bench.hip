// Does a wave64 VALU instruction cost less when its upper 32 lanes are masked off?  (dev microbenchmark, GPU box)
//   hipcc -O3 --offload-arch=gfx950 scripts/halfwave_bench.hip -o scripts/halfwave_bench && scripts/halfwave_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>   // 0: all 64 lanes, 1: lanes 0..31 only, 2: lanes 0..15 only, 3: even lanes only
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    const int lane = threadIdx.x & 63;
    float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
    const bool on = MODE == 0 ? true : MODE == 1 ? lane < 32 : MODE == 2 ? lane < 16 : (lane & 1) == 0;
    if (on) {
#pragma unroll 1
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
                x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

template <int MODE>
float run(float* d, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd;   // 4 waves per block -> one wave per SIMD per `waves_per_simd`
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 64, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 2048, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d; hipMalloc(&d, sizeof(float) * 256 * 256 * 8);
    const double n_inst = 2048.0 * 16 * 8;   // VALU instructions per wave
    for (int w : {1, 2, 4}) {
        const float t0 = run<0>(d, w), t1 = run<1>(d, w), t2 = run<2>(d, w), t3 = run<3>(d, w);
        std::printf("waves/SIMD %d: all lanes %.3f ms (%.2f ns per instruction and SIMD), lanes 0..31 %.3f ms (x%.2f), lanes 0..15 %.3f ms (x%.2f), even lanes %.3f ms (x%.2f)\n",
                    w, t0, t0 * 1e6 / (n_inst * w), t1, t1 / t0, t2, t2 / t0, t3, t3 / t0);
    }
    return 0;
}
