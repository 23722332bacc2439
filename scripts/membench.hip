// Memory-floor microbenchmarks (dev tool): same traffic as one env step (189 B/env), trivial ALU.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));

// (a) our access pattern: per lane 14+1 dword loads (SoA), 3 action dwords, 14 dword stores, reward, done byte, obs via LDS float4
__global__ __launch_bounds__(256) void k_pattern(const float* st, float* st_out, const float* ep, const float* act, float* obs,
                                                 float* rew, uint8_t* done, int64_t n, int64_t ld, int tiled) {
    __shared__ __attribute__((aligned(16))) float tiles[4][64 * 14];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = tiles[wave];
    const int64_t nt = n >> 6, nw = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < nt; t += nw) {
        const int64_t e0 = t * 64;
        float y[14];
        // tiled != 0: state as [tile][14][64] (a tile's 14 row segments contiguous) instead of SoA [14][ld]
        const int64_t sbase = tiled ? e0 * 14 : e0, sstride = tiled ? 64 : ld;
#pragma unroll
        for (int j = 0; j < 14; ++j) y[j] = st[j * sstride + sbase + lane];
        const float m0 = ep[e0 + lane];
        const float a0 = act[(e0 + lane) * 3], a1 = act[(e0 + lane) * 3 + 1], a2 = act[(e0 + lane) * 3 + 2];
#pragma unroll
        for (int j = 0; j < 14; ++j) y[j] = y[j] * 1.0001f + a0 * 1e-6f;
        rew[e0 + lane] = m0 + a1 + a2;
        done[e0 + lane] = (y[0] > 1e30f) ? 1 : 0;
#pragma unroll
        for (int j = 0; j < 14; ++j) st_out[j * sstride + sbase + lane] = y[j];
#pragma unroll
        for (int j = 0; j < 14; ++j) tile[lane * 14 + j] = y[j] * 0.5f;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = lane + 64 * k;
            if (q < 224) *(f4*)(obs + e0 * 14 + 4 * q) = *(f4*)(tile + 4 * q);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// (b) plain float4 streaming copy of the same byte counts: reads 72 B/env, writes 117 B/env
__global__ __launch_bounds__(256) void k_copy(const f4* in, f4* out, int64_t n_in4, int64_t n_out4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in4; i += stride) acc += in[i];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out4; i += stride) out[i] = acc;
}
// (c) interleaved read+write float4 copy: each iteration reads 1 f4 and writes ~1.6 f4
__global__ __launch_bounds__(256) void k_copy2(const f4* in, f4* out, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = in[i] * 1.0001f;
}
extern "C" {
int mb_pattern(const float* st, float* st_out, const float* ep, const float* act, float* obs, float* rew, uint8_t* done,
               int64_t n, int64_t ld, int blocks, void* s, int tiled) {
    hipLaunchKernelGGL(k_pattern, dim3(blocks), dim3(256), 0, (hipStream_t)s, st, st_out, ep, act, obs, rew, done, n, ld, tiled);
    return (int)hipGetLastError();
}
int mb_copy(const void* in, void* out, int64_t n_in4, int64_t n_out4, int blocks, void* s) {
    hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f4*)in, (f4*)out, n_in4, n_out4);
    return (int)hipGetLastError();
}
int mb_copy2(const void* in, void* out, int64_t n4, int blocks, void* s) {
    hipLaunchKernelGGL(k_copy2, dim3(blocks), dim3(256), 0, (hipStream_t)s, (const f4*)in, (f4*)out, n4);
    return (int)hipGetLastError();
}
}
