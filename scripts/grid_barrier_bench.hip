// What a grid-wide exchange costs inside one launch against a kernel boundary (dev microbenchmark, GPU box; round-4 review, Next 6:
// "measure before arguing" -- the PPO minibatch step's reduce (250 workgroups) and optimiser step (11 workgroups, each needing the
// norm of ALL gradients) can share a launch only across such an exchange).
//   hipcc -O3 --offload-arch=gfx950 scripts/grid_barrier_bench.hip -o scripts/grid_barrier_bench && scripts/grid_barrier_bench
// (a) two dependent launches of G workgroups x 1024 threads doing a token amount of work; (b) ONE launch with an atomic-counter barrier
// between the two phases (all G workgroups co-resident: G <= 256); (c) one launch, "last workgroup does phase two" (no spinning: every
// workgroup adds to a counter when its phase-one results are out, the one that arrives last runs phase two alone).
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(1024) void phase(float* buf, int n, float a) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) buf[i] = buf[i] * a + 1.0f;
}

__global__ __launch_bounds__(1024) void fused_barrier(float* buf, int n, float a, unsigned* counter, unsigned epoch) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) buf[i] = buf[i] * a + 1.0f;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(counter, 1u);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * gridDim.x) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __threadfence();
    if (i < n) buf[(i + 4096) % n] = buf[(i + 4096) % n] * a + 1.0f;    // reads what another workgroup wrote in phase one
}

__global__ __launch_bounds__(1024) void fused_last(float* buf, int n, float a, unsigned* counter, unsigned epoch, int tail_n) {
    __shared__ unsigned arrived;
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n) buf[i] = buf[i] * a + 1.0f;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) arrived = atomicAdd(counter, 1u);
    __syncthreads();
    if (arrived + 1 != epoch * gridDim.x) return;
    __threadfence();
    for (int j = threadIdx.x; j < tail_n; j += 1024) buf[j] = buf[j] * a + 1.0f;   // phase two by the last workgroup alone (12 elements per thread)
}

int main() {
    const int G = 250, n = G * 1024;
    float* buf; unsigned* ctr;
    (void)hipMalloc(&buf, sizeof(float) * n); (void)hipMalloc(&ctr, 64);
    (void)hipMemset(buf, 0, sizeof(float) * n); (void)hipMemset(ctr, 0, 64);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int reps = 2000;
    auto timeit = [&](auto&& body) {
        for (int r = 0; r < 50; ++r) body(r);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) body(r);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms * 1e3f / reps;
    };
    const float one = timeit([&](int) { hipLaunchKernelGGL(phase, dim3(G), dim3(1024), 0, 0, buf, n, 0.5f); });
    const float two = timeit([&](int) {
        hipLaunchKernelGGL(phase, dim3(G), dim3(1024), 0, 0, buf, n, 0.5f);
        hipLaunchKernelGGL(phase, dim3(11), dim3(1024), 0, 0, buf, 11 * 1024, 0.5f);
    });
    (void)hipMemset(ctr, 0, 64); (void)hipDeviceSynchronize();
    unsigned ep = 0;
    const float bar = timeit([&](int) { ++ep; hipLaunchKernelGGL(fused_barrier, dim3(G), dim3(1024), 0, 0, buf, n, 0.5f, ctr, ep); });
    (void)hipMemset(ctr, 0, 64); (void)hipDeviceSynchronize();
    ep = 0;
    const float last = timeit([&](int) { ++ep; hipLaunchKernelGGL(fused_last, dim3(G), dim3(1024), 0, 0, buf, n, 0.5f, ctr, ep, 11 * 1024); });
    std::printf("launch-to-launch, us: one launch of %d workgroups %.2f | that + a dependent launch of 11 workgroups %.2f (the boundary costs %.2f) | one launch "
                "with a counter barrier between the phases %.2f (+%.2f) | one launch, last workgroup runs phase two alone %.2f (+%.2f)\n",
                G, one, two, two - one, bar, bar - one, last, last - one);
    return 0;
}
