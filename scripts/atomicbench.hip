// Throughput of device-scope fetch-add on one address / on 8 addresses (one per XCD) from 5120 waves:
// decides whether the tile loop can take its tiles from a global counter.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* ctr, int per_wave, int stride, unsigned* sink) {
    unsigned acc = 0;
    if ((threadIdx.x & 63) == 0) {
        unsigned* c = ctr + (blockIdx.x % 8) * stride;
        for (int i = 0; i < per_wave; ++i) acc += atomicAdd(c, 1u);
        if (acc == 0xFFFFFFFFu) *sink = acc;
    }
}
int main() {
    unsigned *ctr, *sink;
    hipMalloc(&ctr, 4096);
    hipMalloc(&sink, 4);
    hipMemset(ctr, 0, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int stride : {0, 64}) {
        for (int per_wave : {1, 4, 16}) {
            for (int rep = 0; rep < 3; ++rep) k<<<1280, 256>>>(ctr, per_wave, stride, sink);
            hipEventRecord(e0);
            for (int rep = 0; rep < 20; ++rep) k<<<1280, 256>>>(ctr, per_wave, stride, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s, %2d fetch-adds per wave (5120 waves): %.2f us per launch -> %.1f ns per atomic\n",
                   stride ? "8 addresses" : "1 address", per_wave, ms * 1e3 / 20, ms * 1e6 / 20 / (5120.0 * per_wave));
        }
    }
    return 0;
}
