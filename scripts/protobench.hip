// Prototype (dev tool): the step kernel's memory pattern + RK4 integration only (no reward / events / resets), one
// env per lane with plain VALU ops vs two envs per lane as float2 (packed fp32 math, 8-byte loads/stores).
// Answers: how much of the packed-math gain of scripts/pkbench.hip survives next to the HBM traffic of a step?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define DEV __device__ __forceinline__
DEV float vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV f2 vfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
DEV float vrcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV f2 vrcp(f2 x) { return f2{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)}; }
DEV float vsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV f2 vsqrt(f2 x) { return f2{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)}; }
DEV float vexp2(float x) { return __builtin_amdgcn_exp2f(x); }
DEV f2 vexp2(f2 x) { return f2{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)}; }
DEV float vlog2(float x) { return __builtin_amdgcn_logf(x); }
DEV f2 vlog2(f2 x) { return f2{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)}; }
template <class T> DEV T splat(float c);
template <> DEV float splat<float>(float c) { return c; }
template <> DEV f2 splat<f2>(float c) { return f2{c, c}; }
template <class T> struct Ctl { T tbx, tby, tbz, mdot, inv_jt, jdiff; };

template <class T>
DEV void rhs(const T (&y)[13], T im, const Ctl<T>& c, T (&dy)[13]) {
    const T vx = y[3], vy = y[4], vz = y[5], q0 = y[6], q1 = y[7], q2 = y[8], q3 = y[9], wx = y[10], wy = y[11], wz = y[12];
    const T n2 = vfma(q0, q0, vfma(q1, q1, vfma(q2, q2, q3 * q3)));
    const T s = splat<T>(2.0f) * vrcp(n2);
    const T s1 = s * q1, s2 = s * q2, s3 = s * q3;
    const T w1 = s1 * q0, w2 = s2 * q0, w3 = s3 * q0, x1 = s1 * q1, x2 = s2 * q1, x3 = s3 * q1, y2 = s2 * q2, y3 = s3 * q2, z3 = s3 * q3;
    const T one = splat<T>(1.0f);
    const T r00 = one - (y2 + z3), r01 = x2 - w3, r02 = x3 + w2, r10 = x2 + w3, r11 = one - (x1 + z3), r12 = y3 - w1,
            r20 = x3 - w2, r21 = y3 + w1, r22 = one - (x1 + y2);
    const T speed = vsqrt(vfma(vx, vx, vfma(vy, vy, vz * vz)));
    const T base = vfma(y[0], splat<T>(-2.2558e-5f), one);
    const T rho = splat<T>(1.225f) * vexp2(splat<T>(4.2559f) * vlog2(base));
    const T ca = splat<T>(-4.3f) * rho * speed;
    const T aby = ca * vfma(r01, vx, vfma(r11, vy, r21 * vz));
    const T abz = ca * vfma(r02, vx, vfma(r12, vy, r22 * vz));
    const T fx = vfma(r00, c.tbx, vfma(r01, c.tby, vfma(r02, c.tbz, ca * vx)));
    const T fy = vfma(r10, c.tbx, vfma(r11, c.tby, vfma(r12, c.tbz, ca * vy)));
    const T fz = vfma(r20, c.tbx, vfma(r21, c.tby, vfma(r22, c.tbz, ca * vz)));
    dy[0] = vx; dy[1] = vy; dy[2] = vz;
    dy[3] = vfma(fx, im, splat<T>(-9.81f)); dy[4] = fy * im; dy[5] = fz * im;
    const T h = splat<T>(0.5f);
    dy[6] = splat<T>(-0.5f) * vfma(wx, q1, vfma(wy, q2, wz * q3));
    dy[7] = h * vfma(wx, q0, vfma(wz, q2, -wy * q3));
    dy[8] = h * vfma(wy, q0, vfma(-wz, q1, wx * q3));
    dy[9] = h * vfma(wz, q0, vfma(wy, q1, -wx * q2));
    const T tauy = vfma(splat<T>(15.0f), c.tbz, splat<T>(-5.0f) * abz);
    const T tauz = vfma(splat<T>(-15.0f), c.tby, splat<T>(5.0f) * aby);
    const T g = c.jdiff * wx;
    dy[10] = splat<T>(0.0f);
    dy[11] = (tauy - g * wz) * c.inv_jt;
    dy[12] = (tauz + g * wy) * c.inv_jt;
}

template <class T>
DEV void rk4(T (&y)[13], T mass, const Ctl<T>& c) {
    const T hstep = splat<T>(0.1f), hh = splat<T>(0.05f), h6 = splat<T>(0.1f / 6.0f);
    T yt[13], y1[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) { yt[i] = y[i]; y1[i] = y[i]; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const T tm = (s == 0) ? splat<T>(0.0f) : ((s == 3) ? hstep : hh);
        const T w = (s == 0 || s == 3) ? h6 : splat<T>(2.0f) * h6;
        const T an = (s == 2) ? hstep : hh;
        T kk[13];
        rhs<T>(yt, vrcp(vfma(-c.mdot, tm, mass)), c, kk);
#pragma unroll
        for (int i = 0; i < 13; ++i) {
            if (i == 10) continue;
            y1[i] = vfma(w, kk[i], y1[i]);
            yt[i] = vfma(an, kk[i], y[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 13; ++i) y[i] = y1[i];
}

// EPL = envs per lane (1: float, 2: f2).  Tile = 64 * EPL envs per wave; state SoA [14][ld]; obs AoS via LDS.
template <class T, int EPL, int WAVES, bool INTEG>
__global__ __launch_bounds__(256, WAVES) void k(float* st, const float* ep, const float* act, float* obs, float* rew, uint8_t* done,
                                                int64_t n, int64_t ld) {
    __shared__ __attribute__((aligned(16))) float tiles[4][64 * EPL * 14];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* tile = tiles[wave];
    constexpr int TE = 64 * EPL;
    const int64_t nt = n / TE, nw = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + wave; t < nt; t += nw) {
        const int64_t e0 = t * TE;
        T y[13], mass, m0;
        float a[EPL][3];
#pragma unroll
        for (int j = 0; j < 13; ++j) y[j] = *(const T*)&st[j * ld + e0 + lane * EPL];
        mass = *(const T*)&st[13 * ld + e0 + lane * EPL];
        m0 = *(const T*)&ep[e0 + lane * EPL];
#pragma unroll
        for (int q = 0; q < EPL; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) a[q][c] = act[(e0 + lane * EPL + q) * 3 + c];
        Ctl<T> c;
        if constexpr (EPL == 1) {
            c = Ctl<T>{a[0][2] * 5.0e5f, a[0][0] * 1.0e4f, a[0][1] * -2.0e4f, splat<T>(150.0f), vrcp(m0 * 134.0f), m0 * -130.0f};
        } else {
            c = Ctl<T>{T{a[0][2], a[1][2]} * 5.0e5f, T{a[0][0], a[1][0]} * 1.0e4f, T{a[0][1], a[1][1]} * -2.0e4f, splat<T>(150.0f),
                       vrcp(m0 * 134.0f), m0 * -130.0f};
        }
        if constexpr (INTEG) rk4<T>(y, mass, c);
        else {
#pragma unroll
            for (int j = 0; j < 13; ++j) y[j] = vfma(y[j], splat<T>(1.0001f), c.tby * 1e-9f);
        }
        mass = vfma(-c.mdot, splat<T>(0.1f), mass);
        const T r = y[0] * 1e-3f + c.tbx * 1e-8f;
        *(T*)&rew[e0 + lane * EPL] = r;
#pragma unroll
        for (int j = 0; j < 13; ++j) *(T*)&st[j * ld + e0 + lane * EPL] = y[j];
        *(T*)&st[13 * ld + e0 + lane * EPL] = mass;
        if constexpr (EPL == 1) {
            done[e0 + lane] = (y[0] < 0.0f) ? 1 : 0;
#pragma unroll
            for (int j = 0; j < 13; ++j) tile[lane * 14 + j] = y[j] * 0.5f;
            tile[lane * 14 + 13] = mass * 2e-5f;
        } else {
            *(uint16_t*)&done[e0 + lane * 2] = (uint16_t)((y[0].x < 0.0f ? 1 : 0) | ((y[0].y < 0.0f ? 1 : 0) << 8));
#pragma unroll
            for (int j = 0; j < 13; ++j) { tile[(lane * 2) * 14 + j] = y[j].x * 0.5f; tile[(lane * 2 + 1) * 14 + j] = y[j].y * 0.5f; }
            tile[(lane * 2) * 14 + 13] = mass.x * 2e-5f; tile[(lane * 2 + 1) * 14 + 13] = mass.y * 2e-5f;
        }
        __builtin_amdgcn_wave_barrier();
        constexpr int n4 = TE * 14 / 4;
#pragma unroll
        for (int kq = 0; kq < (n4 + 63) / 64; ++kq) {
            const int q = lane + 64 * kq;
            if (q < n4) *(f4*)(obs + e0 * 14 + 4 * q) = *(f4*)(tile + 4 * q);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <class K>
void run(const char* name, K kern, int blocks, float* st, float* ep, float* act, float* obs, float* rew, uint8_t* done, int64_t n, int64_t ld) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, st, ep, act, obs, rew, done, n, ld);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 50; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, st, ep, act, obs, rew, done, n, ld);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0);
    printf("%-44s blocks %5d (resident %d/CU): %7.2f us per step of 2^20 envs\n", name, blocks, occ, ms / 50 * 1e3);
}

int main() {
    const int64_t n = 1 << 20, ld = n + 1024;
    float *st, *ep, *act, *obs, *rew; uint8_t* done;
    (void)hipMalloc(&st, 14 * ld * 4); (void)hipMalloc(&ep, 2 * ld * 4); (void)hipMalloc(&act, n * 3 * 4);
    (void)hipMalloc(&obs, n * 14 * 4); (void)hipMalloc(&rew, n * 4); (void)hipMalloc(&done, n);
    float* h = (float*)malloc(14 * ld * 4);
    for (int64_t i = 0; i < 14 * ld; ++i) h[i] = 0.01f * (i % 97);
    for (int64_t i = 0; i < ld; ++i) { h[i] = 1500.f; h[6 * ld + i] = 1.f; h[13 * ld + i] = 4e4f; }
    (void)hipMemcpy(st, h, 14 * ld * 4, hipMemcpyHostToDevice);
    for (int64_t i = 0; i < 2 * ld; ++i) h[i] = 4e4f;
    (void)hipMemcpy(ep, h, 2 * ld * 4, hipMemcpyHostToDevice);
    for (int64_t i = 0; i < 3 * n; ++i) h[i] = 0.3f;
    (void)hipMemcpy(act, h, 3 * n * 4, hipMemcpyHostToDevice);
#define RUN(NAME, T, EPL, W, I, BL) run(NAME, k<T, EPL, W, I>, BL, st, ep, act, obs, rew, done, n, ld)
    RUN("memory pattern only, 1 env/lane, 5 waves", float, 1, 5, false, 1280);
    RUN("memory pattern only, 2 envs/lane, 4 waves", f2, 2, 4, false, 1024);
    RUN("+ RK4, 1 env/lane scalar, 5 waves", float, 1, 5, true, 1280);
    RUN("+ RK4, 1 env/lane scalar, 8 waves", float, 1, 8, true, 2048);
    RUN("+ RK4, 2 envs/lane packed, 2 waves", f2, 2, 2, true, 512);
    RUN("+ RK4, 2 envs/lane packed, 3 waves", f2, 2, 3, true, 768);
    RUN("+ RK4, 2 envs/lane packed, 4 waves", f2, 2, 4, true, 1024);
    return 0;
}
