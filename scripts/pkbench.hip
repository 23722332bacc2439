// Packed-fp32 probe (dev tool): RK4 steps of the 6-DOF dynamics from registers, one env per lane with
// plain VALU ops vs two envs per lane as float2 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32).
// Prints env-steps/s per variant and occupancy.  hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
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

template <class T>
struct Ctl { T tbx, tby, tbz, mdot, inv_jt, jdiff; };

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

template <class T, int WAVES, bool UNROLL = false>
__global__ __launch_bounds__(256, WAVES) void k(float* out, int iters, float seedv) {
    T y[13], mass;
    const float t = threadIdx.x * 1e-3f + seedv;
    for (int i = 0; i < 13; ++i) y[i] = splat<T>(0.1f * i + t);
    y[0] = splat<T>(1500.0f + t); y[6] = splat<T>(1.0f);
    if constexpr (sizeof(T) == 8) { y[3].y += 1.0f; y[7].y += 0.01f; }
    mass = splat<T>(40000.0f);
    Ctl<T> c{splat<T>(5.0e5f), splat<T>(1.0e4f + t), splat<T>(-2.0e4f), splat<T>(150.0f), splat<T>(2.0e-7f), splat<T>(-4.0e6f)};
    const T hstep = splat<T>(0.1f), hh = splat<T>(0.05f), h6 = splat<T>(0.1f / 6.0f);
    for (int it = 0; it < iters; ++it) {
        T yt[13], y1[13];
        for (int i = 0; i < 13; ++i) { yt[i] = y[i]; y1[i] = y[i]; }
        auto stage = [&](int s) __attribute__((always_inline)) {
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
        };
        if constexpr (UNROLL) {
            stage(0); stage(1); stage(2); stage(3);
        } else {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) stage(s);
        }
        for (int i = 0; i < 13; ++i) y[i] = y1[i];
        mass = vfma(-c.mdot, hstep, mass);
    }
    T acc = mass;
    for (int i = 0; i < 13; ++i) acc += y[i];
    if constexpr (sizeof(T) == 8) out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y;
    else out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class K>
void run(const char* name, K kern, int blocks, int envs_per_lane, float* out) {
    const int iters = 200;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, 0);
    const double steps = 5.0 * blocks * 256.0 * envs_per_lane * iters;
    printf("%-28s blocks %5d (occupancy %d blocks/CU)  %.3f ms  -> %.2f G env-steps/s (integration only)\n", name, blocks, occ,
           ms / 5, steps / (ms * 1e-3) / 1e9);
}
int main() {
    float* out; (void)hipMalloc(&out, 4096 * 256 * 4);
    run("scalar, 5 waves/SIMD", k<float, 5>, 1280, 1, out);
    run("scalar, 8 waves/SIMD", k<float, 8>, 2048, 1, out);
    run("packed x2, 2 waves/SIMD", k<f2, 2>, 512, 2, out);
    run("packed x2, 3 waves/SIMD", k<f2, 3>, 768, 2, out);
    run("packed x2, 4 waves/SIMD", k<f2, 4>, 1024, 2, out);
    run("scalar unrolled, 5 waves", k<float, 5, true>, 1280, 1, out);
    run("scalar unrolled, 8 waves", k<float, 8, true>, 2048, 1, out);
    run("packed unrolled, 3 waves", k<f2, 3, true>, 768, 2, out);
    run("packed unrolled, 4 waves", k<f2, 4, true>, 1024, 2, out);
    return 0;
}
