// r6_device_pk.hpp -- the dynamics of r6_device.hpp for TWO environments per lane, as float2 values.
//
// Why: the step kernel is co-bound by VALU issue and HBM.  gfx950 issues packed fp32
// (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two lanes' worth of work per instruction at ~1.3x the cost of a
// plain one), and two envs per lane make every row access an 8-byte load/store.  The formulas and their operation
// order are those of r6_device.hpp (rhs_dyn, rhs_quat, step_map, integrate -- citations there); only the value
// type changes.  Transcendentals have no packed form and are issued per half.  Everything that is control flow
// per env (event root, reward, termination, reset) stays scalar and is applied to the two halves in turn.
#pragma once

#include "r6_device.hpp"

namespace r6 {
namespace pk {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float c) { return f2{c, c}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 rcp2(f2 x) { return f2{frcp(x.x), frcp(x.y)}; }
__device__ __forceinline__ f2 rsq2(f2 x) { return f2{frsq(x.x), frsq(x.y)}; }
__device__ __forceinline__ f2 sqrt2(f2 x) { return f2{fsqrt(x.x), fsqrt(x.y)}; }
__device__ __forceinline__ f2 exp2_2(f2 x) { return f2{fexp2(x.x), fexp2(x.y)}; }
__device__ __forceinline__ f2 log2_2(f2 x) { return f2{flog2(x.x), flog2(x.y)}; }

struct Control2 {
    f2 tbx, tby, tbz, mdot;
};
struct Inertia2 {
    f2 inv_jt, jdiff;
};

__device__ __forceinline__ Control2 pack(const Control& a, const Control& b) {
    return Control2{f2{a.tbx, b.tbx}, f2{a.tby, b.tby}, f2{a.tbz, b.tbz}, f2{a.mdot, b.mdot}};
}
__device__ __forceinline__ Inertia2 pack(const Inertia& a, const Inertia& b) {
    return Inertia2{f2{a.inv_jt, b.inv_jt}, f2{a.jdiff, b.jdiff}};
}

// sim:145-150
__device__ __forceinline__ f2 density(f2 h) {
    const f2 base = fma2(h, splat((float)(kLb / kTb)), splat(1.0f));
    return splat((float)kRho0) * exp2_2(splat((float)(-kRhoExp)) * log2_2(base));
}

struct Rot2 {
    f2 r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

// sim:189-199
__device__ __forceinline__ Rot2 rotmat(f2 q0, f2 q1, f2 q2, f2 q3) {
    const f2 n2 = fma2(q0, q0, fma2(q1, q1, fma2(q2, q2, q3 * q3)));
    const f2 s = splat(2.0f) * rcp2(n2);
    const f2 s1 = s * q1, s2 = s * q2, s3 = s * q3;
    const f2 w1 = s1 * q0, w2 = s2 * q0, w3 = s3 * q0;
    const f2 x1 = s1 * q1, x2 = s2 * q1, x3 = s3 * q1;
    const f2 y2 = s2 * q2, y3 = s3 * q2, z3 = s3 * q3;
    const f2 one = splat(1.0f);
    Rot2 R;
    R.r00 = one - (y2 + z3);
    R.r01 = x2 - w3;
    R.r02 = x3 + w2;
    R.r10 = x2 + w3;
    R.r11 = one - (x1 + z3);
    R.r12 = y3 - w1;
    R.r20 = x3 - w2;
    R.r21 = y3 + w1;
    R.r22 = one - (x1 + y2);
    return R;
}

// sim:106-143, force / torque half (see r6::rhs_dyn)
__device__ __forceinline__ void rhs_dyn(const f2 (&y)[kDyn], f2 im, const Control2& c, const Inertia2& J, f2 (&dv)[3], f2 (&dw)[2]) {
    const f2 vx = y[3], vy = y[4], vz = y[5];
    const Rot2 R = rotmat(y[6], y[7], y[8], y[9]);
    const f2 speed = sqrt2(fma2(vx, vx, fma2(vy, vy, vz * vz)));
    const f2 ca = splat(-kAeroK) * density(y[0]) * speed;
    const f2 aby = ca * fma2(R.r01, vx, fma2(R.r11, vy, R.r21 * vz));
    const f2 abz = ca * fma2(R.r02, vx, fma2(R.r12, vy, R.r22 * vz));
    const f2 fx = fma2(R.r00, c.tbx, fma2(R.r01, c.tby, fma2(R.r02, c.tbz, ca * vx)));
    const f2 fy = fma2(R.r10, c.tbx, fma2(R.r11, c.tby, fma2(R.r12, c.tbz, ca * vy)));
    const f2 fz = fma2(R.r20, c.tbx, fma2(R.r21, c.tby, fma2(R.r22, c.tbz, ca * vz)));
    dv[0] = fma2(fx, im, splat(-(float)kG0));
    dv[1] = fy * im;
    dv[2] = fz * im;
    const f2 tauy = fma2(splat(-(float)kArmThrust), c.tbz, splat(-(float)kArmCp) * abz);
    const f2 tauz = fma2(splat((float)kArmThrust), c.tby, splat((float)kArmCp) * aby);
    const f2 g = J.jdiff * y[10];
    dw[0] = (tauy - g * y[12]) * J.inv_jt;
    dw[1] = (tauz + g * y[11]) * J.inv_jt;
}

// sim:133-136, :221-229
__device__ __forceinline__ void rhs_quat(const f2 (&y)[kDyn], f2 (&dq)[4]) {
    const f2 q0 = y[6], q1 = y[7], q2 = y[8], q3 = y[9];
    const f2 wx = y[10], wy = y[11], wz = y[12];
    dq[0] = splat(-0.5f) * fma2(wx, q1, fma2(wy, q2, wz * q3));
    dq[1] = splat(0.5f) * fma2(wx, q0, fma2(wz, q2, -wy * q3));
    dq[2] = splat(0.5f) * fma2(wy, q0, fma2(-wz, q1, wx * q3));
    dq[3] = splat(0.5f) * fma2(wz, q0, fma2(wy, q1, -wx * q2));
}

__device__ __forceinline__ void rhs13(const f2 (&y)[kDyn], f2 im, const Control2& c, const Inertia2& J, f2 (&dy)[kDyn]) {
    f2 dv[3], dw[2], dq[4];
    rhs_dyn(y, im, c, J, dv, dw);
    rhs_quat(y, dq);
    dy[0] = y[3];
    dy[1] = y[4];
    dy[2] = y[5];
    dy[3] = dv[0];
    dy[4] = dv[1];
    dy[5] = dv[2];
    dy[6] = dq[0];
    dy[7] = dq[1];
    dy[8] = dq[2];
    dy[9] = dq[3];
    dy[10] = splat(0.0f);
    dy[11] = dw[0];
    dy[12] = dw[1];
}

template <int INTEG>
__device__ __forceinline__ void step_map(const f2 (&y0)[kDyn], f2 mass, const Control2& c, const Inertia2& J, float h, f2 (&y1)[kDyn]) {
    f2 k[kDyn];
    if constexpr (INTEG == R6_EULER) {
        rhs13(y0, rcp2(mass), c, J, k);
#pragma unroll
        for (int i = 0; i < kDyn; ++i) y1[i] = fma2(splat(h), k[i], y0[i]);
    } else {
        const float hh = 0.5f * h, h6 = h * (1.0f / 6.0f);
        f2 yt[kDyn];
#pragma unroll
        for (int i = 0; i < kDyn; ++i) {
            yt[i] = y0[i];
            y1[i] = y0[i];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float tm = (s == 0) ? 0.0f : ((s == 3) ? h : hh);
            const float w = (s == 0 || s == 3) ? h6 : 2.0f * h6;
            const float an = (s == 2) ? h : hh;
            rhs13(yt, rcp2(fma2(-c.mdot, splat(tm), mass)), c, J, k);
#pragma unroll
            for (int i = 0; i < kDyn; ++i) {
                if (i == 10) continue;
                y1[i] = fma2(splat(w), k[i], y1[i]);
                yt[i] = fma2(splat(an), k[i], y0[i]);
            }
        }
    }
}

// One env step for the two envs of a lane (r6::integrate<INTEG, true>): fixed sub-steps, ground event by the
// Hermite dense output of the crossing sub-step.  status[h] = 1 when half h hit the ground; a half that has
// stopped keeps its event state (its further sub-steps are computed and discarded).
template <int INTEG>
__device__ __forceinline__ void integrate(f2 (&y)[kDyn], f2& mass, const Control2& c, const Inertia2& J, float dt, int nsub,
                                          int (&status)[2]) {
    const float h = dt / (float)nsub;
    status[0] = status[1] = 0;
#pragma unroll 1
    for (int s = 0; s < nsub; ++s) {
        f2 y1[kDyn];
        f2 hs = splat(h);
        step_map<INTEG>(y, mass, c, J, h, y1);
        const bool cx = status[0] == 0 && crossing(y[0].x, y1[0].x);
        const bool cy = status[1] == 0 && crossing(y[0].y, y1[0].y);
        if (cx || cy) {
            f2 th = splat(1.0f);
            if constexpr (INTEG == R6_RK4) {
                f2 a0[3], a1[3], w0[2], w1[2];
                rhs_dyn(y1, rcp2(fma2(-c.mdot, splat(h), mass)), c, J, a1, w1);
                rhs_dyn(y, rcp2(mass), c, J, a0, w0);
                f2 d0v[3], c2v[3], c3v[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    d0v[i] = splat(h) * a0[i];
                    const f2 D = (y1[3 + i] - y[3 + i]) - d0v[i];
                    const f2 E = fma2(splat(h), a1[i], -d0v[i]);
                    c2v[i] = fma2(splat(3.0f), D, -E);
                    c3v[i] = fma2(splat(-2.0f), D, E);
                }
                if (cx) {
                    const float cc[5] = {y[0].x, h * y[3].x, h * (0.5f * d0v[0].x), h * (c2v[0].x * (1.0f / 3.0f)), h * (0.25f * c3v[0].x)};
                    th.x = poly_root01(cc, y[0].x, y1[0].x);
                }
                if (cy) {
                    const float cc[5] = {y[0].y, h * y[3].y, h * (0.5f * d0v[0].y), h * (c2v[0].y * (1.0f / 3.0f)), h * (0.25f * c3v[0].y)};
                    th.y = poly_root01(cc, y[0].y, y1[0].y);
                }
                const f2 hth = splat(h) * th;
                f2 dq0[4], dq1[4], ye[kDyn];
                rhs_quat(y, dq0);
                rhs_quat(y1, dq1);
                auto hermite = [&](f2 ya, f2 yb, f2 fa, f2 fb) {
                    const f2 e0 = splat(h) * fa;
                    const f2 D = (yb - ya) - e0;
                    const f2 E = fma2(splat(h), fb, -e0);
                    return fma2(th, fma2(th, fma2(th, fma2(splat(-2.0f), D, E), fma2(splat(3.0f), D, -E)), e0), ya);
                };
#pragma unroll
                for (int i = 0; i < 4; ++i) ye[6 + i] = hermite(y[6 + i], y1[6 + i], dq0[i], dq1[i]);
                ye[10] = y1[10];
                ye[11] = hermite(y[11], y1[11], w0[0], w1[0]);
                ye[12] = hermite(y[12], y1[12], w0[1], w1[1]);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const f2 q4 = fma2(th, splat(0.25f) * c3v[i], c2v[i] * splat(1.0f / 3.0f));
                    ye[i] = fma2(hth, fma2(th, fma2(th, q4, splat(0.5f) * d0v[i]), y[3 + i]), y[i]);
                    ye[3 + i] = fma2(th, fma2(th, fma2(th, c3v[i], c2v[i]), d0v[i]), y[3 + i]);
                }
                // Newton correction in time on the position (per half, only where it applies)
                auto newton = [&](float x, float vx) {
                    const float d = -x / vx;
                    return (vx != 0.0f && fabsf(d) <= h) ? d : 0.0f;
                };
                const f2 dl = f2{newton(ye[0].x, ye[3].x), newton(ye[0].y, ye[3].y)};
                ye[0] = fma2(ye[3], dl, ye[0]);
                ye[1] = fma2(ye[4], dl, ye[1]);
                ye[2] = fma2(ye[5], dl, ye[2]);
#pragma unroll
                for (int i = 0; i < kDyn; ++i) y1[i] = f2{cx ? ye[i].x : y1[i].x, cy ? ye[i].y : y1[i].y};
                hs = f2{cx ? hth.x : h, cy ? hth.y : h};
            } else {
                if (cx) th.x = (y[0].x == 0.0f) ? 0.0f : y[0].x / (y[0].x - y1[0].x);
                if (cy) th.y = (y[0].y == 0.0f) ? 0.0f : y[0].y / (y[0].y - y1[0].y);
#pragma unroll
                for (int i = 0; i < kDyn; ++i) {
                    const f2 yi = fma2(th, y1[i] - y[i], y[i]);
                    y1[i] = f2{cx ? yi.x : y1[i].x, cy ? yi.y : y1[i].y};
                }
                hs = f2{cx ? h * th.x : h, cy ? h * th.y : h};
            }
        }
        // a half that already stopped at its event keeps that state
        const bool keep_x = status[0] != 0, keep_y = status[1] != 0;
#pragma unroll
        for (int i = 0; i < kDyn; ++i) y[i] = f2{keep_x ? y[i].x : y1[i].x, keep_y ? y[i].y : y1[i].y};
        mass = f2{keep_x ? mass.x : fmaf(-c.mdot.x, hs.x, mass.x), keep_y ? mass.y : fmaf(-c.mdot.y, hs.y, mass.y)};
        if (cx) status[0] = 1;
        if (cy) status[1] = 1;
    }
    // sim:97, :153-154
    const f2 n2 = fma2(y[6], y[6], fma2(y[7], y[7], fma2(y[8], y[8], y[9] * y[9])));
    const f2 inv = rsq2(n2);
    y[6] *= inv;
    y[7] *= inv;
    y[8] *= inv;
    y[9] *= inv;
}

}  // namespace pk
}  // namespace r6
