// r6_device.hpp -- per-environment device math of the Rocket6DOF hot path (gfx950, fp32).
//
// One lane owns one environment; nothing here communicates across lanes.  Citations are into the
// reference repo (my_environment/utils/simulator.py = "sim", my_environment/envs/rocket_env.py =
// "env"); the formulas are re-derived for fp32 / CDNA4 (gimbal trig hoisted out of the RK stages,
// mass integrated analytically, aero force applied in the inertial frame, diagonal inertia
// expanded), not transliterated.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/r6dof.h"

namespace r6 {

// ---- constants hard-coded in the reference (sim:39-67) ------------------------------------------
constexpr double kG0 = 9.81;                                   // sim:39
constexpr double kLength = 40.0;                               // sim:43
constexpr double kRadius = 3.66 / 2.0;                         // sim:44
constexpr double kIsp = 360.0;                                 // sim:51
constexpr double kCa = 0.82;                                   // sim:54-56
constexpr double kPi = 3.14159265358979323846;
constexpr double kSref = kPi * kRadius * kRadius;              // sim:57
constexpr double kRho0 = 1.225;                                // sim:58
constexpr double kTb = 288.15;                                 // sim:59
constexpr double kLb = -0.0065;                                // sim:60
constexpr double kMair = 0.0289644;                            // sim:62
constexpr double kRstar = 8.3144598;                           // sim:63
constexpr double kArmThrust = -15.0;                           // sim:66  r_T_B = [-15,0,0]
constexpr double kArmCp = 5.0;                                 // sim:67  r_cp_B = [5,0,0]
constexpr double kRhoExp = 1.0 + kG0 * kMair / kRstar / kLb;   // sim:146-148  (= -4.2558)
constexpr double kGShaping = 9.81;                             // env:548

constexpr float kAeroK = (float)(0.5 * kSref * kCa);           // -> A = -kAeroK * rho * |v| * v
constexpr float kMdotPerN = (float)(1.0 / (kG0 * kIsp));       // sim:140-141
constexpr float kJxPerM = (float)(0.5 * kRadius * kRadius);                              // sim:46
constexpr float kJtPerM = (float)((kLength * kLength + 3.0 * kRadius * kRadius) / 12.0); // sim:47-48

// ---- fast hardware transcendentals (1 ulp class; v_rcp/v_rsq/v_sqrt/v_exp/v_log) ----------------
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

// ---- per-step constants of one env ---------------------------------------------------------------
struct Control {
    float tbx, tby, tbz;  // thrust in body frame (sim:167-175, :202-214: first gimbal column * T)
    float thrust;         // N
    float mdot;           // kg/s  (sim:140-141)
};

struct Inertia {          // sim:42-50, frozen at the episode's initial mass m0
    float inv_jt;         // 1 / Jt
    float jdiff;          // Jx - Jt
};

// sin/cos of a gimbal angle.  |x| <= pi/4 (normalised action within +-2.25) takes minimax
// polynomials (Cephes sinf/cosf kernels, < 1 ulp of float32 on that range); anything larger falls
// back to the full-range library routine.  The branch is wave-uniform.
__device__ __forceinline__ void sincos_gimbal(float x, float* s, float* c) {
    if (__builtin_expect(__any(fabsf(x) > 0.78539816f), 0)) {
        sincosf(x, s, c);
        return;
    }
    const float z = x * x;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    *s = fmaf(x * z, ps, x);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    *c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
}

// env:509-521 -- no clipping of the normalised action
__device__ __forceinline__ Control make_control_normalised(const R6Params& p, float a0, float a1, float a2) {
    const float dy = a0 * p.max_gimbal;
    const float dz = a1 * p.max_gimbal;
    const float thr = (a2 + 1.0f) * 0.5f * p.max_thrust;
    float sy, cy, sz, cz;
    sincos_gimbal(dy, &sy, &cy);
    sincos_gimbal(dz, &sz, &cz);
    Control c;
    c.tbx = thr * (cy * cz);
    c.tby = thr * (sy * cz);
    c.tbz = thr * sz;
    c.thrust = thr;
    c.mdot = thr * kMdotPerN;
    return c;
}

__device__ __forceinline__ Control make_control_physical(float dy, float dz, float thr) {
    float sy, cy, sz, cz;
    sincosf(dy, &sy, &cy);
    sincosf(dz, &sz, &cz);
    Control c;
    c.tbx = thr * (cy * cz);
    c.tby = thr * (sy * cz);
    c.tbz = thr * sz;
    c.thrust = thr;
    c.mdot = thr * kMdotPerN;
    return c;
}

__device__ __forceinline__ Inertia make_inertia(float m0) {
    Inertia j;
    const float jt = m0 * kJtPerM;
    j.inv_jt = frcp(jt);
    j.jdiff = m0 * kJxPerM - jt;
    return j;
}

// sim:145-150 -- rho0 * (Tb / (Tb + Lb h))^e  ==  rho0 * exp2(-e * log2(1 + (Lb/Tb) h))
__device__ __forceinline__ float density(float h) {
    const float base = fmaf(h, (float)(kLb / kTb), 1.0f);
    return (float)kRho0 * fexp2((float)(-kRhoExp) * flog2(base));
}

// Dynamic part of the state: r(0..2) v(3..5) q(6..9) w(10..12).  Mass is linear in time (constant
// mdot within a step, sim:140-141), so it is carried outside the RK stages.
constexpr int kDyn = 13;

// Rotation matrix body->inertial of q/|q| (sim:189-199; scipy normalises), written for a
// non-unit quaternion: R = I + (2/|q|^2) * (...).
struct Rot {
    float r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

template <bool UNIT = false>
__device__ __forceinline__ Rot rotmat(float q0, float q1, float q2, float q3) {
    // UNIT: the caller guarantees |q| = 1 to fp32 round-off (sim:97 renormalises after every step), s = 2 / |q|^2 = 2
    float s = 2.0f;
    if constexpr (!UNIT) s = 2.0f * frcp(fmaf(q0, q0, fmaf(q1, q1, fmaf(q2, q2, q3 * q3))));
    const float s1 = s * q1, s2 = s * q2, s3 = s * q3;
    const float w1 = s1 * q0, w2 = s2 * q0, w3 = s3 * q0;
    const float x1 = s1 * q1, x2 = s2 * q1, x3 = s3 * q1;
    const float y2 = s2 * q2, y3 = s3 * q2, z3 = s3 * q3;
    Rot R;
    R.r00 = 1.0f - (y2 + z3);
    R.r01 = x2 - w3;
    R.r02 = x3 + w2;
    R.r10 = x2 + w3;
    R.r11 = 1.0f - (x1 + z3);
    R.r12 = y3 - w1;
    R.r20 = x3 - w2;
    R.r21 = y3 + w1;
    R.r22 = 1.0f - (x1 + y2);
    return R;
}

// sim:106-143 -- the force / torque half of the RHS: dv[3] = d(v)/dt, dw[2] = d(omega_y, omega_z)/dt
// (d(omega_x)/dt == 0, see below); `im` = 1 / mass at this time.
//
// Only two rotated vectors are needed per evaluation -- R T_B (thrust in the inertial frame) and the body-frame y, z
// components of the velocity (aero torque arm is along body x) -- so the matrix of sim:189-199 is not formed: with
// s = 2 / |q|^2 (scipy normalises the quaternion, sim:198),  R u = u + s q_v x (q_v x u + q0 u)  and
// R^T u = u - s q_v x (q0 u - q_v x u)  (6 multiply-adds less per evaluation than 9 matrix entries + 5 dot products).
__device__ __forceinline__ void rhs_dyn(const float (&y)[kDyn], float im, const Control& c, const Inertia& J,
                                        float (&dv)[3], float (&dw)[2]) {
    const float vx = y[3], vy = y[4], vz = y[5];
    const float speed = fsqrt(fmaf(vx, vx, fmaf(vy, vy, vz * vz)));
    // aero (sim:216-219): A_B = -k rho |v| R^T v ; rotated back: A_I = -k rho |v| v
    const float ca = -kAeroK * density(y[0]) * speed;
    const float q0 = y[6], q1 = y[7], q2 = y[8], q3 = y[9];
    const float n2 = fmaf(q0, q0, fmaf(q1, q1, fmaf(q2, q2, q3 * q3)));
    const float s = 2.0f * frcp(n2);
    // thrust: e = q_v x T_B + q0 T_B,  R T_B = T_B + s q_v x e
    const float ex = fmaf(q2, c.tbz, fmaf(-q3, c.tby, q0 * c.tbx));
    const float ey = fmaf(q3, c.tbx, fmaf(-q1, c.tbz, q0 * c.tby));
    const float ez = fmaf(q1, c.tby, fmaf(-q2, c.tbx, q0 * c.tbz));
    const float tix = fmaf(s, fmaf(q2, ez, -q3 * ey), c.tbx);
    const float tiy = fmaf(s, fmaf(q3, ex, -q1 * ez), c.tby);
    const float tiz = fmaf(s, fmaf(q1, ey, -q2 * ex), c.tbz);
    // body-frame velocity, y and z components: d = q0 v - q_v x v,  R^T v = v - s q_v x d
    const float dx = fmaf(q0, vx, fmaf(q3, vy, -q2 * vz));
    const float dy = fmaf(q0, vy, fmaf(q1, vz, -q3 * vx));
    const float dz = fmaf(q0, vz, fmaf(q2, vx, -q1 * vy));
    const float vby = fmaf(-s, fmaf(q3, dx, -q1 * dz), vy);
    const float vbz = fmaf(-s, fmaf(q1, dy, -q2 * dx), vz);
    const float aby = ca * vby;   // body-y aero force
    const float abz = ca * vbz;   // body-z aero force
    // translational (sim:124-130, :156-165): F_I = R T_B + A_I
    dv[0] = fmaf(fmaf(ca, vx, tix), im, -(float)kG0);
    dv[1] = fmaf(ca, vy, tiy) * im;
    dv[2] = fmaf(ca, vz, tiz) * im;

    // rotational (sim:137, :232-244): tau = r_T x T_B + r_cp x A_B with r = [rx,0,0]
    //   -> tau = (0, -rT*tbz - rcp*abz, rT*tby + rcp*aby);  J = diag(Jx,Jt,Jt):
    //   w x Jw = (0, (Jx-Jt) wx wz, (Jt-Jx) wx wy)  => wx is constant
    const float tauy = fmaf(-(float)kArmThrust, c.tbz, -(float)kArmCp * abz);
    const float tauz = fmaf((float)kArmThrust, c.tby, (float)kArmCp * aby);
    const float g = J.jdiff * y[10];
    dw[0] = (tauy - g * y[12]) * J.inv_jt;
    dw[1] = (tauz + g * y[11]) * J.inv_jt;
}

// quaternion kinematics (sim:133-136, :221-229): qdot = 1/2 Omega(w) q
__device__ __forceinline__ void rhs_quat(const float (&y)[kDyn], float (&dq)[4]) {
    const float q0 = y[6], q1 = y[7], q2 = y[8], q3 = y[9];
    const float wx = y[10], wy = y[11], wz = y[12];
    dq[0] = -0.5f * fmaf(wx, q1, fmaf(wy, q2, wz * q3));
    dq[1] = 0.5f * fmaf(wx, q0, fmaf(wz, q2, -wy * q3));
    dq[2] = 0.5f * fmaf(wy, q0, fmaf(-wz, q1, wx * q3));
    dq[3] = 0.5f * fmaf(wz, q0, fmaf(wy, q1, -wx * q2));
}

// The same without the factor 1/2: (2 qdot)[0] is the NEGATIVE of d2[0] (the caller folds -1/2 and 1/2 into its step weights --
// scaling by a power of two is exact, so  w * (0.5 * X) == (0.5 * w) * X  bit for bit, and four multiplies per RK4 stage go away).
__device__ __forceinline__ void rhs_quat_x2(const float (&y)[kDyn], float (&d2)[4]) {
    const float q0 = y[6], q1 = y[7], q2 = y[8], q3 = y[9];
    const float wx = y[10], wy = y[11], wz = y[12];
    d2[0] = fmaf(wx, q1, fmaf(wy, q2, wz * q3));            // = -2 dq0
    d2[1] = fmaf(wx, q0, fmaf(wz, q2, -wy * q3));           // =  2 dq1
    d2[2] = fmaf(wy, q0, fmaf(-wz, q1, wx * q3));           // =  2 dq2
    d2[3] = fmaf(wz, q0, fmaf(wy, q1, -wx * q2));           // =  2 dq3
}

// sim:106-143 -- derivative of the 13 dynamic components; `im` = 1 / mass at this time.
__device__ __forceinline__ void rhs13(const float (&y)[kDyn], float im, const Control& c, const Inertia& J,
                                      float (&dy)[kDyn]) {
    float dv[3], dw[2], dq[4];
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
    dy[10] = 0.0f;
    dy[11] = dw[0];
    dy[12] = dw[1];
}

#ifndef R6_RK4_ROLLED
#define R6_RK4_ROLLED 0
#endif


#ifndef R6_VGPR_STEP
#define R6_VGPR_STEP 1
#endif
// One application of the Euler / RK4 map.
template <int INTEG>
__device__ __forceinline__ void step_map(const float (&y0)[kDyn], float mass, const Control& c, const Inertia& J,
                                         float h_in, float (&y1)[kDyn]) {
    // The step and the weights derived from it are wave-uniform, and left alone they sit in scalar registers -- but a vector
    // instruction with a scalar-register source issues in ~1.8 ns per SIMD on gfx950, one whose sources are all vector registers
    // (or inline constants / literals) in ~1.2 ns (scripts/valu_operand_bench.hip), and the 96 stage FMAs of an RK4 step all
    // take a weight.  R6_VGPR_STEP (default 1) keeps the step in a vector register, so the weights are vector values too.
    float h = h_in;
#if R6_VGPR_STEP
    asm volatile("" : "+v"(h));
#endif
    float kend[kDyn];
    if constexpr (INTEG == R6_EULER) {
        rhs13(y0, frcp(mass), c, J, kend);
#pragma unroll
        for (int i = 0; i < kDyn; ++i) y1[i] = fmaf(h, kend[i], y0[i]);
    } else {
        // Classical RK4, one loop over the four stages (unrolled by default: rolled, ~20 % of the VALU slots went
        // into register copies at the back edge; R6_RK4_ROLLED=1 keeps one copy of the RHS for experiments):
        // y1 accumulates y0 + h/6 (k1 + 2k2 + 2k3 + k4), yt is the stage argument.  Mass is linear in time:
        // stage masses m, m-h/2 mdot (x2), m-h mdot.
        const float hh = 0.5f * h;
        const float h6 = h * (1.0f / 6.0f);
        float yt[kDyn];
#pragma unroll
        for (int i = 0; i < kDyn; ++i) {
            yt[i] = y0[i];
            y1[i] = y0[i];
        }
#if R6_RK4_ROLLED
#pragma unroll 1
#else
#pragma unroll
#endif
        for (int s = 0; s < 4; ++s) {
            const float tm = (s == 0) ? 0.0f : ((s == 3) ? h : hh);      // stage time offset
            const float w = (s == 0 || s == 3) ? h6 : 2.0f * h6;         // quadrature weight * h
            const float an = (s == 2) ? h : hh;                          // next stage's step
            {   // rhs13 with the quaternion rows left at twice their value (rhs_quat_x2): the 1/2 sits in wq / aq below
                float dv[3], dw[2], d2[4];
                rhs_dyn(yt, frcp(fmaf(-c.mdot, tm, mass)), c, J, dv, dw);
                rhs_quat_x2(yt, d2);
                kend[0] = yt[3];
                kend[1] = yt[4];
                kend[2] = yt[5];
                kend[3] = dv[0];
                kend[4] = dv[1];
                kend[5] = dv[2];
                kend[6] = d2[0];
                kend[7] = d2[1];
                kend[8] = d2[2];
                kend[9] = d2[3];
                kend[10] = 0.0f;
                kend[11] = dw[0];
                kend[12] = dw[1];
            }
            const float wq = 0.5f * w, aq = 0.5f * an;
#pragma unroll
            for (int i = 0; i < kDyn; ++i) {
                if (i == 10) continue;   // d(omega_x)/dt == 0 exactly (rhs13): omega_x stays y0[10]
                const float wi = (i == 6) ? -wq : ((i >= 7 && i <= 9) ? wq : w);
                const float ai = (i == 6) ? -aq : ((i >= 7 && i <= 9) ? aq : an);
                y1[i] = fmaf(wi, kend[i], y1[i]);
                yt[i] = fmaf(ai, kend[i], y0[i]);
            }
        }
    }
}

// scipy find_active_events with direction == 0 (through sim:77-88)
__device__ __forceinline__ bool crossing(float x_old, float x_new) {
    return (x_old <= 0.0f && x_new >= 0.0f) || (x_old >= 0.0f && x_new <= 0.0f);
}

// theta in [0,1] with P(theta) = 0, P = sum c[k] t^k, P(0) = x0, P(1) = x1 of opposite (or zero)
// sign.  Safeguarded Newton from the secant point, at most 8 iterations -- the same procedure, step
// for step, as oracle/oracle_np.py poly_root01.
template <int N>
__device__ __forceinline__ float poly_root01(const float (&c)[N], float x0, float x1) {
    if (x0 == 0.0f) return 0.0f;
    if (x1 == 0.0f) return 1.0f;
    float lo = 0.0f, hi = 1.0f;
    float th = x0 / (x0 - x1);
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        float f = c[N - 1], df = 0.0f;
#pragma unroll
        for (int k = N - 2; k >= 0; --k) {
            df = fmaf(df, th, f);
            f = fmaf(f, th, c[k]);
        }
        if (f == 0.0f) break;
        if ((f > 0.0f) == (x0 > 0.0f)) lo = th; else hi = th;
        float nt = (df != 0.0f) ? th - f / df : 0.5f * (lo + hi);
        if (!(nt >= lo && nt <= hi)) nt = 0.5f * (lo + hi);
        if (nt == th) break;  // converged to working precision
        th = nt;
    }
    return th;
}

// One env step of the simulator (sim:69-104) with fixed sub-steps.
//
// Ground event (oracle_np.event_step_rk4) -- the reference returns its integrator's dense output at the
// event time (sim:77-90 / scipy ivp.py); the fixed-step twin does the same with a Hermite dense output of
// the RK4 step through its end points with the exact end slopes f0 = f(y0), f1 = f(y1) (two RHS
// evaluations; a second, shortened RK4 step would cost four and is no closer to the reference):
//   v, q, omega:  Y(th) = y0 + th (d0 + th (c2 + th c3)),  d0 = h f0, d1 = h f1,
//                 D = (y1 - y0) - d0, E = d1 - d0, c2 = 3D - E, c3 = E - 2D      (cancellation-free form)
//   r:            R(th) = r0 + h th (v0 + th (d0/2 + th (c2/3 + th c3/4)))   (integral of v's interpolant)
//   1. theta = root in [0,1] of the x component of R (a quartic);  2. state = (R, Y)(theta);
//   3. one Newton correction in time on the position only: delta = -x/vx, r += v delta.
// Euler: the map is linear in h, so theta = x0/(x0-x1) and the state is y0 + theta (y1 - y0).
//
// EVENTS = true : full semantics, returns status (1 = ground event handled).
// EVENTS = false: the crossing sub-step stops the integration with status 1 WITHOUT producing the event
//                 state (timing experiments only).
template <int INTEG, bool EVENTS, bool ONE = false>
__device__ __forceinline__ int integrate(float (&y)[kDyn], float& mass, const Control& c, const Inertia& J,
                                         float dt, int nsub) {
    // ONE: the caller guarantees nsub == 1 (the reference's own configuration: one env step = one integrator call).
    // The general form carries the state across a loop back edge, which costs ~25 register copies per sub-step
    // (y0 and y1 are both live until the end of the RK4 map); the single-step form is straight-line code.
    const float h = ONE ? dt : dt / (float)nsub;
    int status = 0;
#pragma unroll 1
    for (int s = 0; s < (ONE ? 1 : nsub); ++s) {
        if (!ONE && status != 0) continue;
        float y1[kDyn];
        float hs = h;
        step_map<INTEG>(y, mass, c, J, h, y1);
        if (crossing(y[0], y1[0])) {
            status = 1;
            if constexpr (EVENTS) {
                if constexpr (INTEG == R6_RK4) {
                    // end slopes of v and (omega_y, omega_z); the quaternion's are formed where they are used
                    float a0[3], a1[3], w0[2], w1[2];
                    rhs_dyn(y1, frcp(fmaf(-c.mdot, h, mass)), c, J, a1, w1);
                    rhs_dyn(y, frcp(mass), c, J, a0, w0);
                    float d0v[3], c2v[3], c3v[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {      // velocity coefficients first: the x quartic needs vx's
                        d0v[i] = h * a0[i];
                        const float D = (y1[3 + i] - y[3 + i]) - d0v[i];
                        const float E = fmaf(h, a1[i], -d0v[i]);
                        c2v[i] = fmaf(3.0f, D, -E);
                        c3v[i] = fmaf(-2.0f, D, E);
                    }
                    const float cx[5] = {y[0], h * y[3], h * (0.5f * d0v[0]), h * (c2v[0] * (1.0f / 3.0f)),
                                         h * (0.25f * c3v[0])};
                    const float th = poly_root01(cx, y[0], y1[0]);
                    const float hth = h * th;
                    float dq0[4], dq1[4];
                    rhs_quat(y, dq0);
                    rhs_quat(y1, dq1);
                    auto hermite = [&](float ya, float yb, float fa, float fb) {
                        const float e0 = h * fa;
                        const float D = (yb - ya) - e0;
                        const float E = fmaf(h, fb, -e0);
                        return fmaf(th, fmaf(th, fmaf(th, fmaf(-2.0f, D, E), fmaf(3.0f, D, -E)), e0), ya);
                    };
#pragma unroll
                    for (int i = 0; i < 4; ++i) y1[6 + i] = hermite(y[6 + i], y1[6 + i], dq0[i], dq1[i]);
                    y1[11] = hermite(y[11], y1[11], w0[0], w1[0]);
                    y1[12] = hermite(y[12], y1[12], w0[1], w1[1]);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const float q4 = fmaf(th, 0.25f * c3v[i], c2v[i] * (1.0f / 3.0f));
                        y1[i] = fmaf(hth, fmaf(th, fmaf(th, q4, 0.5f * d0v[i]), y[3 + i]), y[i]);
                        y1[3 + i] = fmaf(th, fmaf(th, fmaf(th, c3v[i], c2v[i]), d0v[i]), y[3 + i]);
                    }
                    const float delta = -y1[0] / y1[3];
                    if (y1[3] != 0.0f && fabsf(delta) <= h) {
                        y1[0] = fmaf(y1[3], delta, y1[0]);
                        y1[1] = fmaf(y1[4], delta, y1[1]);
                        y1[2] = fmaf(y1[5], delta, y1[2]);
                    }
                    hs = hth;
                } else {
                    const float th = (y[0] == 0.0f) ? 0.0f : y[0] / (y[0] - y1[0]);
#pragma unroll
                    for (int i = 0; i < kDyn; ++i) y1[i] = fmaf(th, y1[i] - y[i], y[i]);
                    hs = th * h;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < kDyn; ++i) y[i] = y1[i];
        mass = fmaf(-c.mdot, hs, mass);
    }
    // sim:97, :153-154 -- quaternion renormalised after the step
    const float n2 = fmaf(y[6], y[6], fmaf(y[7], y[7], fmaf(y[8], y[8], y[9] * y[9])));
    const float inv = frsq(n2);
    y[6] *= inv;
    y[7] *= inv;
    y[8] *= inv;
    y[9] *= inv;
    return status;
}

// ---- adaptive Dormand-Prince RK45 with scipy's step control (sim:83-88 = scipy.integrate.solve_ivp
//      defaults: RK45, rtol 1e-3, atol 1e-6, first_step None, terminal event on x) ----------------------
// The algorithmic twin of the reference's integrator, per lane: initial step by Hairer's rule
// (scipy _ivp/common.py select_initial_step), error norm = RMS of err / (atol + rtol max(|y|,|y_new|))
// over all 14 components, accept if < 1, factor = min(10, 0.9 err^-1/5) (min(1, .) after a rejection),
// else shrink by max(0.2, 0.9 err^-1/5) (scipy _ivp/rk.py _step_impl); the terminal event is located
// on the method's quartic dense output (rk.py RkDenseOutput, P matrix) and the state taken from it.
// Tableau: Dormand & Prince 1980; dense output: Shampine 1986 (values as tabulated in scipy 1.15.3).
// Mass has a constant derivative: it is advanced analytically (its error estimate is exactly 0,
// sum(E) = 0) but is counted in the norms like scipy counts it.
namespace dp45 {
constexpr int S = 7;  // 6 stages + f(y_new)
constexpr float C[6] = {0.0f, 0.2f, 0.3f, 0.8f, (float)(8.0 / 9.0), 1.0f};
constexpr float A[6][5] = {
    {0, 0, 0, 0, 0},
    {(float)0.2, 0, 0, 0, 0},
    {(float)0.075, (float)0.225, 0, 0, 0},
    {(float)0.97777777777777775, (float)-3.7333333333333334, (float)3.5555555555555554, 0, 0},
    {(float)2.9525986892242035, (float)-11.595793324188385, (float)9.8228928516994358, (float)-0.29080932784636487, 0},
    {(float)2.8462752525252526, (float)-10.757575757575758, (float)8.9064227177434727, (float)0.27840909090909088,
     (float)-0.2735313036020583}};
constexpr float B[6] = {(float)0.091145833333333329, 0.0f, (float)0.44923629829290207, (float)0.65104166666666663,
                        (float)-0.322376179245283, (float)0.13095238095238096};
constexpr float E[7] = {(float)-0.0012326388888888888, 0.0f, (float)0.0042527702905061394, (float)-0.036979166666666667,
                        (float)0.05086379716981132, (float)-0.041904761904761903, 0.025f};
constexpr float P[7][4] = {{1.0f, (float)-2.8535800653862835, (float)3.0717434641059005, (float)-1.1270175653862835},
                           {0, 0, 0, 0},
                           {0, (float)4.0231333792303046, (float)-6.2493215652889997, (float)2.675424484351598},
                           {0, (float)-3.7324019615885042, (float)10.068970589843675, (float)-5.6855269615885042},
                           {0, (float)2.5548038301849423, (float)-6.3991123773510168, (float)3.5219323679207912},
                           {0, (float)-1.3744241142186024, (float)3.2726577522467291, (float)-1.7672812570757455},
                           {0, (float)1.3824689317781436, (float)-3.7649378635562871, (float)2.3824689317781438}};
constexpr float kRtol = 1e-3f, kAtol = 1e-6f;
constexpr float kSafety = 0.9f, kMinFactor = 0.2f, kMaxFactor = 10.0f;

// The stage vectors hold only the components whose derivative is not known without evaluating forces: v (3), q (4),
// omega_y, omega_z -- D = 9 of the 14.  The rest follows exactly: d(mass)/dt = -mdot is constant, d(omega_x)/dt == 0 (rhs_dyn),
// and d(r)/dt of stage j is that stage's velocity ARGUMENT, v + h sum_m A[j][m] Kv[m] -- so every place where the method takes
// a linear combination of the stages' position rows (stage arguments, y_new, error estimate, dense output) becomes
//     (sum of the weights) v + h sum_m (weights . A[., m]) Kv[m]
// with the coefficient products below, formed once at compile time in double precision.  (Round 3 kept all 7 x 13 stage values:
// 254-256 VGPRs, two waves per SIMD.)
constexpr int D = 9;
struct Derived {
    float AA[6][5];   // stage st, position: r + h (C[st] v + h sum_m AA[st][m] Kv[m])
    float BA[5];      // y_new, position:     r + h (v + h sum_m BA[m] Kv[m])                         (sum B = 1)
    float EA[6];      // error estimate, position rows: h sum_m EA[m] Kv[m]                            (sum E = 0); the f(y_new) row enters through B
    float PS[4];      // dense output, position rows: sum_j P[j][k]  (= 1, 0, 0, 0 up to round-off)
    float PA[6][4];   // dense output, position rows: h sum_m PA[m][k] Kv[m]
};
constexpr Derived make_derived() {
    constexpr double a[6][5] = {{0, 0, 0, 0, 0},
                                {1.0 / 5, 0, 0, 0, 0},
                                {3.0 / 40, 9.0 / 40, 0, 0, 0},
                                {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0},
                                {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0},
                                {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656}};
    constexpr double bw[6] = {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84};
    constexpr double e[7] = {-71.0 / 57600, 0, 71.0 / 16695, -71.0 / 1920, 17253.0 / 339200, -22.0 / 525, 1.0 / 40};
    constexpr double pm[7][4] = {{1, -8048581381.0 / 2820520608, 8663915743.0 / 2820520608, -12715105075.0 / 11282082432},
                                 {0, 0, 0, 0},
                                 {0, 131558114200.0 / 32700410799, -68118460800.0 / 10900136933, 87487479700.0 / 32700410799},
                                 {0, -1754552775.0 / 470086768, 14199869525.0 / 1410260304, -10690763975.0 / 1880347072},
                                 {0, 127303824393.0 / 49829197408, -318862633887.0 / 49829197408, 701980252875.0 / 199316789632},
                                 {0, -282668133.0 / 205662961, 2019193451.0 / 616988883, -1453857185.0 / 822651844},
                                 {0, 40617522.0 / 29380423, -110615467.0 / 29380423, 69997945.0 / 29380423}};
    Derived t{};
    for (int st = 0; st < 6; ++st)
        for (int m = 0; m < 5; ++m) {
            double acc = 0;
            for (int j = m + 1; j < st; ++j) acc += a[st][j] * a[j][m];
            t.AA[st][m] = (float)acc;
        }
    for (int m = 0; m < 5; ++m) {
        double acc = 0;
        for (int j = m + 1; j < 6; ++j) acc += bw[j] * a[j][m];
        t.BA[m] = (float)acc;
    }
    for (int m = 0; m < 6; ++m) {
        double acc = e[6] * bw[m];
        for (int j = m + 1; j < 6; ++j) acc += e[j] * a[j][m];
        t.EA[m] = (float)acc;
    }
    for (int k = 0; k < 4; ++k) {
        double acc = 0;
        for (int j = 0; j < 7; ++j) acc += pm[j][k];
        t.PS[k] = (float)acc;
    }
    for (int m = 0; m < 6; ++m)
        for (int k = 0; k < 4; ++k) {
            double acc = pm[6][k] * bw[m];
            for (int j = m + 1; j < 6; ++j) acc += pm[j][k] * a[j][m];
            t.PA[m][k] = (float)acc;
        }
    return t;
}
constexpr Derived T = make_derived();
// component of the 13-vector behind stage-vector slot d
__device__ __forceinline__ constexpr int comp(int d) { return d < 7 ? 3 + d : 4 + d; }   // 3,4,5 | 6,7,8,9 | 11,12
// acc + coef * k for a tableau coefficient: a zero coefficient (B[1], E[1], the first column of P below its first row, ...) is no
// instruction at all.  The calls sit in fully unrolled loops over constexpr tables, so the test folds at compile time; written as
// plain fmaf the compiler has to keep `fma(0, k, acc)` (0 * inf is NaN) -- 41 such instructions in the main path of an attempt and 80
// in its event path (round 5, found in the disassembly).  For finite k the result is the same number (scipy's dot products add the
// zero terms too); only the sign of an exact zero can differ.
__device__ __forceinline__ float fma_c(float coef, float k, float acc) { return coef == 0.0f ? acc : fmaf(coef, k, acc); }
// ... and a sum that starts with a PRODUCT: written `acc = 0; acc = fma(c, k, acc)` the first term is a v_fma_f32 with a zero addend
// (not a multiply: -0 + 0 is +0) whose coefficient sits in a scalar register (a VOP3 takes no literal on gfx9: 1.8 ns against the 1.2 ns
// of a v_mul_f32 with a literal) -- 97 of an attempt's 1409 vector instructions.  `first` folds at compile time like the zero test.
__device__ __forceinline__ void acc_c(float coef, float k, float& acc, bool& first) {
    if (coef == 0.0f) return;
    acc = first ? coef * k : fmaf(coef, k, acc);
    first = false;
}
}  // namespace dp45

// What one lane carries from attempt to attempt of the adaptive integrator (next to its state y and the control of the step):
// enough to stop after any attempt and resume later, in another wave (step_kernel's straggler pass).
struct Rk45Lane {
    float K0[dp45::D];   // f(y) of the step in progress (FSAL: f(y_new) of the step before)
    float t, h_abs;      // time reached within [0, dt]; step size the next attempt starts from
    // One word for the bookkeeping (the attempt loop runs at the register limit of four waves per SIMD: three counters more were
    // ten scratch loads per attempt): bits 0..1 status + 1 (1 integrating, 2 ground event, 0 step size underflow),
    // bits 2..8 rejected attempts of the step in progress (scipy's step_rejected <=> tries > 0), bits 9..31 attempts made so far.
    uint32_t ctl;
};
__device__ __forceinline__ int rk45_status(const Rk45Lane& L) { return (int)(L.ctl & 3u) - 1; }
__device__ __forceinline__ int rk45_tries(const Rk45Lane& L) { return (int)((L.ctl >> 2) & 127u); }
__device__ __forceinline__ int rk45_attempts(const Rk45Lane& L) { return (int)(L.ctl >> 9); }
__device__ __forceinline__ int rk45_nfev(const Rk45Lane& L) { return 2 + 6 * rk45_attempts(L); }   // scipy's sol.nfev: 2 for the initial step, 6 per attempt
// integrating, not at the end of the env step, and fewer than 4096 attempts behind it (scipy has no such bound; a lane that hits it
// leaves with status 0 short of dt, like round 4's guard of 4096 accepted steps -- neither is reached by a finite state)
__device__ __forceinline__ bool rk45_active(const Rk45Lane& L, float dt) { return (L.ctl & 3u) == 1u && L.t < dt && (L.ctl >> 9) < 4096u; }

// the force / torque / kinematics part of the RHS at state `ya` (only ya[0], ya[3..12] are read), into a stage vector
__device__ __forceinline__ void rk45_eval(const float (&ya)[kDyn], float im, const Control& c, const Inertia& J, float (&k)[dp45::D]) {
    float dv[3], dw[2], dq[4];
    rhs_dyn(ya, im, c, J, dv, dw);
    rhs_quat(ya, dq);
    k[0] = dv[0];
    k[1] = dv[1];
    k[2] = dv[2];
    k[3] = dq[0];
    k[4] = dq[1];
    k[5] = dq[2];
    k[6] = dq[3];
    k[7] = dw[0];
    k[8] = dw[1];
}

// f(y) and select_initial_step (scipy _ivp/common.py): two RHS evaluations.
__device__ __forceinline__ void rk45_begin(const float (&y)[kDyn], float mass, const Control& c, const Inertia& J, float dt, Rk45Lane& L) {
    using namespace dp45;
    const float mdot = c.mdot;
    auto sq = [](float v) { return v * v; };
    float Kn[D], ynew[kDyn];
    rk45_eval(y, frcp(mass), c, J, L.K0);
    {
        float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
        for (int i = 0; i < kDyn; ++i) {   // the 13 dynamic components: derivative = v for r, K0 for v / q / omega_yz, 0 for omega_x
            const float isc = frcp(fmaf(fabsf(y[i]), kRtol, kAtol));
            d0 += sq(y[i] * isc);
            const float f0 = (i < 3) ? y[3 + i] : ((i == 10) ? 0.0f : L.K0[i < 10 ? i - 3 : i - 4]);
            d1 += sq(f0 * isc);
        }
        const float iscm = frcp(fmaf(fabsf(mass), kRtol, kAtol));
        // (hardware square root / reciprocal / exp2 / log2, 1 ulp: the step-size control needs no more, and the IEEE-exact forms --
        //  sqrtf, /, powf -- were a fifth of the instructions of an attempt)
        d0 = fsqrt((d0 + sq(mass * iscm)) * (1.0f / 14.0f));
        d1 = fsqrt((d1 + sq(mdot * iscm)) * (1.0f / 14.0f));
        float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 * frcp(d1);
        h0 = fminf(h0, dt);
#pragma unroll
        for (int i = 0; i < 3; ++i) ynew[i] = fmaf(h0, y[3 + i], y[i]);
#pragma unroll
        for (int d = 0; d < D; ++d) ynew[comp(d)] = fmaf(h0, L.K0[d], y[comp(d)]);
        ynew[10] = y[10];
        rk45_eval(ynew, frcp(fmaf(-mdot, h0, mass)), c, J, Kn);
        float d2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 3; ++i) d2 += sq((ynew[3 + i] - y[3 + i]) * frcp(fmaf(fabsf(y[i]), kRtol, kAtol)));   // f1 - f0 of the position rows
#pragma unroll
        for (int d = 0; d < D; ++d) d2 += sq((Kn[d] - L.K0[d]) * frcp(fmaf(fabsf(y[comp(d)]), kRtol, kAtol)));
        d2 = fsqrt(d2 * (1.0f / 14.0f)) * frcp(h0);
        const float h1 = (d1 <= 1e-15f && d2 <= 1e-15f) ? fmaxf(1e-6f, h0 * 1e-3f) : fexp2(0.2f * flog2(0.01f * frcp(fmaxf(d1, d2))));
        L.h_abs = fminf(fminf(100.0f * h0, h1), dt);
    }
    L.t = 0.0f;
    L.ctl = 1u;
}

// ONE attempt of a step (rk.py _step_impl: rk_step, error norm, accept / reject, next step size) and, when it was accepted, the
// terminal-event check on it (ivp.py find_active_events / handle_events).  The caller guarantees rk45_active(L).  `mass0` is the
// mass at t = 0 (the mass is linear in time: it is advanced analytically).  The loop  while (rk45_active) rk45_attempt  is,
// operation for operation, scipy's nested loop (steps x tries) of one lane.
__device__ __forceinline__ void rk45_attempt(float (&y)[kDyn], float mass0, const Control& c, const Inertia& J, float dt, Rk45Lane& L) {
    using namespace dp45;
    const float mdot = c.mdot;
    auto sq = [](float v) { return v * v; };
    float K[6][D];    // stages 1..6 of the attempt; K[0] = L.K0
    float Kn[D];      // f(y_new)
    float ynew[kDyn];
    // (dt is wave-uniform; re-derived from the scalar here on purpose: hoisted out of the caller's attempt loop the product lives in
    //  a vector register that loop does not have -- it came back as a scratch load, waited for on the spot, in every attempt)
    float dts = dt;
    asm volatile("" : "+s"(dts));
    const float min_step = 1e-6f * dts;   // fp32 stand-in for scipy's 10 ulp(t)
    const int tries = rk45_tries(L);
    if (tries == 0) L.h_abs = fmaxf(L.h_abs, min_step);   // a new step starts
    if (L.h_abs < min_step || tries >= 64) {
        L.ctl &= ~3u;   // status -1
        return;
    }
    const float t = L.t;
    const float h = fminf(L.h_abs, dt - t);
    L.h_abs = h;
    const float hh = h * h;
#pragma unroll
    for (int d = 0; d < D; ++d) K[0][d] = L.K0[d];
    // ---- rk_step (rk.py): stages 1..5, y_new, f_new --------------------------------------------
#pragma unroll
    for (int st = 1; st < 6; ++st) {
        float ys[kDyn];
        {   // altitude (the only position the forces read): x + h C v_x + h^2 sum_m AA[st][m] Kv_x[m]
            float acc = 0.0f;
            bool f1 = true;
#pragma unroll
            for (int m = 0; m + 1 < st; ++m) acc_c(T.AA[st][m], K[m][0], acc, f1);
            ys[0] = fmaf(hh, acc, fmaf(h * C[st], y[3], y[0]));
            ys[1] = ys[2] = 0.0f;   // never read
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float acc = 0.0f;
            bool f1 = true;
#pragma unroll
            for (int j = 0; j < st; ++j) acc_c(A[st][j], K[j][d], acc, f1);
            ys[comp(d)] = fmaf(h, acc, y[comp(d)]);
        }
        ys[10] = y[10];
        rk45_eval(ys, frcp(fmaf(-mdot, t + C[st] * h, mass0)), c, J, K[st]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float acc = 0.0f;
        bool f1 = true;
#pragma unroll
        for (int m = 0; m < 5; ++m) acc_c(T.BA[m], K[m][i], acc, f1);
        ynew[i] = fmaf(hh, acc, fmaf(h, y[3 + i], y[i]));
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float acc = 0.0f;
        bool f1 = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) acc_c(B[j], K[j][d], acc, f1);
        ynew[comp(d)] = fmaf(h, acc, y[comp(d)]);
    }
    ynew[10] = y[10];
    rk45_eval(ynew, frcp(fmaf(-mdot, t + h, mass0)), c, J, Kn);
    L.ctl += 1u << 9;   // one more attempt (6 RHS evaluations)
    // ---- error norm ------------------------------------------------------------------------------
    float en = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // position rows: h^2 sum_m EA[m] Kv[m]
        float e = 0.0f;
        bool f1 = true;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc_c(T.EA[m], K[m][i], e, f1);
        const float sc = fmaf(fmaxf(fabsf(y[i]), fabsf(ynew[i])), kRtol, kAtol);
        en += sq(e * hh * frcp(sc));
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        float e = E[6] * Kn[d];
#pragma unroll
        for (int j = 5; j >= 0; --j) e = fma_c(E[j], K[j][d], e);
        const float sc = fmaf(fmaxf(fabsf(y[comp(d)]), fabsf(ynew[comp(d)])), kRtol, kAtol);
        en += sq(e * h * frcp(sc));
    }
    // en = error_norm^2 (omega_x and the mass contribute 0 to the sum, 1 each to the count); norm < 1 <=> en < 1, and
    // norm^-1/5 = en^-1/10 = exp2(-0.1 log2 en)  (en == 0: exp2(+inf) = inf, cut to kMaxFactor like scipy's special case)
    en *= (1.0f / 14.0f);
    const float pw = kSafety * fexp2(-0.1f * flog2(en));
    if (!(en < 1.0f)) {
        L.h_abs *= fmaxf(kMinFactor, pw);
        L.ctl += 1u << 2;   // one more rejection of this step
        return;
    }
    {
        float factor = fminf(kMaxFactor, pw);
        if (tries > 0) factor = fminf(1.0f, factor);
        L.h_abs *= factor;
        L.ctl &= ~(127u << 2);   // accepted: the next attempt starts a new step
    }
    // ---- terminal event on the accepted sub-step (ivp.py find_active_events / handle_events) ---------
    if (crossing(y[0], ynew[0])) {
        // y(s) = y_old + h s (Q0 + Q1 s + Q2 s^2 + Q3 s^3), Q = K^T P; position rows through PS / PA (see Derived)
        auto q_pos = [&](int i, float (&qq)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float acc = 0.0f;
                bool f1 = true;
#pragma unroll
                for (int m = 0; m < 6; ++m) acc_c(T.PA[m][k], K[m][i], acc, f1);
                qq[k] = fmaf(h, acc, T.PS[k] * y[3 + i]);
            }
        };
        float q[4];
        q_pos(0, q);
        const float cx[5] = {y[0], h * q[0], h * q[1], h * q[2], h * q[3]};
        const float th = poly_root01(cx, y[0], ynew[0]);
        float yev[kDyn];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            float qq[4];
            q_pos(i, qq);
            yev[i] = fmaf(h * th, fmaf(fmaf(fmaf(qq[3], th, qq[2]), th, qq[1]), th, qq[0]), y[i]);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            float qq[4] = {0, 0, 0, 0};
            bool fq[4] = {true, true, true, true};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc_c(P[j][k], K[j][d], qq[k], fq[k]);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc_c(P[6][k], Kn[d], qq[k], fq[k]);
            yev[comp(d)] = fmaf(h * th, fmaf(fmaf(fmaf(qq[3], th, qq[2]), th, qq[1]), th, qq[0]), y[comp(d)]);
        }
#pragma unroll
        for (int i = 0; i < kDyn; ++i) {
            if (i != 10) y[i] = yev[i];
        }
        L.t = t + th * h;
        L.ctl = (L.ctl & ~3u) | 2u;   // status 1
    } else {
#pragma unroll
        for (int i = 0; i < kDyn; ++i) y[i] = ynew[i];
#pragma unroll
        for (int d = 0; d < D; ++d) L.K0[d] = Kn[d];   // FSAL
        L.t = t + h;
    }
}

// End of the env step: the mass at the time reached, the quaternion renormalised (sim:97).
__device__ __forceinline__ void rk45_end(float (&y)[kDyn], float& mass, const Control& c, const Rk45Lane& L) {
    mass = fmaf(-c.mdot, L.t, mass);
    const float n2 = fmaf(y[6], y[6], fmaf(y[7], y[7], fmaf(y[8], y[8], y[9] * y[9])));
    const float inv = frsq(n2);
    y[6] *= inv;
    y[7] *= inv;
    y[8] *= inv;
    y[9] *= inv;
}

// One env step [0, dt] of the adaptive integrator.  Returns status (0 reached dt, 1 ground event,
// -1 step size underflow -- the reference folds any non-zero status into done, env:213).
// nfev receives the number of RHS evaluations (scipy's sol.nfev).
__device__ __forceinline__ int integrate_rk45(float (&y)[kDyn], float& mass, const Control& c, const Inertia& J, float dt,
                                              int& nfev) {
    Rk45Lane L;
    rk45_begin(y, mass, c, J, dt, L);
#pragma unroll 1
    while (rk45_active(L, dt)) rk45_attempt(y, mass, c, J, dt, L);
    rk45_end(y, mass, c, L);
    nfev = rk45_nfev(L);
    return rk45_status(L);
}

// ---- reward (env:317-402) ------------------------------------------------------------------------

// env:528-546 -- t_go = largest positive real root of  g^2 t^4 + c2 t^2 + c1 t + c0,
// c2 = -4|v|^2 <= 0, c1 = -24 r.v, c0 = -36|r|^2 <= 0.  (np.roots order + "first positive real"
// selects the largest positive real root: 38k multi-root cases checked in the build container and
// golden G5.)  With the monic coefficients b2, b1, b0:  p is convex for t > ti = sqrt(-b2/6) and
// concave on [0, ti], p(0) = b0 < 0.  The branch that holds the largest root is decided up front
// from p(ti), p'(ti):
//   * p(ti) <= 0, or p'(ti) < 0:  look on the convex branch.  Newton from a point right of every
//     root -- tq = largest root of the bi-quadratic t^4 + b2 t^2 + b0 when b1 >= 0 (p(tq) = b1 tq >= 0),
//     a Fujiwara bound when b1 < 0 -- descends monotonically to the largest root;
//   * p(ti) > 0 and p'(ti) >= 0:  no root right of ti; the single crossing of the concave branch is
//     reached monotonically by Newton from t = 0.
// Only when p(ti) > 0 > p'(ti) and the convex branch turns out to hold no root (it has a positive
// minimum) does the iteration restart from 0 (0.1 % of states).  Eight unrolled, branch-free Newton
// steps cover 99 % of the lanes; a wave-uniform tail loop finishes the rest.  fp32 iterations, then
// one Newton correction with the residual evaluated in fp64.
// (Round 3 tried a Laguerre step in front of four Newton steps -- 65 instead of 88 issue slots and, emulated against float64
// roots on 655 360 rollout states, the same accuracy.  On the GPU it was 1.5 us per 2^20-env step SLOWER: the convergence test
// below looks at the size of the LAST step, which after five iterations is still ~1e-4 t for lanes that have in fact arrived,
// so most waves entered the safeguarded tail; it also failed the near-double-root family.  Not kept.)
struct Quartic {
    float b2, b1, b0, ti;
};

// General Newton step with the safeguards (monotone clamp, branch switch, restart): used by the rare
// wave-uniform tail loop only.
__device__ __forceinline__ void quartic_newton_safe(const Quartic& q, float& t, bool& left, bool& conv, bool allow_restart) {
    const float t2 = t * t;
    const float p = fmaf(fmaf(t2 + q.b2, t, q.b1), t, q.b0);
    const float dp = fmaf(fmaf(4.0f, t2, 2.0f * q.b2), t, q.b1);
    const float step = (dp > 0.0f) ? p * frcp(dp) : 0.0f;
    float tn = t - step;
    // On the convex branch (t > ti) a Newton iterate taken right of a root stays right of it, so an iterate
    // that leaves the branch -- slope <= 0, or a step that lands left of the inflection point, e.g. from the
    // nearly flat bottom of a positive minimum -- proves the branch holds no root: restart on the concave one.
    if (allow_restart && !left && p > 0.0f && (dp <= 0.0f || t < q.ti || tn < q.ti)) {
        left = true;
        t = 0.0f;
        conv = false;
        return;
    }
    left = left || (p <= 0.0f);                      // at / left of the root: finish from the left
    tn = left ? fmaxf(tn, t) : fminf(tn, t);          // monotone (guards round-off)
    conv = fabsf(tn - t) <= 2e-7f * fabsf(tn);
    t = tn;
}

__device__ __forceinline__ float t_go_quartic(float c2, float c1, float c0) {
    constexpr float g2 = (float)(kGShaping * kGShaping);
    constexpr float ig2 = (float)(1.0 / (kGShaping * kGShaping));
    Quartic q;
    q.b2 = c2 * ig2;
    q.b1 = c1 * ig2;
    q.b0 = c0 * ig2;
    const float ti2 = q.b2 * (-1.0f / 6.0f);
    q.ti = fsqrt(ti2);
    const float p_i = fmaf(-5.0f * ti2, ti2, fmaf(q.b1, q.ti, q.b0));   // p(ti), using b2 = -6 ti^2
    const float dp_i = fmaf(-8.0f * ti2, q.ti, q.b1);                   // p'(ti)
    bool left = (p_i > 0.0f) && (dp_i >= 0.0f);
    const bool may_restart = (p_i > 0.0f) && (dp_i < 0.0f);
    float t;
    {
        const float tq = fsqrt(0.5f * (fsqrt(fmaf(q.b2, q.b2, -4.0f * q.b0)) - q.b2));
        const float m1 = fexp2(flog2(fmaxf(fabsf(q.b1), 1e-30f)) * (1.0f / 3.0f));
        const float fuji = 2.002f * fmaxf(fsqrt(fabsf(q.b2)), fmaxf(m1, fsqrt(fsqrt(0.5f * fabsf(q.b0))))) + 1e-6f;
        t = left ? 0.0f : ((q.b1 >= 0.0f) ? tq : fuji);
    }
    // Fast path: 8 bare Newton steps.  On the branch chosen above p' > 0 between the start and the root
    // (convex branch right of the largest root / concave branch left of its only root), so no clamp
    // is needed: an overshoot by round-off lands next to the root, where Newton simply steps back.
    const float b2x2 = 2.0f * q.b2;
    float step = 0.0f;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const float t2 = t * t;
        const float p = fmaf(fmaf(t2 + q.b2, t, q.b1), t, q.b0);
        const float dp = fmaf(fmaf(4.0f, t2, b2x2), t, q.b1);
        step = p * frcp(dp);
        t -= step;
    }
    bool conv = fabsf(step) <= 4e-7f * t;   // also false for NaN / a fall off the branch (dp <= 0 => huge or negative step)
    if (__builtin_expect(__any(!conv || may_restart), 0)) {
        // tail: slow lanes, and the rare lanes that may have to restart on the concave branch
        left = (p_i > 0.0f) && (dp_i >= 0.0f);
        if (!(t > 0.0f) || !(t < 3.0e38f)) {   // re-seed a lane whose bare iteration went astray (NaN, <= 0, inf)
            const float tq = fsqrt(0.5f * (fsqrt(fmaf(q.b2, q.b2, -4.0f * q.b0)) - q.b2));
            t = left ? 0.0f : 2.002f * fmaxf(tq, fexp2(flog2(fmaxf(fabsf(q.b1), 1e-30f)) * (1.0f / 3.0f))) + 1e-6f;
        }                                      // slow-but-healthy lanes simply continue from where they are
#pragma unroll 1
        for (int it = 0; it < 40; ++it) {
            quartic_newton_safe(q, t, left, conv, may_restart);
            if (__all(conv)) break;
        }
    }
    // fp64 polish on the un-normalised polynomial (coefficients as the reference hands them to
    // np.roots: float32 values promoted to float64)
    const double td = (double)t;
    const double pd = fma(fma(fma((double)g2 * td, td, (double)c2), td, (double)c1), td, (double)c0);
    const float dpf = fmaf(fmaf(4.0f * g2, t * t, 2.0f * c2), t, c1);
    const float corr = (float)pd * frcp(dpf);
    if (!(dpf > 0.0f)) return t;
    // An ill-conditioned root (next to the local maximum: p' small) is still ~1e-5 off after one correction;
    // such lanes -- a few per million -- take two more Newton steps entirely in fp64 (wave-uniform branch).
    if (__builtin_expect(__any(fabsf(corr) > 2e-6f * t), 0)) {
        double x = td - (double)corr;
#pragma unroll 1
        for (int it = 0; it < 2; ++it) {
            const double px = fma(fma(fma((double)g2 * x, x, (double)c2), x, (double)c1), x, (double)c0);
            const double dx = fma(fma(4.0 * (double)g2 * x, x, 2.0 * (double)c2), x, (double)c1);
            if (dx > 0.0) x -= px / dx;
        }
        return (fabsf(corr) > 2e-6f * t) ? (float)x : t - corr;
    }
    return t - corr;
}

struct RewardOut {
    float terms[R6_NUM_TERMS];
    float sum;
    bool landed;  // all landing conditions of env:382-390 hold
};

// env:317-402 on the fp32 post-step state; `c` is the control applied during the step (env:339 and
// sim:177-186 use the post-step attitude with the current action).
template <int SHAPING>
__device__ __forceinline__ RewardOut reward(const R6Params& p, const float (&y)[kDyn], float mass, const Control& c,
                                            float v0) {
    RewardOut o;
    const float rx = y[0], ry = y[1], rz = y[2];
    const float vx = y[3], vy = y[4], vz = y[5];
    const Rot R = rotmat<true>(y[6], y[7], y[8], y[9]);   // post-step attitude: renormalised by integrate()
    const float vv = fmaf(vx, vx, fmaf(vy, vy, vz * vz));
    const float rr = fmaf(rx, rx, fmaf(ry, ry, rz * rz));
    const float vn = fsqrt(vv);
    const float rn = fsqrt(rr);

    float track;
    if constexpr (SHAPING == R6_SHAPING_ACCELERATION) {
        // env:526-566
        const float rv = fmaf(rx, vx, fmaf(ry, vy, rz * vz));
        const float tgo = t_go_quartic(-4.0f * (vn * vn), -24.0f * rv, -36.0f * (rn * rn));
        const float it = frcp(tgo);
        const float k6 = -6.0f * it * it, k4 = -4.0f * it;
        float ax = fmaf(k6, rx, fmaf(k4, vx, (float)kGShaping));  // - g,  g = [-9.81,0,0]
        float ay = fmaf(k6, ry, k4 * vy);
        float az = fmaf(k6, rz, k4 * vz);
        const float an = fsqrt(fmaf(ax, ax, fmaf(ay, ay, az * az)));
        const float im = frcp(mass);
        const float lim = p.max_thrust * im;                      // env:562
        if (!(an <= lim)) {
            const float sc = lim * frcp(an);
            ax *= sc;
            ay *= sc;
            az *= sc;
        }
        // env:339-340: a = R(q_post) T_B / m
        const float tx = fmaf(R.r00, c.tbx, fmaf(R.r01, c.tby, R.r02 * c.tbz)) * im - ax;
        const float ty = fmaf(R.r10, c.tbx, fmaf(R.r11, c.tby, R.r12 * c.tbz)) * im - ay;
        const float tz = fmaf(R.r20, c.tbx, fmaf(R.r21, c.tby, R.r22 * c.tbz)) * im - az;
        track = p.alfa * fsqrt(fmaf(tx, tx, fmaf(ty, ty, tz * tz)));
    } else {
        // env:646-674
        float hx, hy, hz, ux, tau;
        if (rx > p.waypoint) {
            hx = rx - p.waypoint; hy = ry; hz = rz; ux = vx + 2.0f; tau = 20.0f;
        } else {
            hx = rx + 1.0f; hy = 0.0f; hz = 0.0f; ux = vx + 1.0f; tau = 100.0f;
        }
        const float hn = fsqrt(fmaf(hx, hx, fmaf(hy, hy, hz * hz)));
        const float un = fsqrt(fmaf(ux, ux, fmaf(vy, vy, vz * vz)));
        const float tgo = hn / un;
        const float g = -v0 * (1.0f - expf(-tgo / tau)) / fmaxf(1e-3f, hn);
        const float dx = vx - g * hx, dy = vy - g * hy, dz = vz - g * hz;
        track = p.alfa * fsqrt(fmaf(dx, dx, fmaf(dy, dy, dz * dz)));
    }

    // env:366-369 / :386-388 -- Euler zyx limits without inverse trig (see r6dof.h)
    const float h0 = fsqrt(fmaf(R.r00, R.r00, R.r01 * R.r01));
    const float h2 = fsqrt(fmaf(R.r22, R.r22, R.r12 * R.r12));
    const float a1 = fabsf(R.r02);
    const bool exceed = (R.r00 < p.traj_c0 * h0) || (a1 > p.traj_s1) || (R.r22 < p.traj_c2 * h2);
    const bool att_ok = (R.r00 > p.land_c0 * h0) || (a1 < p.land_s1) || (R.r22 > p.land_c2 * h2);

    // env:371-402
    const bool zero_h = rx <= 1e-3f;
    const bool om_ok = (fabsf(y[10]) < p.omega_lim[0]) || (fabsf(y[11]) < p.omega_lim[1]) ||
                       (fabsf(y[12]) < p.omega_lim[2]);
    const bool landed = zero_h && (vn < p.land_vmax) && (rn < p.land_radius) && att_ok && om_ok;

    o.terms[0] = track;
    o.terms[1] = p.beta * c.thrust;                                   // env:355
    o.terms[2] = p.eta;                                               // env:356
    o.terms[3] = exceed ? p.gamma : 0.0f;                             // env:357
    o.terms[4] = landed ? p.kappa : 0.0f;                             // env:399
    o.terms[5] = fmaxf(p.max_r_f - rn, 0.0f) * p.w_r_f;               // env:400
    o.terms[6] = (rn < p.max_r_f && zero_h) ? fmaxf(p.max_v_f - vn, 0.0f) * p.w_v_f : 0.0f;  // env:401
    o.terms[7] = 0.0f;
    o.landed = landed;
    {
        // The sum of the seven terms, in the reference's order (env:342-361) and with NO contraction: several terms are products
        // (beta T, w_r_f max(..), ...), and hipcc may fuse such a product into the addition that consumes it in one inlined
        // copy of this function and not in another -- the copy of step_kernel's tile loop and the copy of its guest flush
        // disagreed in the last bit of the total with all seven terms equal (round 5).  Without the licence to contract, every
        // copy rounds every term and every partial sum: the same total wherever this is inlined.
#pragma clang fp contract(off)
        o.sum = ((((((o.terms[0] + o.terms[1]) + o.terms[2]) + o.terms[3]) + o.terms[4]) + o.terms[5]) + o.terms[6]);
    }
    return o;
}

// VerticalAttitudeReward term (wrappers.py:143-147): clip(2 * degrees(acos(q0)) * weight, -10, 10).
// acos(q0) of the unit quaternion is taken as atan2(|q_v|, q0): near upright (q0 -> 1) acos is
// ill-conditioned in fp32 (3e-5 relative at 2.6 deg), the vector part carries the angle exactly.
// Out of line on purpose: it runs at touch-down only and atan2f's registers must not count against
// the hot loop's 96-VGPR budget.
__device__ __noinline__ float vertical_attitude_term(float q0, float qv, float weight) {
    return fminf(fmaxf(2.0f * 57.29577951308232f * atan2f(qv, q0) * weight, -10.0f), 10.0f);
}

// Total step reward from the 7 terms: env:228-229 (bounds penalty) and the reference's optional reward
// wrappers (my_environment/wrappers/wrappers.py).  `a2n` is the normalised throttle action, `q0` / `x` the
// post-step attitude scalar and height.  May rewrite terms[] (annealed mode, vertical-attitude term).
__device__ __forceinline__ float finalize_reward(const R6Params& p, RewardOut& rw, bool bnd, float a2n, const float (&y)[kDyn]) {
    const float x = y[0], q0 = y[6];
    float r;
    if (p.reward_mode == R6_REWARD_ANNEALED) {
        // RewardAnnealing (wrappers.py:39-61): the env reward is discarded (bounds penalty included) and
        // rebuilt from four terms plus a new thrust penalty -xi * (action[2] + 1)
        const float tpen = -p.xi * (a2n + 1.0f);
        rw.terms[0] = 0.0f;
        rw.terms[1] = tpen;
        rw.terms[2] = 0.0f;
        {
#pragma clang fp contract(off)
            r = (((rw.terms[3] + rw.terms[4]) + rw.terms[5]) + rw.terms[6]) + tpen;
        }
    } else {
        r = rw.sum + (bnd ? p.bounds_penalty : 0.0f);
    }
    if (p.vatt_enable && x < p.vatt_threshold && rw.terms[6] > 0.0f) {
        // VerticalAttitudeReward (wrappers.py:128-155)
        const float qv = fsqrt(fmaf(y[7], y[7], fmaf(y[8], y[8], y[9] * y[9])));
        const float va = vertical_attitude_term(q0, qv, p.vatt_weight);
        rw.terms[7] = va;
        r += va;
    }
    return r;
}

// env:591-593 -- inclusive float32 box; NaN counts as a violation (Box.contains is False).
__device__ __forceinline__ bool bounds_violation(const R6Params& p, float x, float y, float z) {
    const bool inside = (x >= p.bound_lo[0]) && (x <= p.bound_hi[0]) && (y >= p.bound_lo[1]) &&
                        (y <= p.bound_hi[1]) && (z >= p.bound_lo[2]) && (z <= p.bound_hi[2]);
    return !inside;
}

// ---- counter-based reset RNG: Philox-4x32-10 ------------------------------------------------------
struct U4 {
    uint32_t x, y, z, w;
};

__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// env:180-199 -- initial condition of global env `gid` reset at global step `step_index`
// (same stream definition as oracle_np.reset_sample).  Returns the 14-vector, quaternion normalised.
__device__ __forceinline__ void sample_ic(const R6Params& p, uint64_t seed, uint64_t gid, uint64_t step_index,
                                          float (&ic)[R6_STATE_DIM]) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t w[16];
#pragma unroll
    for (uint32_t b = 0; b < 4; ++b) {
        const U4 r = philox4x32_10(U4{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step_index,
                                      ((uint32_t)(step_index >> 32) & 0x3FFFFFFFu) | (b << 30)},
                                   k0, k1);
        w[4 * b + 0] = r.x;
        w[4 * b + 1] = r.y;
        w[4 * b + 2] = r.z;
        w[4 * b + 3] = r.w;
    }
#pragma unroll
    for (int j = 0; j < R6_STATE_DIM; ++j) {
        const float u = (float)(w[j] >> 8) * 5.9604644775390625e-08f;  // 2^-24
        ic[j] = fmaf(u, p.ic_range[j], p.ic_lo[j]);
    }
    const float n2 = fmaf(ic[6], ic[6], fmaf(ic[7], ic[7], fmaf(ic[8], ic[8], ic[9] * ic[9])));
    const float inv = 1.0f / sqrtf(n2);
    ic[6] *= inv;
    ic[7] *= inv;
    ic[8] *= inv;
    ic[9] *= inv;
}

// The same draw for ONE env, computed by a whole wave: the four Philox blocks of sample_ic() run side by side in the lanes
// (block = lane & 3) and the sixteen words come back through v_readlane, so every lane leaves with the same 14 values.
// Where one lane of a wave needs a fresh initial condition (r6_step_k: ~1 finished env per wave and step) this is one
// Philox pass instead of four with 63 lanes masked off.  `gid` / `step_index` are wave-uniform.  Bit for bit sample_ic().
__device__ __forceinline__ void sample_ic_wave(const R6Params& p, uint64_t seed, uint64_t gid, uint64_t step_index, int lane,
                                               float (&ic)[R6_STATE_DIM]) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t blk = (uint32_t)lane & 3u;
    const U4 r = philox4x32_10(U4{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step_index,
                                  ((uint32_t)(step_index >> 32) & 0x3FFFFFFFu) | (blk << 30)},
                               k0, k1);
    uint32_t w[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        w[4 * b + 0] = (uint32_t)__builtin_amdgcn_readlane((int)r.x, b);
        w[4 * b + 1] = (uint32_t)__builtin_amdgcn_readlane((int)r.y, b);
        w[4 * b + 2] = (uint32_t)__builtin_amdgcn_readlane((int)r.z, b);
        w[4 * b + 3] = (uint32_t)__builtin_amdgcn_readlane((int)r.w, b);
    }
#pragma unroll
    for (int j = 0; j < R6_STATE_DIM; ++j) {
        const float u = (float)(w[j] >> 8) * 5.9604644775390625e-08f;  // 2^-24
        ic[j] = fmaf(u, p.ic_range[j], p.ic_lo[j]);
    }
    const float n2 = fmaf(ic[6], ic[6], fmaf(ic[7], ic[7], fmaf(ic[8], ic[8], ic[9] * ic[9])));
    const float inv = 1.0f / sqrtf(n2);
    ic[6] *= inv;
    ic[7] *= inv;
    ic[8] *= inv;
    ic[9] *= inv;
}

// sample_ic_wave() with the reset distribution in LDS (`lo`, `range`: 16 floats each, entries 14 and 15 zero) and the words turned into
// initial conditions BEFORE they leave their lanes: lane b < 4 converts the four words of Philox block b (14 of the 16 are used) --
// 16 conversion instructions for the wave instead of 56 on wave-uniform values, and no scalar registers for the 28 distribution
// parameters (r6_step_k's dynamics wave pays every instruction of a reset on the chain of its tile's K steps: 0.84 us per episode
// end, round 5).  The same operations on the same operands: bit for bit sample_ic().
__device__ __forceinline__ void sample_ic_wave_lds(const float* lo, const float* range, uint64_t seed, uint64_t gid, uint64_t step_index, int lane,
                                                   float (&ic)[R6_STATE_DIM]) {
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t blk = (uint32_t)lane & 3u;
    const U4 r = philox4x32_10(U4{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step_index,
                                  ((uint32_t)(step_index >> 32) & 0x3FFFFFFFu) | (blk << 30)},
                               k0, k1);
    const int j0 = 4 * (int)blk;
    const float f0 = fmaf((float)(r.x >> 8) * 5.9604644775390625e-08f, range[j0 + 0], lo[j0 + 0]);
    const float f1 = fmaf((float)(r.y >> 8) * 5.9604644775390625e-08f, range[j0 + 1], lo[j0 + 1]);
    const float f2 = fmaf((float)(r.z >> 8) * 5.9604644775390625e-08f, range[j0 + 2], lo[j0 + 2]);
    const float f3 = fmaf((float)(r.w >> 8) * 5.9604644775390625e-08f, range[j0 + 3], lo[j0 + 3]);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        ic[4 * b + 0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f0), b));
        ic[4 * b + 1] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f1), b));
        if (4 * b + 2 < R6_STATE_DIM) ic[4 * b + 2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f2), b));
        if (4 * b + 3 < R6_STATE_DIM) ic[4 * b + 3] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f3), b));
    }
    const float n2 = fmaf(ic[6], ic[6], fmaf(ic[7], ic[7], fmaf(ic[8], ic[8], ic[9] * ic[9])));
    const float inv = 1.0f / sqrtf(n2);
    ic[6] *= inv;
    ic[7] *= inv;
    ic[8] *= inv;
    ic[9] *= inv;
}

}  // namespace r6
