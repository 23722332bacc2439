/*
 * oracle_c.c -- plain-C (float64) restatement of the Rocket6DOF.step() hot path.
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Built by oracle/Makefile into oracle/_build/liboracle.so
 * and used (through oracle/oracle_c.py) only by tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py, as the checker / the timed "best CPU" baseline (OpenMP over environments).  The
 * product never links or calls it.
 *
 * It restates the fixed-step algorithm the HIP kernel implements (same spec as
 * oracle/oracle_np.py sim_step_fixed + env_step), written independently of both: inverse trig for
 * the Euler-angle limits (the kernel is trig-free), a float64 quartic solver, array-of-structs I/O.
 * It is pinned by the golden vectors (tests/test_oracle_c.py).  Citations: reference
 * my_environment/utils/simulator.py ("sim") and my_environment/envs/rocket_env.py ("env").
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define NS 14

typedef struct OcParams {
    double dt;
    int32_t integrator; /* 0 euler, 1 rk4 */
    int32_t n_substeps;
    int32_t shaping;    /* 0 acceleration, 1 velocity */
    int32_t auto_reset;
    double max_gimbal, max_thrust;
    double norm[NS];
    float bound_lo[3], bound_hi[3];
    float pad_[2];
    double alfa, beta, eta, gamma, kappa, w_r_f, w_v_f, max_r_f, max_v_f;
    double att_traj[3], att_land[3]; /* rad */
    double land_radius, land_vmax, omega_lim[3], waypoint;
    float ic_lo[NS], ic_range[NS];
    int32_t clip_reward;
    int32_t max_episode_steps;
    double clip_lo, clip_hi;
} OcParams;

/* sim:39-67 */
static const double G0 = 9.81, LEN = 40.0, RAD = 3.66 / 2.0, ISP = 360.0, CA = 0.82;
static const double RHO0 = 1.225, TB = 288.15, LB = -0.0065, MAIR = 0.0289644, RSTAR = 8.3144598;
static const double ARM_T = -15.0, ARM_CP = 5.0, PI_ = 3.14159265358979323846;

static void rotmat(const double *q, double R[9]) { /* sim:189-199: normalised quaternion */
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}

static void thrust_body(const float u[3], double tb[3]) { /* sim:167-175, :202-214 (float32 trig) */
    float cy = cosf(u[0]), cz = cosf(u[1]), sy = sinf(u[0]), sz = sinf(u[1]);
    tb[0] = (double)(cy * cz) * (double)u[2];
    tb[1] = (double)(sy * cz) * (double)u[2];
    tb[2] = (double)sz * (double)u[2];
}

void oc_rhs(const double *y, const float u[3], double m0, double *dy) { /* sim:106-143 */
    double R[9], tb[3], ab[3], vb[3];
    const double *v = y + 3, *q = y + 6, *w = y + 10;
    double rho = RHO0 * pow(TB / (TB + y[0] * LB), 1.0 + G0 * MAIR / RSTAR / LB);
    rotmat(q, R);
    thrust_body(u, tb);
    double speed = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double sref = PI_ * RAD * RAD;
    for (int i = 0; i < 3; ++i) vb[i] = R[i] * v[0] + R[3 + i] * v[1] + R[6 + i] * v[2]; /* R^T v */
    for (int i = 0; i < 3; ++i) ab[i] = -0.5 * rho * speed * sref * CA * vb[i];           /* sim:216-219 */
    double fb[3] = {tb[0] + ab[0], tb[1] + ab[1], tb[2] + ab[2]};
    for (int i = 0; i < 3; ++i) {
        dy[i] = v[i];
        dy[3 + i] = (R[3 * i] * fb[0] + R[3 * i + 1] * fb[1] + R[3 * i + 2] * fb[2]) / y[13];
    }
    dy[3] -= G0;
    dy[6] = 0.5 * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]); /* sim:221-229 */
    dy[7] = 0.5 * (w[0] * q[0] + w[2] * q[2] - w[1] * q[3]);
    dy[8] = 0.5 * (w[1] * q[0] - w[2] * q[1] + w[0] * q[3]);
    dy[9] = 0.5 * (w[2] * q[0] + w[1] * q[1] - w[0] * q[2]);
    double J[3] = {0.5 * m0 * RAD * RAD, m0 * (LEN * LEN + 3 * RAD * RAD) / 12.0, m0 * (LEN * LEN + 3 * RAD * RAD) / 12.0};
    double tau[3] = {0.0, -ARM_T * tb[2] - ARM_CP * ab[2], ARM_T * tb[1] + ARM_CP * ab[1]}; /* sim:232-244 */
    double Jw[3] = {J[0] * w[0], J[1] * w[1], J[2] * w[2]};
    double cr[3] = {w[1] * Jw[2] - w[2] * Jw[1], w[2] * Jw[0] - w[0] * Jw[2], w[0] * Jw[1] - w[1] * Jw[0]};
    for (int i = 0; i < 3; ++i) dy[10 + i] = (tau[i] - cr[i]) / J[i];
    dy[13] = -(double)u[2] / (G0 * ISP); /* sim:140-141 */
}

static void map_step(const double *y, const float u[3], double m0, double h, int rk4, double *out) {
    double k1[NS], k2[NS], k3[NS], k4[NS], t[NS];
    oc_rhs(y, u, m0, k1);
    if (!rk4) {
        for (int i = 0; i < NS; ++i) out[i] = y[i] + h * k1[i];
        return;
    }
    for (int i = 0; i < NS; ++i) t[i] = y[i] + 0.5 * h * k1[i];
    oc_rhs(t, u, m0, k2);
    for (int i = 0; i < NS; ++i) t[i] = y[i] + 0.5 * h * k2[i];
    oc_rhs(t, u, m0, k3);
    for (int i = 0; i < NS; ++i) t[i] = y[i] + h * k3[i];
    oc_rhs(t, u, m0, k4);
    for (int i = 0; i < NS; ++i) out[i] = y[i] + h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
}

static double poly_root01(const double *c, int n, double x0, double x1) { /* same procedure as oracle_np */
    if (x0 == 0.0) return 0.0;
    if (x1 == 0.0) return 1.0;
    double lo = 0, hi = 1, th = x0 / (x0 - x1);
    for (int it = 0; it < 8; ++it) {
        double f = c[n - 1], df = 0;
        for (int k = n - 2; k >= 0; --k) { df = df * th + f; f = f * th + c[k]; }
        if (f == 0.0) break;
        if ((f > 0) == (x0 > 0)) lo = th; else hi = th;
        double nt = df != 0.0 ? th - f / df : 0.5 * (lo + hi);
        if (!(nt >= lo && nt <= hi)) nt = 0.5 * (lo + hi);
        if (nt == th) break;
        th = nt;
    }
    return th;
}

/* Simulator6DOF.step with fixed sub-steps + ground-event localisation (oracle_np.sim_step_fixed). */
int oc_sim_step(double *y, const float u[3], double m0, double dt, int rk4, int nsub) {
    double h = dt / nsub, y1[NS];
    int status = 0;
    for (int s = 0; s < nsub && !status; ++s) {
        map_step(y, u, m0, h, rk4, y1);
        int cross = (y[0] <= 0 && y1[0] >= 0) || (y[0] >= 0 && y1[0] <= 0); /* scipy direction==0 rule */
        if (cross) {
            status = 1;
            if (rk4) {
                /* Hermite dense output of the step with exact end slopes -- oracle_np.event_step_rk4 */
                double f0[NS], f1[NS], d0[NS], c2[NS], c3[NS], ye[NS];
                oc_rhs(y, u, m0, f0);
                oc_rhs(y1, u, m0, f1);
                for (int i = 0; i < NS; ++i) {
                    d0[i] = h * f0[i];
                    double D = (y1[i] - y[i]) - d0[i], E = h * f1[i] - d0[i];
                    c2[i] = 3 * D - E;
                    c3[i] = E - 2 * D;
                }
                double c[5] = {y[0], h * y[3], h * (d0[3] / 2), h * (c2[3] / 3), h * (c3[3] / 4)};
                double th = poly_root01(c, 5, y[0], y1[0]);
                for (int i = 0; i < NS; ++i) ye[i] = y[i] + th * (d0[i] + th * (c2[i] + th * c3[i]));
                for (int i = 0; i < 3; ++i)
                    ye[i] = y[i] + (h * th) * (y[3 + i] + th * (d0[3 + i] / 2 + th * (c2[3 + i] / 3 + th * (c3[3 + i] / 4))));
                memcpy(y1, ye, sizeof(ye));
                if (y1[3] != 0.0) {
                    double delta = -y1[0] / y1[3];
                    if (fabs(delta) <= h) for (int i = 0; i < 3; ++i) y1[i] += y1[3 + i] * delta;
                }
            } else {
                double th = y[0] == 0.0 ? 0.0 : y[0] / (y[0] - y1[0]);
                for (int i = 0; i < NS; ++i) y1[i] = y[i] + th * (y1[i] - y[i]);
            }
        }
        memcpy(y, y1, sizeof(y1));
    }
    double n = sqrt(y[6] * y[6] + y[7] * y[7] + y[8] * y[8] + y[9] * y[9]); /* sim:97 */
    for (int i = 6; i < 10; ++i) y[i] /= n;
    return status;
}

/* env:528-546 -- largest positive real root of a t^4 + c t^2 + d t + e (a > 0, e < 0), float64. */
static double tgo_quartic(double a, double c, double d, double e) {
    double b2 = c / a, b1 = d / a, b0 = e / a;
    double B = 1.0 + fmax(fabs(b2), fmax(fabs(b1), fabs(b0))); /* Cauchy bound on root moduli */
    double ti = sqrt(fabs(b2) / 6.0);
    double t = B;
    int left = 0;
    for (int it = 0; it < 200; ++it) {
        double p = ((t * t + b2) * t + b1) * t + b0, dp = (4 * t * t + 2 * b2) * t + b1;
        if (!left) {
            if (p <= 0) left = 1;
            /* leaving the convex branch (t > ti) with p > 0 -- by a non-positive slope or by a Newton step
               that lands left of the inflection point -- proves it holds no root: restart from 0 */
            else if (dp <= 0 || t < ti || t - p / dp < ti) { left = 1; t = 0; continue; }
        }
        if (!(dp > 0)) break;
        double tn = t - p / dp;
        tn = left ? fmax(tn, t) : fmin(tn, t);
        if (fabs(tn - t) <= 1e-15 * fabs(tn)) { t = tn; break; }
        t = tn;
    }
    return t;
}

static int bounds_violation(const float s[3], const OcParams *p) { /* env:591-593 */
    for (int i = 0; i < 3; ++i) if (!(s[i] >= p->bound_lo[i] && s[i] <= p->bound_hi[i])) return 1;
    return 0;
}

/* env:317-402 on the float32-cast post-step state; q64 = simulator's float64 quaternion (env:339). */
static void reward_terms(const float *s, const float u[3], const double *q64, const OcParams *p, double v0, double *terms,
                         int *landed) {
    double R[9], tb[3], track;
    float rn = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]); /* float32 norms like the reference */
    float vn = sqrtf(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
    if (p->shaping == 0) {
        float rv = s[0] * s[3] + s[1] * s[4] + s[2] * s[5];
        double g = 9.81; /* env:548 */
        double tgo = tgo_quartic(g * g, (double)(-4.0f * vn * vn), (double)(-24.0f * rv), (double)(-36.0f * rn * rn));
        double at[3];
        for (int i = 0; i < 3; ++i) at[i] = (double)(-6.0f * s[i]) / (tgo * tgo) - (double)(4.0f * s[3 + i]) / tgo;
        at[0] += g;
        double an = sqrt(at[0] * at[0] + at[1] * at[1] + at[2] * at[2]), lim = p->max_thrust / (double)s[13];
        if (!(an <= lim)) for (int i = 0; i < 3; ++i) at[i] *= lim / an;
        rotmat(q64, R);
        thrust_body(u, tb);
        double d2 = 0;
        for (int i = 0; i < 3; ++i) {
            double ai = (R[3 * i] * tb[0] + R[3 * i + 1] * tb[1] + R[3 * i + 2] * tb[2]) / (double)s[13] - at[i];
            d2 += ai * ai;
        }
        track = p->alfa * sqrt(d2);
    } else { /* env:646-674 */
        double rh[3], vh[3] = {s[3], s[4], s[5]}, tau;
        if (s[0] > p->waypoint) { rh[0] = s[0] - p->waypoint; rh[1] = s[1]; rh[2] = s[2]; vh[0] += 2.0; tau = 20; }
        else { rh[0] = s[0] + 1.0; rh[1] = 0; rh[2] = 0; vh[0] += 1.0; tau = 100; }
        double rhn = sqrt(rh[0] * rh[0] + rh[1] * rh[1] + rh[2] * rh[2]);
        double tgo = rhn / sqrt(vh[0] * vh[0] + vh[1] * vh[1] + vh[2] * vh[2]);
        double gsc = -v0 * (1 - exp(-tgo / tau)) / fmax(1e-3, rhn), d2 = 0;
        for (int i = 0; i < 3; ++i) { double di = (double)s[3 + i] - gsc * rh[i]; d2 += di * di; }
        track = p->alfa * sqrt(d2);
    }
    double q32[4] = {s[6], s[7], s[8], s[9]};
    rotmat(q32, R); /* env:210: rotation object from the float32 quaternion */
    double m02 = R[2] > 1 ? 1 : (R[2] < -1 ? -1 : R[2]);
    double eul[3] = {atan2(-R[1], R[0]), asin(m02), atan2(-R[5], R[8])}; /* scipy as_euler("zyx") */
    int exceed = 0, att_ok = 0, om_ok = 0;
    for (int i = 0; i < 3; ++i) {
        exceed |= fabs(eul[i]) > p->att_traj[i];               /* env:366-369 */
        att_ok |= fabs(eul[i]) < p->att_land[i];               /* env:386-388 (np.any) */
        om_ok |= fabsf(s[10 + i]) < (float)p->omega_lim[i];    /* env:389 (np.any) */
    }
    int zero_h = s[0] <= 1e-3f;
    *landed = zero_h && ((double)vn < p->land_vmax) && ((double)rn < p->land_radius) && att_ok && om_ok;
    terms[0] = track;
    terms[1] = p->beta * (double)u[2];
    terms[2] = p->eta;
    terms[3] = exceed ? p->gamma : 0.0;
    terms[4] = *landed ? p->kappa : 0.0;
    terms[5] = fmax(p->max_r_f - (double)rn, 0.0) * p->w_r_f;
    terms[6] = ((double)rn < p->max_r_f && zero_h) ? fmax(p->max_v_f - (double)vn, 0.0) * p->w_v_f : 0.0;
}

/* ---- Philox-4x32-10 reset stream (same definition as the kernel / oracle_np.reset_sample) ---------- */
static void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

void oc_reset_sample(const OcParams *p, uint64_t seed, uint64_t gid, uint64_t step_index, float *ic) {
    uint32_t w[16];
    for (uint32_t b = 0; b < 4; ++b) {
        uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step_index,
                         ((uint32_t)(step_index >> 32) & 0x3FFFFFFFu) | (b << 30)};
        philox(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        memcpy(w + 4 * b, c, sizeof(c));
    }
    for (int j = 0; j < NS; ++j) ic[j] = fmaf((float)(w[j] >> 8) * 5.9604644775390625e-08f, p->ic_range[j], p->ic_lo[j]);
    float n = sqrtf(ic[6] * ic[6] + ic[7] * ic[7] + ic[8] * ic[8] + ic[9] * ic[9]);
    for (int j = 6; j < 10; ++j) ic[j] /= n;
}

/*
 * Rocket6DOF.step (env:201-231) for n envs, array-of-structs float64 state.
 *   state (n,14) in/out; epconst (n,2) in/out = m0, v0; action (n,3) f32; obs (n,14) f32; reward (n) f64;
 *   done (n) u8; flags (n) u8 (1 ground, 2 bounds, 4 truncated); terms (n,7) f64 or NULL;
 *   ep_step (n) u32 / ep_return (n) f64 in/out or NULL; final_obs (n,14) f32 or NULL;
 *   fin_return (n) f64 / fin_length (n) u32 or NULL: return and length of the episode that ended in this step
 *   (what gym's Monitor reports, main_6DOF.py:51), written for done envs only.
 */
void oc_env_step(const OcParams *p, double *state, double *epconst, const float *action, float *obs, double *reward,
                 uint8_t *done, uint8_t *flags, double *terms, uint32_t *ep_step, double *ep_return, float *final_obs,
                 double *fin_return, uint32_t *fin_length, uint64_t seed, uint64_t step_index, uint64_t env_offset, int64_t n) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double *y = state + i * NS;
        const float *a = action + i * 3;
        float u[3] = {(float)((double)a[0] * p->max_gimbal), (float)((double)a[1] * p->max_gimbal),
                      (a[2] + 1.0f) / 2.0f * (float)p->max_thrust}; /* env:509-521 */
        int status = oc_sim_step(y, u, epconst[2 * i], p->dt, p->integrator, p->n_substeps);
        float s32[NS];
        for (int j = 0; j < NS; ++j) s32[j] = (float)y[j]; /* env:206 */
        int bnd = bounds_violation(s32, p), landed = 0;
        double t7[7];
        reward_terms(s32, u, y + 6, p, epconst[2 * i + 1], t7, &landed);
        double r = ((((((t7[0] + t7[1]) + t7[2]) + t7[3]) + t7[4]) + t7[5]) + t7[6]) + (bnd ? -50.0 : 0.0); /* env:228-229 */
        if (p->clip_reward) r = fmin(fmax(r, p->clip_lo), p->clip_hi);
        int dn = status || bnd, trunc = 0;
        if (ep_step) {
            ep_step[i] += 1;
            ep_return[i] += r;
            if (p->max_episode_steps > 0 && ep_step[i] >= (uint32_t)p->max_episode_steps) { trunc = !dn; dn = 1; }
        }
        reward[i] = r;
        done[i] = (uint8_t)dn;
        if (flags) flags[i] = (uint8_t)((status ? 1 : 0) | (bnd ? 2 : 0) | (trunc ? 4 : 0));
        if (terms) memcpy(terms + 7 * i, t7, sizeof(t7));
        for (int j = 0; j < NS; ++j) obs[i * NS + j] = (float)(y[j] / p->norm[j]); /* env:503-504 */
        if (dn && ep_step) {
            if (fin_return) fin_return[i] = ep_return[i];
            if (fin_length) fin_length[i] = ep_step[i];
        }
        if (dn && p->auto_reset) {
            if (final_obs) memcpy(final_obs + i * NS, obs + i * NS, NS * sizeof(float));
            float ic[NS];
            oc_reset_sample(p, seed, env_offset + (uint64_t)i, step_index, ic);
            for (int j = 0; j < NS; ++j) { y[j] = ic[j]; obs[i * NS + j] = (float)((double)ic[j] / p->norm[j]); }
            epconst[2 * i] = ic[13];
            epconst[2 * i + 1] = sqrt((double)ic[3] * ic[3] + (double)ic[4] * ic[4] + (double)ic[5] * ic[5]);
            if (ep_step) { ep_step[i] = 0; ep_return[i] = 0; }
        }
    }
}

int oc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
size_t oc_params_size(void) { return sizeof(OcParams); }
