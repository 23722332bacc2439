// r6_kernels.hip -- HIP kernels (gfx950) + the C ABI of include/r6dof.h.
//
// Launch shape: one lane per environment, 256-lane workgroups (4 wave64), grid = ceil(n/256)
// (N = 2^20 -> 4096 workgroups >> 256 CUs).  State is structure-of-arrays, so every state access is
// a fully coalesced 256 B per wave-instruction.  The two array-of-structures edges -- actions
// (n,3) in, observations (n,13|14) out -- go through an LDS tile per workgroup so that global
// memory only ever sees contiguous 16 B-per-lane accesses.  No MFMA: there is no contraction here.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "r6_device.hpp"

namespace r6 {

constexpr int kBlock = R6_BLOCK;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// -------------------------------------------------------------------------------------------------
// Rocket6DOF.step (env:201-231) for one shard.
// -------------------------------------------------------------------------------------------------
template <int INTEG, int SHAPING, int OD>
__global__ __launch_bounds__(kBlock) void step_kernel(const R6Params p, const R6Buffers b, const uint64_t seed,
                                                      const uint64_t step_index, const uint64_t env_offset) {
    __shared__ __attribute__((aligned(16))) float tile[kBlock * OD];
    __shared__ float wave_stats[kBlock / 64][R6_NUM_STATS];

    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * kBlock;
    const int64_t i = base + tid;
    const int64_t ld = b.ld;
    const int64_t rem = b.n - base;
    const int nblk = rem < kBlock ? (int)rem : kBlock;  // live envs in this workgroup
    const bool live = tid < nblk;

    // ---- actions: AoS (n,3) -> LDS, 16 B per lane ------------------------------------------------
    {
        const float* src = b.action + base * R6_ACTION_DIM;
        if (nblk == kBlock) {
            if (tid < kBlock * R6_ACTION_DIM / 4) st4(&tile[4 * tid], ld4(src + 4 * tid));
        } else {
            for (int k = tid; k < nblk * R6_ACTION_DIM; k += kBlock) tile[k] = src[k];
        }
    }

    // ---- state: SoA rows, coalesced ---------------------------------------------------------------
    float y[kDyn];
    float mass, m0, v0 = 0.0f;
    if (live) {
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = b.state[j * ld + i];
        mass = b.state[13 * ld + i];
        m0 = b.epconst[i];
        if constexpr (SHAPING == R6_SHAPING_VELOCITY) v0 = b.epconst[ld + i];
    } else {  // benign padding lane
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = 0.0f;
        y[0] = 1000.0f;
        y[6] = 1.0f;
        mass = m0 = 4.0e4f;
    }
    uint32_t ep_step = 0;
    float ep_ret = 0.0f;
    if (live && b.ep_step) ep_step = b.ep_step[i];
    if (live && b.ep_return) ep_ret = b.ep_return[i];

    __syncthreads();
    float a0 = 0.0f, a1 = 0.0f, a2 = -1.0f;
    if (live) {
        a0 = tile[3 * tid + 0];
        a1 = tile[3 * tid + 1];
        a2 = tile[3 * tid + 2];
    }
    __syncthreads();  // tile is reused for the observations below

    // ---- dynamics (sim:69-143) --------------------------------------------------------------------
    const Control c = make_control_normalised(p, a0, a1, a2);
    const Inertia J = make_inertia(m0);
    const int status = integrate<INTEG>(y, mass, c, J, p.dt, p.n_substeps);

    // ---- termination (env:213, :591-593) ------------------------------------------------------------
    const bool bnd = bounds_violation(p, y[0], y[1], y[2]);
    bool done = (status != 0) || bnd;

    // ---- reward (env:215, :228-229, :317-402) -------------------------------------------------------
    const RewardOut rw = reward<SHAPING>(p, y, mass, c, v0);
    float rew = rw.sum + (bnd ? p.bounds_penalty : 0.0f);
    if (p.clip_reward) rew = fminf(fmaxf(rew, p.clip_lo), p.clip_hi);  // main_6DOF.py:33-42

    // ---- fused TimeLimit / Monitor (main_6DOF.py:47-51) ---------------------------------------------
    ep_step += 1u;
    ep_ret += rew;
    bool truncated = false;
    if (p.max_episode_steps > 0 && ep_step >= (uint32_t)p.max_episode_steps) {
        truncated = !done;
        done = true;
    }

    float obs[R6_STATE_DIM];
#pragma unroll
    for (int j = 0; j < kDyn; ++j) obs[j] = y[j] * p.inv_norm[j];
    obs[13] = mass * p.inv_norm[13];

    if (live) {
        b.reward[i] = rew;
        b.done[i] = done ? 1 : 0;
        if (b.flags)
            b.flags[i] = (uint8_t)((status ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) |
                                   (truncated ? R6_FLAG_TRUNCATED : 0));
        if (b.terms) {
#pragma unroll
            for (int t = 0; t < R6_NUM_TERMS; ++t) b.terms[t * ld + i] = rw.terms[t];
        }
    }

    // ---- per-workgroup episode statistics (deterministic tree sums, no atomics) -------------------------
    if (b.stats) {
        const bool fin = live && done;
        if (__any(fin)) {
            float sv[R6_NUM_STATS];
            sv[R6_STAT_EPISODES] = fin ? 1.0f : 0.0f;
            sv[R6_STAT_RETURN] = fin ? ep_ret : 0.0f;
            sv[R6_STAT_RETURN_SQ] = fin ? ep_ret * ep_ret : 0.0f;
            sv[R6_STAT_LENGTH] = fin ? (float)ep_step : 0.0f;
            sv[R6_STAT_GROUND] = (fin && status) ? 1.0f : 0.0f;
            sv[R6_STAT_BOUNDS] = (fin && bnd) ? 1.0f : 0.0f;
            sv[R6_STAT_TRUNCATED] = (fin && truncated) ? 1.0f : 0.0f;
            sv[R6_STAT_LANDED] = (fin && rw.landed) ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < R6_NUM_STATS; ++k) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) sv[k] += __shfl_xor(sv[k], o, 64);
            }
            if ((tid & 63) == 0) {
#pragma unroll
                for (int k = 0; k < R6_NUM_STATS; ++k) wave_stats[tid >> 6][k] = sv[k];
            }
        } else if ((tid & 63) == 0) {
#pragma unroll
            for (int k = 0; k < R6_NUM_STATS; ++k) wave_stats[tid >> 6][k] = 0.0f;
        }
    }

    // ---- in-kernel auto-reset (VecEnv semantics: obs of the NEW episode is returned, the terminal
    //      observation goes to final_obs) ------------------------------------------------------------
    if (live && done && p.auto_reset) {
        if (b.final_obs) {
#pragma unroll
            for (int j = 0; j < OD; ++j) b.final_obs[i * OD + j] = obs[j];
        }
        if (b.fin_return) b.fin_return[i] = ep_ret;
        if (b.fin_length) b.fin_length[i] = ep_step;
        float ic[R6_STATE_DIM];
        sample_ic(p, seed, env_offset + (uint64_t)i, step_index, ic);
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = ic[j];
        mass = ic[13];
        b.epconst[i] = ic[13];
        b.epconst[ld + i] = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
#pragma unroll
        for (int j = 0; j < R6_STATE_DIM; ++j) obs[j] = ic[j] * p.inv_norm[j];
        ep_step = 0u;
        ep_ret = 0.0f;
    }

    // ---- write back state (SoA) ----------------------------------------------------------------------
    if (live) {
#pragma unroll
        for (int j = 0; j < kDyn; ++j) b.state[j * ld + i] = y[j];
        b.state[13 * ld + i] = mass;
        if (b.ep_step) b.ep_step[i] = ep_step;
        if (b.ep_return) b.ep_return[i] = ep_ret;
    }

    // ---- observations: registers -> LDS (row-major tile) -> 16 B-per-lane contiguous stores ---------
#pragma unroll
    for (int j = 0; j < OD; ++j) tile[tid * OD + j] = obs[j];
    __syncthreads();
    if (b.stats && tid < R6_NUM_STATS) {
        const float add = (wave_stats[0][tid] + wave_stats[1][tid]) + (wave_stats[2][tid] + wave_stats[3][tid]);
        const float cnt = (wave_stats[0][0] + wave_stats[1][0]) + (wave_stats[2][0] + wave_stats[3][0]);
        if (cnt > 0.0f) b.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += add;
    }
    {
        float* dst = b.obs + base * OD;
        if (nblk == kBlock) {
            constexpr int n4 = kBlock * OD / 4;
#pragma unroll
            for (int k = 0; k < (n4 + kBlock - 1) / kBlock; ++k) {
                const int q = tid + k * kBlock;
                if (q < n4) st4(dst + 4 * q, ld4(&tile[4 * q]));
            }
        } else {
            for (int k = tid; k < nblk * OD; k += kBlock) dst[k] = tile[k];
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Rocket6DOF.reset (env:180-199) for masked envs.
// -------------------------------------------------------------------------------------------------
template <int OD>
__global__ __launch_bounds__(kBlock) void reset_kernel(const R6Params p, const R6Buffers b, const uint8_t* mask,
                                                       const uint64_t seed, const uint64_t step_index,
                                                       const uint64_t env_offset) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= b.n) return;
    if (mask && !mask[i]) return;
    const int64_t ld = b.ld;
    float ic[R6_STATE_DIM];
    sample_ic(p, seed, env_offset + (uint64_t)i, step_index, ic);
#pragma unroll
    for (int j = 0; j < R6_STATE_DIM; ++j) b.state[j * ld + i] = ic[j];
    b.epconst[i] = ic[13];
    b.epconst[ld + i] = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
    if (b.obs) {
#pragma unroll
        for (int j = 0; j < OD; ++j) b.obs[i * OD + j] = ic[j] * p.inv_norm[j];
    }
    if (b.ep_step) b.ep_step[i] = 0u;
    if (b.ep_return) b.ep_return[i] = 0.0f;
}

template <int OD>
__global__ __launch_bounds__(kBlock) void observe_kernel(const R6Params p, const R6Buffers b) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= b.n) return;
#pragma unroll
    for (int j = 0; j < OD; ++j) b.obs[i * OD + j] = b.state[j * b.ld + i] * p.inv_norm[j];
}

// -------------------------------------------------------------------------------------------------
// Simulator6DOF.step (sim:69-104) / RHS (sim:106-143) on physical controls.
// -------------------------------------------------------------------------------------------------
template <int INTEG>
__global__ __launch_bounds__(kBlock) void sim_step_kernel(const R6Params p, float* state, const float* epconst,
                                                          const float* u, uint8_t* status, int64_t n, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float y[kDyn];
#pragma unroll
    for (int j = 0; j < kDyn; ++j) y[j] = state[j * ld + i];
    float mass = state[13 * ld + i];
    const Control c = make_control_physical(u[3 * i + 0], u[3 * i + 1], u[3 * i + 2]);
    const Inertia J = make_inertia(epconst[i]);
    const int st = integrate<INTEG>(y, mass, c, J, p.dt, p.n_substeps);
#pragma unroll
    for (int j = 0; j < kDyn; ++j) state[j * ld + i] = y[j];
    state[13 * ld + i] = mass;
    if (status) status[i] = (uint8_t)st;
}

__global__ __launch_bounds__(kBlock) void rhs_kernel(const float* ys, const float* epconst, const float* u, float* dys,
                                                     int64_t n, int64_t ld) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float y[kDyn], dy[kDyn];
#pragma unroll
    for (int j = 0; j < kDyn; ++j) y[j] = ys[j * ld + i];
    const float mass = ys[13 * ld + i];
    const Control c = make_control_physical(u[3 * i + 0], u[3 * i + 1], u[3 * i + 2]);
    const Inertia J = make_inertia(epconst[i]);
    rhs13(y, mass, c, J, dy);
#pragma unroll
    for (int j = 0; j < kDyn; ++j) dys[j * ld + i] = dy[j];
    dys[13 * ld + i] = -c.mdot;
}

}  // namespace r6

// =================================================================================================
// C ABI
// =================================================================================================
namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* what) {
    std::snprintf(g_err, sizeof(g_err), "%s", what);
    return code;
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int validate_common(const R6Params* p, const R6Buffers* b) {
    if (!p || !b) return fail(R6_E_BADARG, "null params/buffers");
    if (b->n < 0 || b->ld < b->n) return fail(R6_E_BADARG, "need 0 <= n <= ld");
    if (p->obs_dim != 13 && p->obs_dim != 14) return fail(R6_E_BADARG, "obs_dim must be 13 or 14");
    if (p->integrator != R6_EULER && p->integrator != R6_RK4) return fail(R6_E_BADARG, "bad integrator");
    if (p->n_substeps < 1 || p->n_substeps > 4096) return fail(R6_E_BADARG, "n_substeps out of range");
    if (p->shaping != R6_SHAPING_ACCELERATION && p->shaping != R6_SHAPING_VELOCITY)
        return fail(R6_E_BADARG, "bad shaping");
    if (!(p->dt > 0.0f)) return fail(R6_E_BADARG, "dt must be > 0");
    return 0;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + r6::kBlock - 1) / r6::kBlock)); }

template <int INTEG, int SHAPING>
void launch_step_od(const R6Params& p, const R6Buffers& b, uint64_t seed, uint64_t step, uint64_t off, hipStream_t s) {
    if (p.obs_dim == 14)
        hipLaunchKernelGGL((r6::step_kernel<INTEG, SHAPING, 14>), grid_for(b.n), dim3(r6::kBlock), 0, s, p, b, seed, step, off);
    else
        hipLaunchKernelGGL((r6::step_kernel<INTEG, SHAPING, 13>), grid_for(b.n), dim3(r6::kBlock), 0, s, p, b, seed, step, off);
}

}  // namespace

extern "C" {

int r6_abi_version(void) { return R6_ABI_VERSION; }
size_t r6_params_size(void) { return sizeof(R6Params); }
size_t r6_buffers_size(void) { return sizeof(R6Buffers); }
const char* r6_last_error(void) { return g_err; }

int r6_step(const R6Params* params, const R6Buffers* buf, uint64_t seed, uint64_t step_index, uint64_t env_offset,
            void* stream) {
    if (int rc = validate_common(params, buf)) return rc;
    const R6Buffers& b = *buf;
    if (b.n == 0) return 0;
    if (!b.state || !b.epconst || !b.action || !b.obs || !b.reward || !b.done)
        return fail(R6_E_BADARG, "r6_step: state/epconst/action/obs/reward/done are required");
    if (!aligned16(b.action) || !aligned16(b.obs))
        return fail(R6_E_ALIGN, "r6_step: action and obs must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const R6Params& p = *params;
    if (p.integrator == R6_RK4) {
        if (p.shaping == R6_SHAPING_ACCELERATION)
            launch_step_od<R6_RK4, R6_SHAPING_ACCELERATION>(p, b, seed, step_index, env_offset, s);
        else
            launch_step_od<R6_RK4, R6_SHAPING_VELOCITY>(p, b, seed, step_index, env_offset, s);
    } else {
        if (p.shaping == R6_SHAPING_ACCELERATION)
            launch_step_od<R6_EULER, R6_SHAPING_ACCELERATION>(p, b, seed, step_index, env_offset, s);
        else
            launch_step_od<R6_EULER, R6_SHAPING_VELOCITY>(p, b, seed, step_index, env_offset, s);
    }
    return check_launch("r6_step");
}

int r6_reset(const R6Params* params, const R6Buffers* buf, const uint8_t* mask, uint64_t seed, uint64_t step_index,
             uint64_t env_offset, void* stream) {
    if (int rc = validate_common(params, buf)) return rc;
    const R6Buffers& b = *buf;
    if (b.n == 0) return 0;
    if (!b.state || !b.epconst) return fail(R6_E_BADARG, "r6_reset: state/epconst are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (params->obs_dim == 14)
        hipLaunchKernelGGL((r6::reset_kernel<14>), grid_for(b.n), dim3(r6::kBlock), 0, s, *params, b, mask, seed, step_index, env_offset);
    else
        hipLaunchKernelGGL((r6::reset_kernel<13>), grid_for(b.n), dim3(r6::kBlock), 0, s, *params, b, mask, seed, step_index, env_offset);
    return check_launch("r6_reset");
}

int r6_observe(const R6Params* params, const R6Buffers* buf, void* stream) {
    if (int rc = validate_common(params, buf)) return rc;
    const R6Buffers& b = *buf;
    if (b.n == 0) return 0;
    if (!b.state || !b.obs) return fail(R6_E_BADARG, "r6_observe: state/obs are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (params->obs_dim == 14)
        hipLaunchKernelGGL((r6::observe_kernel<14>), grid_for(b.n), dim3(r6::kBlock), 0, s, *params, b);
    else
        hipLaunchKernelGGL((r6::observe_kernel<13>), grid_for(b.n), dim3(r6::kBlock), 0, s, *params, b);
    return check_launch("r6_observe");
}

int r6_sim_step(const R6Params* params, float* state, const float* epconst, const float* u_phys, uint8_t* status,
                int64_t n, int64_t ld, void* stream) {
    if (!params || !state || !epconst || !u_phys) return fail(R6_E_BADARG, "r6_sim_step: null argument");
    if (n < 0 || ld < n) return fail(R6_E_BADARG, "need 0 <= n <= ld");
    if (params->n_substeps < 1 || !(params->dt > 0.0f)) return fail(R6_E_BADARG, "bad dt / n_substeps");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (params->integrator == R6_RK4)
        hipLaunchKernelGGL((r6::sim_step_kernel<R6_RK4>), grid_for(n), dim3(r6::kBlock), 0, s, *params, state, epconst, u_phys, status, n, ld);
    else if (params->integrator == R6_EULER)
        hipLaunchKernelGGL((r6::sim_step_kernel<R6_EULER>), grid_for(n), dim3(r6::kBlock), 0, s, *params, state, epconst, u_phys, status, n, ld);
    else
        return fail(R6_E_BADARG, "bad integrator");
    return check_launch("r6_sim_step");
}

int r6_rhs(const R6Params* params, const float* y, const float* epconst, const float* u_phys, float* dy, int64_t n,
           int64_t ld, void* stream) {
    (void)params;
    if (!y || !epconst || !u_phys || !dy) return fail(R6_E_BADARG, "r6_rhs: null argument");
    if (n < 0 || ld < n) return fail(R6_E_BADARG, "need 0 <= n <= ld");
    if (n == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(r6::rhs_kernel, grid_for(n), dim3(r6::kBlock), 0, s, y, epconst, u_phys, dy, n, ld);
    return check_launch("r6_rhs");
}

}  // extern "C"
