// r6_rollout.hip -- K closed-loop rollout steps in ONE launch: policy forward + Rocket6DOF.step per wave, K times.
//
// What it replaces: the per-timestep loop of SB3's OnPolicyAlgorithm.collect_rollouts behind the reference's
// model.learn() (main_6DOF.py:136-139): per step `policy.forward(obs)` -> clip -> `env.step(action)` -> (with a TimeLimit)
// `policy.predict_values(terminal_observation)` for the truncated envs.  As launches that is r6_policy_act, r6_step and a masked
// r6_policy_act per step -- 3 K launches whose only traffic between them is the (n, 13) observation and the (n, 3) action.  Here a
// wave keeps the state of its 32 envs in registers for the K steps: the observation never leaves the CU between the env step
// and the forward that consumes it, the action never leaves the wave between the forward and the env step.
//
// Results are BIT FOR BIT those of the launch sequence (tests/test_gpu_rollout_fused.py): the forward is the same device code as
// r6::policy_act_kernel (r6_policy_device.hpp: f16 hi/lo-split MFMAs, Philox + Box-Muller noise keyed by (seed, step, global
// env)), the env step the same device code as r6::step_kernel / step_k_kernel (r6_device.hpp), the in-kernel reset the same
// Philox draw (sample_ic_wave).
//
// Shape: tile = 32 envs per wave -- one column block of v_mfma_f32_32x32x16_f16, env = lane & 31.  The two lane halves of the
// matrix layout hold the two K-halves of a layer's features for the SAME env; the env step runs in both halves (the same env
// twice: with at most a tile or two per SIMD, what counts is the length of the instruction stream, not its lane efficiency), so
// that each half has the whole observation in registers and picks its 8 features of the B operand with selects -- no LDS round
// trip, no cross-lane traffic between the env step and the forward.  Observations leave as dword stores straight from registers
// (each half stores its 8 / 5..6 features).  Weights are staged once per workgroup in LDS (r6_policy.hip's image).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../../include/r6dof.h"
#include "r6_device.hpp"
#include "r6_policy_device.hpp"

extern "C" __attribute__((visibility("hidden"))) void r6_internal_set_error(const char* what);
extern "C" __attribute__((visibility("hidden"))) int r6_internal_device_error_guard(const char* who, void* stream);   // r6_kernels.hip: the device error word

namespace r6r {

using namespace r6;
using namespace r6p;

constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

// Episode statistics: exact fixed-point sums (the convention of r6_kernels.hip: counts and lengths as integers, the return and
// its square in units of 2^-16), so that a rollout's totals equal those of the launch sequence whatever the launch shape.
typedef unsigned long long stat_t;
__device__ __forceinline__ void stats_add_episode(stat_t* blk, float ep_ret, uint32_t ep_step, bool ground, bool bnd, bool truncated, bool landed) {
    const float clamped = fminf(fmaxf(ep_ret, -32767.0f), 32767.0f);
    const int q = __float2int_rn(clamped * 65536.0f);
    const long long q2 = ((long long)q * (long long)q) >> 16;
    atomicAdd(&blk[R6_STAT_EPISODES], (stat_t)1);
    atomicAdd(&blk[R6_STAT_RETURN], (stat_t)(long long)q);
    atomicAdd(&blk[R6_STAT_RETURN_SQ], (stat_t)q2);
    atomicAdd(&blk[R6_STAT_LENGTH], (stat_t)ep_step);
    if (ground) atomicAdd(&blk[R6_STAT_GROUND], (stat_t)1);
    if (bnd) atomicAdd(&blk[R6_STAT_BOUNDS], (stat_t)1);
    if (truncated) atomicAdd(&blk[R6_STAT_TRUNCATED], (stat_t)1);
    if (landed) atomicAdd(&blk[R6_STAT_LANDED], (stat_t)1);
}

template <int H1, int H2, int OD, int INTEG, int SHAPING>
__global__ __launch_bounds__(kBlock, 1) void rollout_kernel(const R6Params p, const R6Buffers b, const R6PolicyNet net, const R6Rollout rk,
                                                            const uint64_t env_seed, const uint64_t env_step_index, const uint64_t pol_seed,
                                                            const uint64_t pol_step_index, const uint64_t* __restrict__ pol_step_base,
                                                            const uint64_t env_offset) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ stat_t blk_stats[R6_NUM_STATS];
    constexpr int kBiasK = OD;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n = b.n, ld = b.ld;
    const int K = rk.k;
    const bool has_ep = b.ep_step != nullptr && b.ep_return != nullptr;

    if (tid < R6_NUM_STATS) blk_stats[tid] = 0ull;
    stage_policy_image<H1, H2, OD>(net, lds, tid, kBlock);
    __syncthreads();
    const HeadConsts hc = load_head_consts<H2>(net);
    const uint64_t env_base = b.step_base ? *b.step_base : (uint64_t)0;      // device-resident parts of the two step counters
    const uint64_t pol_base = pol_step_base ? *pol_step_base : (uint64_t)0;  // (graph replays), constant during the launch

    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t nw = (int64_t)gridDim.x * kWaves;
#pragma unroll 1
    for (int64_t t = (int64_t)blockIdx.x * kWaves + wave; t < n_tiles; t += nw) {
        const int64_t e0 = t * 32;
        const int64_t env = e0 + col;
        const bool live = env < n;
        const bool writer = live && h == 0;      // one of the two lanes of an env owns its per-env stores

        // ---- state of the tile: read once (both lane halves hold the same env) --------------------------------------------
        float y[kDyn], mass = 4.0e4f, m0 = 4.0e4f, v0 = 0.0f, ep_ret = 0.0f;
        uint32_t ep_step = 0u;
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = (j == 0) ? 1000.0f : ((j == 6) ? 1.0f : 0.0f);   // benign padding lane
        if (live) {
#pragma unroll
            for (int j = 0; j < kDyn; ++j) y[j] = b.state[j * ld + env];
            mass = b.state[13 * ld + env];
            m0 = b.epconst[env];
            v0 = b.epconst[ld + env];
            if (has_ep) {
                ep_step = b.ep_step[env];
                ep_ret = b.ep_return[env];
            }
        }

        // features k = s + 8 h of the observation (state / normaliser, env:503-504) of this lane's env: the B operand of
        // layer 1, bias column = 1 (r6_policy.hip)
        auto features = [&](const float (&yy)[kDyn], float mm, float (&x)[8]) __attribute__((always_inline)) {
            float ob[16];
#pragma unroll
            for (int q = 0; q < kDyn; ++q) ob[q] = yy[q] * p.inv_norm[q];
            ob[13] = mm * p.inv_norm[13];
            ob[14] = ob[15] = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (q == kBiasK) ob[q] = 1.0f;
                else if (q > kBiasK) ob[q] = 0.0f;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) x[s] = h ? ob[8 + s] : ob[s];
        };
        // obs row `row` of a (rows, n, OD) array <- the features in x (each lane half stores its own features k < OD)
        auto store_obs = [&](float* base, int64_t row, const float (&x)[8]) __attribute__((always_inline)) {
            if (live) {
                float* dst = base + (row * n + env) * OD + 8 * h;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s + 8 * h < OD) dst[s] = x[s];
                }
            }
        };

        float x[8];
        features(y, mass, x);

#pragma unroll 1
        for (int j = 0; j < K; ++j) {
            // ---- policy.forward(obs) (SB3 ActorCriticPolicy.forward) + the clip to the Box of collect_rollouts -----------
            float part[4];
            policy_tile_forward<H1, H2>(net, lds, split8(x), lane, h, part);
            const Heads hd = combine_halves(part, hc);
            const Sample sm = gaussian_sample(hd, hc, rk.deterministic != 0, pol_seed, pol_step_index + (uint64_t)j + pol_base,
                                              env_offset + (uint64_t)env);
            const float a0 = fminf(fmaxf(sm.a0, -1.0f), 1.0f);
            const float a1 = fminf(fmaxf(sm.a1, -1.0f), 1.0f);
            const float a2 = fminf(fmaxf(sm.a2, -1.0f), 1.0f);
            if (writer) {
                const int64_t o = (int64_t)j * n + env;
                if (rk.raw_action) {
                    rk.raw_action[o * 3 + 0] = sm.a0;
                    rk.raw_action[o * 3 + 1] = sm.a1;
                    rk.raw_action[o * 3 + 2] = sm.a2;
                }
                if (rk.value) rk.value[o] = hd.v;
                if (rk.logp) rk.logp[o] = sm.logp;
            }

            // ---- Rocket6DOF.step (env:201-231): the sequence of r6::step_kernel ------------------------------------------
            if (!has_ep) {   // no episode accounting: every step starts from zeroed counters, as r6_step without ep_step / ep_return does
                ep_step = 0u;
                ep_ret = 0.0f;
            }
            const Control c = make_control_normalised(p, a0, a1, a2);
            const Inertia J = make_inertia(m0);
            const int status = integrate<INTEG, true, true>(y, mass, c, J, p.dt, 1);
            const bool bnd = bounds_violation(p, y[0], y[1], y[2]);
            bool done = (status != 0) || bnd, truncated = false;
            RewardOut rw = reward<SHAPING>(p, y, mass, c, v0);
            const bool landed = rw.landed;
            float rew = finalize_reward(p, rw, bnd, fmaf(2.0f * c.thrust, frcp(p.max_thrust), -1.0f), y);
            if (p.clip_reward) rew = fminf(fmaxf(rew, p.clip_lo), p.clip_hi);
            ep_step += 1u;
            ep_ret += rew;
            if (p.max_episode_steps > 0 && ep_step >= (uint32_t)p.max_episode_steps) {
                truncated = !done;
                done = true;
            }
            if (writer) {
                const int64_t o = (int64_t)j * n + env;
                rk.reward[o] = rew;
                rk.done[o] = done ? 1 : 0;
                if (rk.flags)
                    rk.flags[o] = (uint8_t)((status == 1 ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) | (truncated ? R6_FLAG_TRUNCATED : 0));
            }
            features(y, mass, x);   // the step's observation: the terminal observation of an env that finished

            // ---- finished episodes: statistics, terminal outputs, V(terminal observation) of truncated envs ----------------
            if (writer && done) {
                if (b.stats) stats_add_episode(blk_stats, ep_ret, ep_step, status == 1, bnd, truncated, landed);
                if (p.auto_reset) {   // terminal outputs exist under auto-reset only, as in r6_step / r6_step_k
                    if (b.fin_return) b.fin_return[env] = ep_ret;
                    if (b.fin_length) b.fin_length[env] = ep_step;
                }
            }
            if (p.auto_reset && b.final_obs && live && done) {
                float* dst = b.final_obs + env * OD + 8 * h;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s + 8 * h < OD) dst[s] = x[s];
                }
            }
            if (rk.boot_value && __any(live && truncated)) {
                // SB3's time-limit bootstrap: reward += gamma V(terminal observation) for a truncated env (r6_gae reads it)
                float pb[4];
                policy_tile_forward<H1, H2>(net, lds, split8(x), lane, h, pb);
                const Heads hb = combine_halves(pb, hc);
                if (writer && truncated) rk.boot_value[(int64_t)j * n + env] = hb.v;
            }

            // ---- Rocket6DOF.reset (env:180-199) for the finished envs, one env at a time with the whole wave on its draw -----
            if (p.auto_reset) {
                uint64_t fin = __ballot(live && done) & 0xFFFFFFFFull;     // lanes 0..31: one per env
                if (fin != 0ull) {
                    do {
                        const int jj = __builtin_ctzll(fin);
                        fin &= fin - 1ull;
                        float ic[R6_STATE_DIM];
                        sample_ic_wave(p, env_seed, env_offset + (uint64_t)(e0 + jj), env_step_index + (uint64_t)j + env_base, lane, ic);
                        if (col == jj) {
#pragma unroll
                            for (int q = 0; q < kDyn; ++q) y[q] = ic[q];
                            mass = m0 = ic[13];
                            v0 = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
                            ep_step = 0u;
                            ep_ret = 0.0f;
                        }
                    } while (fin != 0ull);
                    features(y, mass, x);   // the fresh episodes' first observation
                }
            }
            // ---- the observation the next action is computed from: obs_seq row j + 1 (and the env's own obs buffer at the end) --
            if (rk.obs_seq) store_obs(rk.obs_seq, (int64_t)j + 1, x);
            if (b.obs && j + 1 == K) store_obs(b.obs, 0, x);
        }

        // ---- state of the tile: written once ------------------------------------------------------------------------------
        if (writer) {
#pragma unroll
            for (int j = 0; j < kDyn; ++j) b.state[j * ld + env] = y[j];
            b.state[13 * ld + env] = mass;
            b.epconst[env] = m0;
            b.epconst[ld + env] = v0;
            if (has_ep) {
                b.ep_step[env] = ep_step;
                b.ep_return[env] = ep_ret;
            }
        }
    }

    __syncthreads();
    if (b.stats && tid < R6_NUM_STATS && blk_stats[R6_STAT_EPISODES] != 0ull) {
        const double v = (double)(long long)blk_stats[tid];
        b.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += (tid == R6_STAT_RETURN || tid == R6_STAT_RETURN_SQ) ? v * (1.0 / 65536.0) : v;
    }
}

template <int H1, int H2, int OD, int INTEG, int SHAPING>
int launch(const R6Params& p, const R6Buffers& b, const R6PolicyNet& net, const R6Rollout& rk, uint64_t env_seed, uint64_t env_step,
           uint64_t pol_seed, uint64_t pol_step, const uint64_t* pol_base, uint64_t env_offset, hipStream_t s) {
    auto kernel = rollout_kernel<H1, H2, OD, INTEG, SHAPING>;
    constexpr size_t lds_bytes = sizeof(float) * Image<H1, H2>::kTotalFloats;
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    static std::atomic<int> resident[16];   // (per_cu << 16) | cus per device, 0 = not configured yet
    const int di = (dev >= 0 && dev < 16) ? dev : 0;
    int packed = (dev == di) ? resident[di].load(std::memory_order_relaxed) : 0;
    if (packed == 0) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            r6_internal_set_error("r6_rollout: cannot reserve LDS for the weight image");
            return R6_E_BADARG;
        }
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
        (void)hipGetLastError();
        packed = (per_cu << 16) | (cus & 0xFFFF);
        if (dev == di) resident[di].store(packed, std::memory_order_relaxed);
    }
    per_cu = packed >> 16;
    cus = packed & 0xFFFF;
    const int64_t n_tiles = (b.n + 31) / 32;
    int64_t blocks = (n_tiles + kWaves - 1) / kWaves;
    if (blocks > (int64_t)per_cu * cus) blocks = (int64_t)per_cu * cus;
    if (blocks > R6_MAX_BLOCKS) blocks = R6_MAX_BLOCKS;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, p, b, net, rk, env_seed, env_step, pol_seed, pol_step,
                       pol_base, env_offset);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

}  // namespace r6r

extern "C" size_t r6_rollout_size(void) { return sizeof(R6Rollout); }

extern "C" int r6_rollout(const R6Params* params, const R6Buffers* buf, const R6PolicyNet* net, const R6Rollout* ro, uint64_t env_seed,
                          uint64_t env_step_index, uint64_t policy_seed, uint64_t policy_step_index, const uint64_t* policy_step_base,
                          uint64_t env_offset, void* stream) {
    if (!params || !buf || !net || !ro) {
        r6_internal_set_error("r6_rollout: params, buffers, net and rollout are required");
        return R6_E_BADARG;
    }
    const R6Params& p = *params;
    const R6Buffers& b = *buf;
    const R6PolicyNet& pn = *net;
    const R6Rollout& rk = *ro;
    if (b.n < 0 || b.ld < b.n || b.n >= ((int64_t)1 << 31)) {
        r6_internal_set_error("r6_rollout: need 0 <= n <= ld, n < 2^31");
        return R6_E_BADARG;
    }
    if (rk.k < 1 || rk.k > 65536) {
        r6_internal_set_error("r6_rollout: k must be in 1..65536");
        return R6_E_BADARG;
    }
    if (b.n == 0) return 0;
    if (!b.state || !b.epconst || !rk.reward || !rk.done) {
        r6_internal_set_error("r6_rollout: state, epconst and the reward / done arrays are required");
        return R6_E_BADARG;
    }
    if ((p.integrator != R6_RK4 && p.integrator != R6_EULER) || p.n_substeps != 1 || !(p.dt > 0.0f)) {
        r6_internal_set_error("r6_rollout: supported integrators are euler / rk4 with one sub-step per env step");
        return R6_E_BADARG;
    }
    if ((p.obs_dim != 13 && p.obs_dim != 14) || pn.obs_dim != p.obs_dim) {
        r6_internal_set_error("r6_rollout: obs_dim must be 13 or 14 and equal for env and net");
        return R6_E_BADARG;
    }
    if (!pn.pi_w1 || !pn.pi_b1 || !pn.pi_w2 || !pn.pi_b2 || !pn.pi_w3 || !pn.pi_b3 || !pn.vf_w3 || !pn.vf_b3 || !pn.log_std ||
        (!pn.shared && (!pn.vf_w1 || !pn.vf_b1 || !pn.vf_w2 || !pn.vf_b2))) {
        r6_internal_set_error("r6_rollout: weight pointers of R6PolicyNet are required");
        return R6_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = r6_internal_device_error_guard("r6_rollout", stream)) return rc;
#define R6R_CASE(A, B, D, I, S) \
    if (pn.hidden1 == A && pn.hidden2 == B && p.obs_dim == D && p.integrator == I && p.shaping == S) \
        return r6r::launch<A, B, D, I, S>(p, b, pn, rk, env_seed, env_step_index, policy_seed, policy_step_index, policy_step_base, env_offset, s);
#define R6R_NETS(D, I, S) R6R_CASE(64, 64, D, I, S) R6R_CASE(128, 64, D, I, S)
    R6R_NETS(13, R6_RK4, R6_SHAPING_ACCELERATION)
    R6R_NETS(14, R6_RK4, R6_SHAPING_ACCELERATION)
    R6R_NETS(13, R6_EULER, R6_SHAPING_ACCELERATION)
    R6R_NETS(14, R6_EULER, R6_SHAPING_ACCELERATION)
    R6R_NETS(13, R6_RK4, R6_SHAPING_VELOCITY)
    R6R_NETS(14, R6_RK4, R6_SHAPING_VELOCITY)
#undef R6R_NETS
#undef R6R_CASE
    r6_internal_set_error("r6_rollout: supported: hidden (64,64) | (128,64), obs_dim 13 | 14, rk4 (either shaping) or euler (acceleration shaping)");
    return R6_E_BADARG;
}
