// Fused actor-critic forward for the rollout loop around Rocket6DOF.step.
//
// What it replaces: SB3 1.6 `ActorCriticPolicy.forward(obs)` as called once per env step by
// `OnPolicyAlgorithm.collect_rollouts` (the reference trains with `PPO("MlpPolicy", env, policy_kwargs=
// {'net_arch': [128, 64]})`, main_6DOF.py:107-139, and evaluates with `model.predict`,
// montecarlo_script.py:33-51): two separate tanh MLPs (policy and value), a state-independent log_std,
// a diagonal Gaussian sample, its log-probability and the action clipped to the Box for `env.step`.
//
// One launch: obs (n, od) -> clipped action (n,3), raw sample (n,3), value [n], log-prob [n].
// In torch this is 6 skinny GEMMs + ~15 elementwise kernels whose (n, 64) intermediates go through HBM
// (~1 ms at n = 2^20); here the hidden activations never leave registers.
//
// Arithmetic: fp32 to ~1e-6.  The GEMMs run on the matrix cores as v_mfma_f32_32x32x16_f16 with every fp32
// operand split into two halves-precision terms, x = hi + lo (22 significant bits), and three products per
// K-step, hi*hi + hi*lo + lo*hi, accumulated in fp32.  (The exact-fp32 instruction v_mfma_f32_32x32x2_f32
// was the first version: it runs at 2x the VALU rate but shares the VALU's issue -- a wave's tanh cannot run
// under another wave's fp32 MFMAs, measured in scripts/mfma_valu_overlap.hip -- and took 256 us at 2^20.)
//   out^T[feature][env] = W[feature][k] * act^T[k][env]         (A = weights, B = activations)
// so that a layer's 32x32 accumulator block -- its column (env) on the lane, its rows (features) in the 16
// registers -- IS the B operand of the next layer with no LDS round trip and no lane movement: accumulator
// register r of lane half h holds feature row(r,h) = (r&3) + 8(r>>2) + 4h, and the B fragment of K-step s
// (16 features) is registers 8s..8s+7 converted in place; the A operand (weights, staged once per workgroup
// in LDS in exactly that order, already split) supplies W[out][32b + row(8s+j, h)] as element j of lane half
// h.  The last layer (4 outputs per env) is 128 VALU FMAs per lane plus one cross-half add -- a 32-row MFMA
// block would be 7/8 padding.
//
// Tile = 32 envs per wave (one MFMA column block); weights live in LDS for the whole (persistent) launch.
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

#ifndef R6P_BLOCK64
#define R6P_BLOCK64 512
#define R6P_WAVES64 4
#endif
#ifndef R6P_BLOCK128
#define R6P_BLOCK128 768
#define R6P_WAVES128 3
#endif

namespace r6p {

// Workgroup size per net shape: the LDS weight image is shared by the whole workgroup, so the wider net
// (83 KB image, one workgroup per CU) runs 768 threads to keep 3 waves per SIMD like the (64,64) net
// (43 KB image, three 256-thread workgroups per CU).  3 waves/SIMD = a 168-register budget: enough for
// two layers of accumulators, and what lets one wave's tanh run under another wave's MFMAs.
template <int H1>
struct Shape {
    static constexpr int kBlock = (H1 > 64) ? R6P_BLOCK128 : R6P_BLOCK64;
    static constexpr int kWaves = kBlock / 64;
};
template <int H1, int H2, int OD>
__global__ __launch_bounds__(Shape<H1>::kBlock, (H1 > 64) ? R6P_WAVES128 : R6P_WAVES64) void policy_act_kernel(const R6PolicyNet net, const float* __restrict__ obs, const int64_t n,
                                                            float* __restrict__ action, float* __restrict__ raw_action,
                                                            float* __restrict__ value, float* __restrict__ logp,
                                                            const uint64_t seed, const uint64_t step_index_arg,
                                                            const uint64_t* __restrict__ step_base,
                                                            const uint64_t env_offset, const int deterministic,
                                                            const uint8_t* __restrict__ row_flags, const int flag_mask) {
    const uint64_t step_index = step_index_arg + (step_base ? *step_base : (uint64_t)0);
    using Im [[maybe_unused]] = Image<H1, H2>;
    constexpr int kBlock = Shape<H1>::kBlock, kWaves = Shape<H1>::kWaves;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int col = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int kBiasK = OD;   // the obs column that is set to 1 and multiplies b1

    // masked evaluation (row_flags): a workgroup none of whose tiles holds a wanted row leaves before it stages the 43 KB weight
    // image -- V(terminal observation) is asked for after EVERY rollout step and answered for a handful of steps per episode
    if (row_flags) {
        const int64_t n_tiles0 = (n + 31) >> 5;
        bool want = false;
        for (int64_t t = (int64_t)blockIdx.x * kWaves + wave; t < n_tiles0; t += (int64_t)gridDim.x * kWaves) {
            const int64_t env = t * 32 + col;
            want = want || (env < n && (row_flags[env] & flag_mask) != 0);
        }
        if (!__syncthreads_or(want)) return;
    }
    stage_policy_image<H1, H2, OD>(net, lds, tid, kBlock);
    __syncthreads();
    const HeadConsts hc = load_head_consts<H2>(net);

    const int64_t n_tiles = (n + 31) >> 5;
    const int64_t nw = (int64_t)gridDim.x * kWaves;
    // B operand of layer 1: this lane's 8 obs features (k = s + 8h) of env t*32 + col, bias column = 1.
    // The next tile's features are requested before the current tile is computed (8 registers), so the
    // ~2 us of load latency sit under ~10 k cycles of matrix-core work instead of in front of them.
    auto load_x = [&](int64_t tt, float (&xx)[8]) __attribute__((always_inline)) {
        const int64_t e = tt * 32 + col;
        const bool in = tt < n_tiles && e < n;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int k = s + 8 * h;
            xx[s] = 0.0f;
            if (k < OD) {
                if (in) xx[s] = obs[e * OD + k];
            } else if (k == kBiasK) {
                xx[s] = 1.0f;
            }
        }
    };
    float xn[8];
    load_x((int64_t)blockIdx.x * kWaves + wave, xn);
#pragma unroll 1
    for (int64_t t = (int64_t)blockIdx.x * kWaves + wave; t < n_tiles; t += nw) {
        const int64_t env = t * 32 + col;
        const bool live = env < n;
        const Split xs = split8(xn);
        load_x(t + nw, xn);
        if (row_flags) {   // masked evaluation: a tile none of whose rows is wanted is skipped, outputs untouched
            const bool want = live && (row_flags[env] & flag_mask) != 0;
            if (__ballot(want) == 0) continue;
        }

        float part[4];   // this lane half's share of (mean0, mean1, mean2, value)
        policy_tile_forward<H1, H2>(net, lds, xs, lane, h, part);
        const Heads hd = combine_halves(part, hc);   // the other lane half holds the other 16 rows of every block

        // ---- diagonal Gaussian (SB3 DiagGaussianDistribution): sample, log-prob, clip -----------------------
        if (h == 0 && live) {
            const Sample sm = gaussian_sample(hd, hc, deterministic != 0, seed, step_index, env_offset + (uint64_t)env);
            const float a0 = sm.a0, a1v = sm.a1, a2v = sm.a2, lp = sm.logp, vv = hd.v;
            if (raw_action) {
                raw_action[env * 3 + 0] = a0;
                raw_action[env * 3 + 1] = a1v;
                raw_action[env * 3 + 2] = a2v;
            }
            action[env * 3 + 0] = fminf(fmaxf(a0, -1.0f), 1.0f);
            action[env * 3 + 1] = fminf(fmaxf(a1v, -1.0f), 1.0f);
            action[env * 3 + 2] = fminf(fmaxf(a2v, -1.0f), 1.0f);
            if (value) value[env] = vv;
            if (logp) logp[env] = lp;
        }
    }
}

template <int H1, int H2, int OD>
int launch(const R6PolicyNet& net, const float* obs, int64_t n, float* action, float* raw_action, float* value, float* logp,
           uint64_t seed, uint64_t step_index, const uint64_t* step_base, uint64_t env_offset, int deterministic,
           const uint8_t* row_flags, int flag_mask, hipStream_t s) {
    auto kernel = policy_act_kernel<H1, H2, OD>;
    constexpr int kBlock = Shape<H1>::kBlock, kWaves = Shape<H1>::kWaves;
    constexpr size_t lds_bytes = sizeof(float) * Image<H1, H2>::kTotalFloats;
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    static std::atomic<bool> configured[16];   // the attribute is a property of (function, device)
    const int di = (dev >= 0 && dev < 16) ? dev : 0;
    if (!configured[di].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) {
            r6_internal_set_error("r6_policy_act: cannot reserve LDS for the weight image");
            return R6_E_BADARG;
        }
        configured[di].store(dev == di, std::memory_order_relaxed);   // devices beyond the table are configured every call
    }
    // queried once per (function, device): a launch recorded into a HIP graph should issue nothing but the launch
    static std::atomic<int> resident[16];   // (per_cu << 16) | cus, 0 = not known yet
    int packed = (dev == di) ? resident[di].load(std::memory_order_relaxed) : 0;
    if (packed == 0) {
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
        (void)hipGetLastError();
        packed = (per_cu << 16) | (cus & 0xFFFF);
        if (dev == di) resident[di].store(packed, std::memory_order_relaxed);
    }
    per_cu = packed >> 16;
    cus = packed & 0xFFFF;
    if (std::getenv("R6_DEBUG")) std::fprintf(stderr, "[r6_policy_act] H1=%d H2=%d block=%d lds=%zu B resident/CU=%d CUs=%d\n", H1, H2, kBlock, lds_bytes, per_cu, cus);
    const int64_t n_tiles = (n + 31) / 32;
    int64_t blocks = (n_tiles + kWaves - 1) / kWaves;
    if (blocks > (int64_t)per_cu * cus) blocks = (int64_t)per_cu * cus;
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kBlock), lds_bytes, s, net, obs, n, action, raw_action, value, logp,
                       seed, step_index, step_base, env_offset, deterministic, row_flags, flag_mask);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}

}  // namespace r6p

extern "C" int r6_policy_act(const R6PolicyNet* net, const float* obs, int64_t n, float* action, float* raw_action,
                             float* value, float* logp, uint64_t seed, uint64_t step_index, const uint64_t* step_base,
                             uint64_t env_offset, int deterministic, const uint8_t* row_flags, int flag_mask, void* stream) {
    if (!net || !obs || !action) {
        r6_internal_set_error("r6_policy_act: net, obs and action are required");
        return R6_E_BADARG;
    }
    if (n < 0 || n >= ((int64_t)1 << 31)) {
        r6_internal_set_error("r6_policy_act: need 0 <= n < 2^31");
        return R6_E_BADARG;
    }
    if (n == 0) return 0;
    const R6PolicyNet& p = *net;
    if (!p.pi_w1 || !p.pi_b1 || !p.pi_w2 || !p.pi_b2 || !p.pi_w3 || !p.pi_b3 || !p.vf_w3 || !p.vf_b3 || !p.log_std) {
        r6_internal_set_error("r6_policy_act: trunk / head weight pointers of R6PolicyNet are required");
        return R6_E_BADARG;
    }
    if (!p.shared && (!p.vf_w1 || !p.vf_b1 || !p.vf_w2 || !p.vf_b2)) {
        r6_internal_set_error("r6_policy_act: a net with separate MLPs (shared == 0) needs vf_w1, vf_b1, vf_w2, vf_b2");
        return R6_E_BADARG;
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
#define R6P_CASE(A, B, D) \
    if (p.hidden1 == A && p.hidden2 == B && p.obs_dim == D) \
        return r6p::launch<A, B, D>(p, obs, n, action, raw_action, value, logp, seed, step_index, step_base, env_offset, deterministic, \
                                    row_flags, flag_mask, s);
    R6P_CASE(64, 64, 14)
    R6P_CASE(64, 64, 13)
    R6P_CASE(128, 64, 14)
    R6P_CASE(128, 64, 13)
#undef R6P_CASE
    r6_internal_set_error("r6_policy_act: supported nets are obs_dim 13|14 with hidden (64,64) or (128,64)");
    return R6_E_BADARG;
}

// -------------------------------------------------------------------------------------------------
// PPO minibatch loss and its gradient with respect to the policy outputs, one launch.
//
// SB3 1.6 `PPO.train` per minibatch (stable_baselines3/ppo/ppo.py; the reference trains through it,
// main_6DOF.py:107-139):  ratio = exp(logp - old_logp);  pg = -mean(min(A ratio, A clip(ratio, 1-c, 1+c)));
// v = mse(returns, values);  ent = -mean(entropy);  loss = pg + ent_coef ent + vf_coef v,  A normalised over the
// minibatch when `adv_norm` is given ([mean, 1/(std + 1e-8)], device).  In torch this is ~60 elementwise / reduction
// kernels forward and backward; here each sample is read once and the gradients that the MLPs' backward needs come
// out directly:
//   grad_mean (n,3), grad_value [n]   d loss / d (action mean, value) of every sample
//   grad_log_std [3]                  accumulated (atomicAdd) -- zero it before the launch
//   stats [8]                         accumulated sums: pg, v, entropy, clip_frac, approx_kl (each already / n), n
// torch.minimum's tie rule (ratio inside the clip range: both branches equal, gradient flows) is what the
// `active` mask below reproduces.
// -------------------------------------------------------------------------------------------------
namespace r6p {

__global__ __launch_bounds__(256) void ppo_loss_grad_kernel(const float* __restrict__ mean, const float* __restrict__ value,
                                                            const float* __restrict__ log_std, const float* __restrict__ action,
                                                            const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                            const float* __restrict__ ret, const float* __restrict__ adv_norm,
                                                            const int64_t n, const float clip, const float vf_coef, const float ent_coef,
                                                            float* __restrict__ grad_mean, float* __restrict__ grad_value,
                                                            float* __restrict__ grad_log_std, float* __restrict__ stats) {
    __shared__ float red[8];
    if (threadIdx.x < 8) red[threadIdx.x] = 0.0f;
    __syncthreads();
    const float ls0 = log_std[0], ls1 = log_std[1], ls2 = log_std[2];
    const float iv0 = __expf(-2.0f * ls0), iv1 = __expf(-2.0f * ls1), iv2 = __expf(-2.0f * ls2);
    const float a_mean = adv_norm ? adv_norm[0] : 0.0f, a_istd = adv_norm ? adv_norm[1] : 1.0f;
    const float inv_n = 1.0f / (float)n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // pg, v, clip_frac, kl, dls0, dls1, dls2
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float d0 = action[3 * i + 0] - mean[3 * i + 0], d1 = action[3 * i + 1] - mean[3 * i + 1], d2 = action[3 * i + 2] - mean[3 * i + 2];
        const float q0 = d0 * d0 * iv0, q1 = d1 * d1 * iv1, q2 = d2 * d2 * iv2;
        const float logp = -0.5f * (q0 + q1 + q2) - (ls0 + ls1 + ls2) - 2.756815599614018f;
        const float dl = logp - old_logp[i];
        const float ratio = __expf(dl);
        const float A = (adv[i] - a_mean) * a_istd;
        const float s1 = A * ratio, s2 = A * fminf(fmaxf(ratio, 1.0f - clip), 1.0f + clip);
        acc[0] += -fminf(s1, s2);
        const bool active = !((A > 0.0f && ratio > 1.0f + clip) || (A < 0.0f && ratio < 1.0f - clip));
        const float g_logp = active ? -s1 * inv_n : 0.0f;          // d pg / d logp
        grad_mean[3 * i + 0] = g_logp * d0 * iv0;
        grad_mean[3 * i + 1] = g_logp * d1 * iv1;
        grad_mean[3 * i + 2] = g_logp * d2 * iv2;
        acc[4] += g_logp * (q0 - 1.0f);
        acc[5] += g_logp * (q1 - 1.0f);
        acc[6] += g_logp * (q2 - 1.0f);
        const float ve = value[i] - ret[i];
        acc[1] += ve * ve;
        grad_value[i] = vf_coef * 2.0f * ve * inv_n;
        acc[2] += (fabsf(ratio - 1.0f) > clip) ? 1.0f : 0.0f;
        acc[3] += (ratio - 1.0f) - dl;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        float v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&red[k], v);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&stats[0], red[0] * inv_n);
        atomicAdd(&stats[1], red[1] * inv_n);
        atomicAdd(&stats[3], red[2] * inv_n);
        atomicAdd(&stats[4], red[3] * inv_n);
        // entropy is state-independent: ent = -(sum_i 0.5 + 0.5 ln(2 pi) + log_std_i); its gradient is -ent_coef per component
        atomicAdd(&grad_log_std[0], red[4]);
        atomicAdd(&grad_log_std[1], red[5]);
        atomicAdd(&grad_log_std[2], red[6]);
        if (blockIdx.x == 0) {
            atomicAdd(&stats[2], 3.0f * 1.4189385332046727f + ls0 + ls1 + ls2);
            atomicAdd(&stats[5], (float)n);
            atomicAdd(&grad_log_std[0], -ent_coef);
            atomicAdd(&grad_log_std[1], -ent_coef);
            atomicAdd(&grad_log_std[2], -ent_coef);
        }
    }
}

}  // namespace r6p

extern "C" int r6_ppo_loss_grad(const float* mean, const float* value, const float* log_std, const float* action,
                                const float* old_logp, const float* adv, const float* ret, const float* adv_norm, int64_t n,
                                float clip_range, float vf_coef, float ent_coef, float* grad_mean, float* grad_value,
                                float* grad_log_std, float* stats, void* stream) {
    if (!mean || !value || !log_std || !action || !old_logp || !adv || !ret || !grad_mean || !grad_value || !grad_log_std || !stats) {
        r6_internal_set_error("r6_ppo_loss_grad: all pointers except adv_norm are required");
        return R6_E_BADARG;
    }
    if (n <= 0 || n >= ((int64_t)1 << 31)) {
        r6_internal_set_error("r6_ppo_loss_grad: need 0 < n < 2^31");
        return R6_E_BADARG;
    }
    int64_t blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(r6p::ppo_loss_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mean, value,
                       log_std, action, old_logp, adv, ret, adv_norm, n, clip_range, vf_coef, ent_coef, grad_mean, grad_value,
                       grad_log_std, stats);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        r6_internal_set_error(hipGetErrorString(e));
        return -(int)e;
    }
    return 0;
}
