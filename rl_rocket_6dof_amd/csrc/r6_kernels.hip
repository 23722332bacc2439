// r6_kernels.hip -- HIP kernels (gfx950) + the C ABI of include/r6dof.h.
//
// Launch shape of the step: one lane per environment, 256-lane workgroups (4 wave64), a PERSISTENT grid sized to
// the resident capacity of the chip (occupancy x 256 CUs); every wave walks over 64-env tiles on its own, with
// no workgroup barrier in the tile loop.  State is structure-of-arrays, so every state access is a fully
// coalesced 256 B per wave-instruction with a wave-uniform SGPR base.  The two array-of-structures edges:
// observations (n,13|14) leave through a per-WAVE LDS tile as contiguous 16 B-per-lane stores; actions (n,3)
// are read directly at a 12 B lane stride (the wave touches the same 768 B three times, the repeats are L1
// hits; an LDS-staged variant was measured and is not faster).  No MFMA: there is no
// contraction here.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <type_traits>

#include "r6_device.hpp"

namespace r6 {

constexpr int kBlock = R6_BLOCK;

typedef float f4 __attribute__((ext_vector_type(4)));  // plain vector type: usable in any address space
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f4 ld4(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void st4(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }

// -------------------------------------------------------------------------------------------------
// Rocket6DOF.step (env:201-231) for one shard.
//
// Wave-autonomous tiles: the grid is sized to the resident capacity of the chip and every wave64
// walks over 64-env tiles (tile = wave_gid + k * total_waves).  There is no workgroup barrier inside
// the tile loop -- each wave owns a private LDS tile for the AoS observation edge -- so the waves of
// a CU drift apart and their load / compute / store phases overlap.  The hot loop is kept lean so
// that many waves are resident (occupancy is what hides HBM latency here):
//   * a ground event costs two evaluations of the force/torque half of the RHS (Hermite dense output of the
//     step, r6_device.hpp integrate<>), not a second RK4 step;
//   * finished episodes are not re-sampled in the divergent per-lane path: their env indices go to a
//     per-workgroup LDS queue that is drained densely once, after the loop (a full queue -- mass termination --
//     makes the excess lanes re-sample in the tile loop);
//   * kernel parameters are re-read from the kernarg segment per phase instead of living in ~110
//     SGPRs across the loop.
// -------------------------------------------------------------------------------------------------
constexpr int kWaves = kBlock / 64;
#ifndef R6_SWEEP_DEFAULT
#define R6_SWEEP_DEFAULT -1     // odd steps on the resident grid: 0 = as even steps, 1 = every wave walks its own list backwards, 2 = whole range mirrored, -1 = per integrator
#endif
#ifndef R6_ACT_LDS
#define R6_ACT_LDS 0            // A/B switch: action edge staged through LDS (1) or one 12-byte load per lane (0)
#endif
#ifndef R6_WG_DYNAMIC
#define R6_WG_DYNAMIC 1         // A/B switch (Euler instances): the waves of a workgroup pull the workgroup's tiles from an LDS counter (1) or own fixed lists (0)
#endif
#ifndef R6_RESET_QUEUE
#define R6_RESET_QUEUE 2048
#endif
#ifndef R6_RESET_QUAD
#define R6_RESET_QUAD 1         // A/B switch: the dense drain re-samples one env per quad of lanes (1) or one per lane (0)
#endif
#define R6_K_DIRECT_OBS_INTERNAL 0x4000     // R6StepK.flags bit set by the launcher (not part of the ABI): obs stored from registers
#ifndef R6_K_DIRECT_MAX_TILES
#define R6_K_DIRECT_MAX_TILES 1024          // r6_step_k: up to one 64-env tile per SIMD the direct form is used (65 536 envs: 3.47 -> 3.25 us per step at K = 16; at 262 144 envs it is 35 % slower than the transposing form)
#endif
#ifndef R6_EARLY_LOAD
#define R6_EARLY_LOAD 1         // A/B switch (RK4 / RK45 instances): rotated tile loop, first tile requested before the set-up barrier
#endif
constexpr int kResetQueueCap = R6_RESET_QUEUE;

__device__ __forceinline__ void wave_lds_sync() {
    // LDS instructions of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the point.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wave-uniform pointer made opaque to the optimiser: keeps `base + lane` from being folded into
// one divergent 64-bit index, so accesses are selected as  global_* v, v_lane_offset, s[base]  --
// one VGPR of addressing for all rows instead of a 64-bit VGPR pair per row.
template <typename T>
using gptr = __attribute__((address_space(1))) T*;

template <typename T>
__device__ __forceinline__ gptr<T> uniform_ptr(T* p) {
    asm volatile("" : "+s"(p));
    return (gptr<T>)p;  // global address space: selected as global_* with an SGPR base
}

// Streaming outputs: obs / reward / done are never read back by the step.  Once the working set (189 B/env) is far
// beyond the 256 MB Infinity Cache, storing them with the non-temporal hint keeps the cache for the state rows the
// next launch re-reads: A/B at 2^23 envs 334 -> 290 us, at 2^24 766 -> 659 us; 2^21 / 2^22 +-0; at 2^20 (everything
// fits) +0.3 us.  NT is a template parameter of the kernel (chosen by the launcher from n): as a launch-uniform
// branch the hint cost the wrapped variant 1.7 us of perturbed code even when off.  (The hint on the one-shot action
// read costs 5 us: its three 12 B-strided loads per lane live on L1 hits.)
#ifndef R6_OBS_BATCH
#define R6_OBS_BATCH 1
#endif
#ifndef R6_NT_MIN_ENVS
#define R6_NT_MIN_ENVS (3ll << 21)
#endif
#ifndef R6_NT_WINDOW_LO
#define R6_NT_WINDOW_LO 1200000ll
#define R6_NT_WINDOW_HI 2360000ll
#endif
template <bool NT, typename T, typename V>
__device__ __forceinline__ void stream_store(gptr<T> p, V v) {
    if constexpr (NT) __builtin_nontemporal_store((T)v, p);
    else *p = (T)v;
}

// The kernel parameters live in the kernarg segment.  Reading them through a pointer that is
// re-laundered per phase of the tile loop makes the compiler issue the scalar loads where the
// values are used (a handful of s_load_dwordxN per phase, scalar-cache hits) instead of hoisting all
// ~110 parameter dwords out of the loop and spilling SGPRs.  R6Params is kernel argument 0,
// R6Buffers kernel argument 1 right behind it (both are 8-byte multiples).
typedef const __attribute__((address_space(4))) char* kernarg_ptr_t;
static_assert(sizeof(R6Params) % 8 == 0, "R6Buffers must follow R6Params without padding");
__device__ __forceinline__ R6Params load_params() {
    kernarg_ptr_t k = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const R6Params*)k;
}
__device__ __forceinline__ R6Buffers load_buffers() {
    kernarg_ptr_t k = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const R6Buffers*)(k + sizeof(R6Params));
}

__device__ __forceinline__ R6StepK load_stepk() {   // kernel argument 2 of the r6_step_k kernels, right behind R6Buffers
    static_assert(sizeof(R6Buffers) % 8 == 0, "R6StepK must follow R6Buffers without padding");
    kernarg_ptr_t k = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return *(const R6StepK*)(k + sizeof(R6Params) + sizeof(R6Buffers));
}

// env:180-199 for one env, written straight to HBM (used by the dense drain of the reset queue).
template <int OD>
__device__ __forceinline__ void reset_env_global(const R6Params& p, const R6Buffers& b, int64_t i, uint64_t seed,
                                                 uint64_t step_index, uint64_t env_offset) {
    const int64_t ld = b.ld;
    float ic[R6_STATE_DIM];
    // R6Buffers.step_base: the device-resident part of the step counter (graph replays); read on this rare path only
    sample_ic(p, seed, env_offset + (uint64_t)i, step_index + (b.step_base ? *b.step_base : (uint64_t)0), ic);
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

// The same for one env per QUAD of lanes (round 5): lane b = lane & 3 of a quad runs Philox block b and turns its four words into the
// components 4b .. 4b + 3 of the initial condition, stores them, and the few values that need components of another lane (the
// quaternion's norm, the speed v0) are fetched within the quad.  A wave re-samples 16 envs in one pass of ~150 instructions; one env
// per LANE is four Philox blocks in a row, ~450 instructions of a single, latency-bound wave (the drain sits on the critical path of
// every workgroup, after the barrier).  The same operations on the same operands as sample_ic(): bit for bit the same state.
// `inv_norm`: the observation normaliser reciprocals in LDS (per-lane index).
template <int OD>
__device__ __forceinline__ void reset_env_quad(const R6Buffers& b, const float* inv_norm, int64_t i, bool valid, int lane, uint64_t seed,
                                               uint64_t step_index, uint64_t env_offset) {
    const uint64_t step = step_index + (b.step_base ? *b.step_base : (uint64_t)0);
    const uint64_t gid = env_offset + (uint64_t)i;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const uint32_t blk = (uint32_t)lane & 3u;
    const U4 r = philox4x32_10(U4{(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step, ((uint32_t)(step >> 32) & 0x3FFFFFFFu) | (blk << 30)}, k0, k1);
    // the distribution, per lane, from the kernarg segment (R6Params is kernel argument 0)
    const kernarg_ptr_t ka = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    const auto* lo = (const __attribute__((address_space(4))) float*)(ka + offsetof(R6Params, ic_lo));
    const auto* rg = (const __attribute__((address_space(4))) float*)(ka + offsetof(R6Params, ic_range));
    const int j0 = 4 * (int)blk;
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = (j0 + k < R6_STATE_DIM) ? j0 + k : 0;
        f[k] = fmaf((float)(w[k] >> 8) * 5.9604644775390625e-08f, rg[j], lo[j]);
    }
    const int q = lane & ~3;
    // quaternion (components 6, 7 in lane 1 of the quad, 8, 9 in lane 2): normalised with sample_ic()'s expression
    const float q6 = __shfl(f[2], q + 1), q7 = __shfl(f[3], q + 1), q8 = __shfl(f[0], q + 2), q9 = __shfl(f[1], q + 2);
    const float n2 = fmaf(q6, q6, fmaf(q7, q7, fmaf(q8, q8, q9 * q9)));
    const float inv = 1.0f / sqrtf(n2);
    if (blk == 1u) {
        f[2] *= inv;
        f[3] *= inv;
    } else if (blk == 2u) {
        f[0] *= inv;
        f[1] *= inv;
    }
    // speed of the initial condition (components 3 | 4, 5)
    const float c3 = __shfl(f[3], q), c4 = __shfl(f[0], q + 1), c5 = __shfl(f[1], q + 1);
    const float v0 = sqrtf(fmaf(c3, c3, fmaf(c4, c4, c5 * c5)));
    if (!valid) return;
    const int64_t ld = b.ld;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + k;
        if (j < R6_STATE_DIM) {
            b.state[j * ld + i] = f[k];
            if (b.obs && j < OD) b.obs[i * OD + j] = f[k] * inv_norm[j];
        }
    }
    if (blk == 3u) b.epconst[i] = f[1];          // m0 = the initial mass (component 13)
    if (blk == 0u) {
        b.epconst[ld + i] = v0;
        if (b.ep_step) b.ep_step[i] = 0u;
        if (b.ep_return) b.ep_return[i] = 0.0f;
    }
}

// Episode statistics of a workgroup: exact integer / fixed-point sums in LDS (native 64-bit integer atomics), so that the
// totals depend neither on the order in which lanes and waves arrive nor on the launch shape: counts and lengths as integers,
// the episode return in units of 2^-16 (|return| < 32768; its square in units of 2^-16 too, from the quantised return).
// A workgroup's sums go to its own float64 row of R6Buffers.stats (exact: integers * 2^-16 far below 2^53).
typedef unsigned long long stat_t;
constexpr float kStatScale = 65536.0f;
__device__ __forceinline__ void stats_add_episode(stat_t* blk, float ep_ret, uint32_t ep_step, bool ground, bool bnd, bool truncated, bool landed) {
    const float clamped = fminf(fmaxf(ep_ret, -32767.0f), 32767.0f);
    const int q = __float2int_rn(clamped * kStatScale);
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
__device__ __forceinline__ double stats_value(const stat_t* blk, int j) {
    const double v = (double)(long long)blk[j];
    return (j == R6_STAT_RETURN || j == R6_STAT_RETURN_SQ) ? v * (1.0 / 65536.0) : v;
}

// Inputs of one 64-env tile, one lane = one env.
struct TileIn {
    float y[kDyn];
    float mass, m0, v0, a0, a1, a2, ep_ret;
    uint32_t ep_step;
};

template <int SHAPING, bool FULL>
__device__ __forceinline__ void load_tile(const R6Buffers& b, bool has_ep, int64_t e0, int lane, TileIn& in, float* lds = nullptr) {
    const int64_t ld = b.ld;
    const bool live = FULL ? true : (lane < b.n - e0);
    in.mass = in.m0 = 4.0e4f;
    in.v0 = in.a0 = in.a1 = in.ep_ret = 0.0f;
    in.a2 = -1.0f;
    in.ep_step = 0u;
#pragma unroll
    for (int j = 0; j < kDyn; ++j) in.y[j] = (j == 0) ? 1000.0f : ((j == 6) ? 1.0f : 0.0f);  // benign padding lane
    if (live) {
        const float* row = b.state + e0;   // SoA rows: coalesced 256 B per row; the row pointer advances in SALU
#pragma unroll
        for (int j = 0; j < kDyn; ++j) {
            in.y[j] = uniform_ptr(row)[lane];
            row += ld;
        }
        in.mass = uniform_ptr(row)[lane];
        in.m0 = uniform_ptr(b.epconst + e0)[lane];
        if constexpr (SHAPING == R6_SHAPING_VELOCITY) in.v0 = uniform_ptr(b.epconst + (ld + e0))[lane];
        if (has_ep) {
            in.ep_step = uniform_ptr(b.ep_step + e0)[lane];
            in.ep_ret = uniform_ptr(b.ep_return + e0)[lane];
        }
        const gptr<const float> act = uniform_ptr(b.action + e0 * R6_ACTION_DIM);
#if R6_ACT_LDS
        // A/B variant: the (64,3) action block staged through the wave's LDS tile -- three coalesced 256 B loads,
        // row-major image in LDS, read back at a 3-word lane stride (conflict-free: 3 is odd).
        if constexpr (FULL) {
            const float g0 = act[lane], g1 = act[lane + 64], g2 = act[lane + 128];
            lds[lane] = g0;
            lds[lane + 64] = g1;
            lds[lane + 128] = g2;
            wave_lds_sync();
            in.a0 = lds[3 * lane + 0];
            in.a1 = lds[3 * lane + 1];
            in.a2 = lds[3 * lane + 2];
            wave_lds_sync();
        } else
#endif
        {
            // 12 contiguous bytes per lane: the compiler issues ONE global_load_dwordx3 per lane, i.e. the wave reads its
            // 768 B action block with a single instruction over 6 whole cache lines -- already the access an LDS-staged
            // edge would make, without the LDS round trip (A/B with R6_ACT_LDS=1: DESIGN.md section 4).
            in.a0 = act[3 * lane + 0];
            in.a1 = act[3 * lane + 1];
            in.a2 = act[3 * lane + 2];
        }
    }
}

#ifndef R6_PREFETCH_ALL
#define R6_PREFETCH_ALL 0   // A/B: 1 (with -DR6_EARLY_LOAD=0) makes the RK4 instances prefetch the next tile into registers like the Euler ones
#endif
#ifndef R6_RK4_WAVES
#define R6_RK4_WAVES 5
#endif
#ifndef R6_RK4_SUBSTEP_WAVES
#define R6_RK4_SUBSTEP_WAVES 4   // RK4 with several sub-steps per env step: the loop-carried state does not fit the 96 registers of five waves
                                 // (round 3: two registers spilled to scratch); these instances are compiled for four
#endif
#ifndef R6_RK45_WAVES
#define R6_RK45_WAVES 4    // waves per SIMD the adaptive instances are compiled for (128 registers; round 3: 1 -> 254 registers, two waves)
#endif
#ifndef R6_RK45_STASH
#define R6_RK45_STASH 12   // guest list of the adaptive instances (entries per WAVE; see step_kernel)
#endif
#ifndef R6_RK45_DEFER_DEFAULT
#define R6_RK45_DEFER_DEFAULT 6   // a wave hands its last <= 6 unfinished lanes to the workgroup's straggler pass (0 = never: every wave follows its slowest lane)
#endif
#ifndef R6_RK45_WG_DYNAMIC
#define R6_RK45_WG_DYNAMIC 0   // A/B: the adaptive instances' waves pull their workgroup's tiles from an LDS counter (tiles take 2-4 attempts: equal lists are unequal work)
#endif
#ifndef R6_EULER_WAVES
#define R6_EULER_WAVES 5   // 90 VGPRs with the register prefetch of the next tile; a budget of 4 waves lets the allocator
                           // drift past 96 and silently lose the fifth wave (34.2 -> 34.9 us when that happened)
#endif
// __launch_bounds__(256, 5): 5 waves per SIMD = a 96-VGPR budget (the RK4 instance would take 101 and drop
// to 4 waves; the cap costs no scratch).  Measured A/B at N = 2^20: 46.7 -> 44.0 us.
template <int INTEG, int SHAPING, int OD, bool NT = false, bool ONE = false>
__global__ __launch_bounds__(kBlock, (INTEG == R6_RK45 ? R6_RK45_WAVES : (INTEG == R6_EULER ? R6_EULER_WAVES : (ONE ? R6_RK4_WAVES : R6_RK4_SUBSTEP_WAVES)))) void step_kernel(const R6Params p, const R6Buffers b, const uint64_t seed,
                                                      const uint64_t step_index, const uint64_t env_offset,
                                                      const uint32_t even_share_q16, const uint32_t sweep) {
    // Hot scalars are read from `p` / `b` directly (SGPRs, hoisted out of the tile loop by the
    // compiler).  Everything that is needed once per tile as a per-lane value (the 14 normaliser
    // reciprocals) sits in LDS, and everything that only rare paths need (terminal outputs, the
    // reset distribution, the RNG keys) is re-read from the kernarg segment inside those paths
    // (load_params / load_buffers), so that neither occupies SGPRs across the loop.
    __shared__ __attribute__((aligned(16))) float tiles[kWaves][64 * OD];
    __shared__ float lds_inv_norm[R6_STATE_DIM];
    __shared__ int rq[kResetQueueCap];
    __shared__ int rq_count;
    __shared__ int wg_next;
    __shared__ stat_t blk_stats[R6_NUM_STATS];
    // Adaptive instances: guests.  scipy's RK45 needs one step for 32 % of the env steps of the bench's steady state, two for 66.5 %,
    // three or four for 1.4 % (gpurun_out/r5_rk45_attempts.json) -- and a wave follows its slowest lane: 2.74 attempts per 64-env tile
    // against 1.70 per lane, 0.9 lanes per wave alive in the attempts beyond the second.  A wave therefore stops after its second
    // attempt when at most `defer_max` lanes are still integrating: their state goes to the wave's own guest list in LDS
    // (structure of arrays, one column per env) and the wave finishes its tile without them.  In the attempts of its LATER tiles the
    // lanes that have nothing to integrate any more (a third of them in every second attempt) lend themselves to the guests: such a
    // lane parks its own finished state in LDS, runs the attempt on a guest's, puts the guest back and takes its own state
    // again.  A guest thus costs no wave-level attempt of its own.  After its last tile (where nothing is handed over) the wave
    // finishes whatever guest still integrates and does for all of them what process_tile does for a tile's lanes, with
    // per-lane addresses.  Nothing is shared between waves; same arithmetic per env, whoever's lane runs it: bit for bit the
    // results without the hand-over (tests/test_gpu_parity.py::test_rk45_straggler_pass_is_bit_identical).
    // (Round 5 first built the per-WORKGROUP form the round-4 review asked for -- one stash, drained densely by the last wave
    // to leave the tile loop: 16 % fewer vector instructions and 114.3 -> 109.6 us only: the pass is two attempts and a reward
    // phase of ONE wave alone on its SIMD, latency-bound at ~3 ns per instruction, ~12 us at the end of every workgroup while
    // its other waves wait at the barrier (SQ_WAIT_ANY +29 %; gpurun_out/pmc_rk45_r5/summary.json; code at commit 8b63f76).)
    constexpr bool kAdaptive = (INTEG == R6_RK45);
    constexpr int kG = kAdaptive ? R6_RK45_STASH : 1;
    static_assert(kG <= 32, "the pending mask is one 32-bit word");
    constexpr int kGuestWords = 34;   // env | y[13] | mass(t=0) | t, h_abs | K0[9] | ctl | thrust vector, thrust, mdot | 1/Jt, Jx-Jt
    constexpr int kSaveWords = 23;    // a hosting lane's own (finished) state: y[13] | mass(t=0) | t | ctl | control[5] | inertia[2]
    __shared__ float gst[kAdaptive ? kWaves : 1][kAdaptive ? kGuestWords : 1][kG];
    __shared__ float sv[kAdaptive ? kWaves : 1][kAdaptive ? kSaveWords : 1][kG];
    // ... and a parking lot: what a lane needs again only AFTER the integration (v0, the episode counters, the thrust magnitude)
    // waits in LDS meanwhile -- the attempt loop runs at the 128-register limit of four waves per SIMD, and every register it
    // does not have is a scratch access per attempt
    __shared__ float park[kAdaptive ? kWaves : 1][kAdaptive ? 4 : 1][64];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform -> scalar tile arithmetic
    float* tile = tiles[wave];
    const int defer_max = kAdaptive ? (int)((sweep >> 8) & 0xffu) : 0;
    int g_n = 0;             // adaptive instances: entries of this wave's guest list (wave-uniform)
    uint32_t g_pend = 0u;    // ... and which of them still integrate
    const int64_t n = b.n;
    const int64_t n_tiles = (n + 63) >> 6;
    // XCD-aware tile split.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8 = XCC_ID, checked with
    // s_getreg in the trace build), and the odd XCDs of an MI355X see a ~1 us longer load latency than the even
    // ones (per-wave stamps, scripts/trace_timeline.py: load wait 3.1 vs 2.0 us per tile, same compute time): with
    // equal shares the even XCDs are done at 0.8 of the launch.  even_share_q16 > 0 (persistent grids only): the tile
    // range is cut into an even-XCD pool of n_tiles * share / 65536 tiles and an odd-XCD pool with the rest, each
    // walked round-robin by the waves of its own workgroups.  0: one pool for everybody.
    // (Round 3 also measured a pool of late-dispatched tail workgroups, a fork/join split of the shard over two streams and
    // staggered first loads: +-0, +17 us eager and +-0 -- profiles/r3_experiments.json; removed in round 4.  Round 4 measured,
    // on one set of buffers (scripts/ab_swap.py; code at commit c9001c0, numbers in profiles/r4_experiments.json): finished envs
    // re-sampled by their own wave inside the tile loop instead of the dense drain below, +3.4 ... +4.1 us; the draining wave
    // rotated over the SIMDs, +-0.5; the hot kernel arguments first and preloaded into SGPRs by the command processor
    // (-amdgpu-kernarg-preload-count), +0.6; a 1024-workgroup grid with exactly four tiles per wave, +2.3; the register
    // prefetch at four and five waves per SIMD, +3.2 / +5.2; the last 3-20 % of the tiles handed out in the final round from 32
    // self-resetting counters, claimed a tile ahead, -0.1 ... +0.6 (code at commit d51f2ad).  None moved the launch by the
    // 1 us the review's stop rule asks for: DESIGN.md section 4.  Also measured: the next tile requested in the MIDDLE of a tile, between
    // the dynamics and the reward phase (the stages of the dynamics decide the register budget): the reward phase does not have the 18
    // registers free either -- 25-48 spills to scratch, +9.6 ... +19 us.)
    int64_t nw = (int64_t)gridDim.x * kWaves;
    int64_t t_first = (int64_t)blockIdx.x * kWaves + wave;
    int64_t pool_base = 0, pool_tiles = n_tiles;
    if (even_share_q16 != 0u) {
        const int64_t n_even = (n_tiles * (int64_t)even_share_q16) >> 16;
        const bool odd = (blockIdx.x & 1u) != 0u;
        pool_base = odd ? n_even : 0;
        pool_tiles = odd ? n_tiles - n_even : n_even;
        nw = (int64_t)(gridDim.x >> 1) * kWaves;
        t_first = (int64_t)(blockIdx.x >> 1) * kWaves + wave;
    }
    const bool has_ep = b.ep_step != nullptr && b.ep_return != nullptr;
    const bool opt_flags_terms = b.flags != nullptr || b.terms != nullptr || b.nfev != nullptr;

#if defined(R6_PHASE_TIMING) || defined(R6_TRACE)   // dev builds: 100 MHz wall-clock stamps instead of the episode statistics
    const uint64_t pt0 = wall_clock64();
#endif
#ifdef R6_CLOCK_PROBE   // dev build (scripts/clock_probe.py): shader cycles against the 100 MHz wall clock over a workgroup's life
    const uint64_t cp_r0 = wall_clock64();
    const uint64_t cp_s0 = __builtin_amdgcn_s_memtime();
#endif
    if (tid == 0) {
        rq_count = 0;
        wg_next = kWaves;   // list entries 0..kWaves-1 are the waves' fixed first tiles
    }
    if (tid < R6_NUM_STATS) blk_stats[tid] = 0ull;
    if (tid < R6_STATE_DIM) {  // per-lane read of the kernarg segment (R6Params is argument 0)
        const kernarg_ptr_t k = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        lds_inv_norm[tid] = ((const __attribute__((address_space(4))) float*)(k + offsetof(R6Params, inv_norm)))[tid];
    }
    // kEarly (RK4 / RK45 instances): the barrier that publishes the LDS set-up above comes AFTER the wave has requested its
    // first tile (below) -- nothing of the set-up is needed before a tile's reward phase, and the per-lane kernarg read is a
    // memory round trip that the first loads of the launch need not queue behind.
    constexpr bool kEarly = (R6_EARLY_LOAD != 0) && (INTEG != R6_EULER);
    if constexpr (!kEarly) __syncthreads();

    // One 64-env tile.  FULL = all 64 lanes are live (every tile but possibly the last one).
    auto process_tile = [&](const int64_t e0, const TileIn& in, auto full_c, const bool tile_may_defer) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;
        bool overflow = false;       // this lane's finished episode did not fit into the reset queue
        const int64_t rem = n - e0;  // envs from the tile's start to the end of the shard (> 0)
        const bool lane_live = FULL ? true : (lane < rem);
        bool deferred = false;       // adaptive instances: this lane's env went to the straggler stash, the pass after the loop finishes it
        float y[kDyn];
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = in.y[j];
        float mass = in.mass, ep_ret = in.ep_ret;
        const float m0 = in.m0, a0 = in.a0, a1 = in.a1, a2 = in.a2;
        float v0 = in.v0;
        uint32_t ep_step = in.ep_step;

        // ---- dynamics (sim:69-143) -----------------------------------------------------------------
        Control c = make_control_normalised(p, a0, a1, a2);
        const Inertia J = make_inertia(m0);
#ifdef R6_ABLATE_INTEG   // timing-only build: dynamics skipped, inputs kept live
        int status = 0, nfev = 0;
        asm volatile("" :: "v"(c.tbx), "v"(c.tby), "v"(c.tbz), "v"(J.inv_jt), "v"(J.jdiff));
        y[3] += p.dt * c.mdot;
#else
        int nfev = 0;
        int status;
        if constexpr (INTEG == R6_RK45) {
            // integrate_rk45() with the wave-level loop written out: attempt after attempt while any lane of the tile integrates; from the
            // third attempt on, a remainder of at most defer_max lanes becomes guests (not in a wave's last tile: nobody would host them)
            park[wave][0][lane] = v0;
            park[wave][1][lane] = ep_ret;
            park[wave][2][lane] = __uint_as_float(ep_step);
            park[wave][3][lane] = c.thrust;
            asm volatile("" ::: "memory");
            Rk45Lane L;
            Inertia Jl = J;          // (a hosting lane runs a guest's inertia for one attempt)
            rk45_begin(y, mass, c, Jl, p.dt, L);
            int attempts = 0;
            float(*const gl)[kG] = gst[wave];
            float(*const sl)[kG] = sv[wave];
#pragma unroll 1
            for (;;) {
                const bool active = rk45_active(L, p.dt);
                const uint64_t m = __ballot(active);
                if (m == 0ull) break;
                // (padding lanes of the shard's last tile integrate a benign state like everybody else, but nobody needs their
                //  result: they are dropped at the hand-over and never take a slot)
                const uint64_t ml = FULL ? m : __ballot(active && lane_live);
                const int cnt = __popcll(ml);
                if (tile_may_defer && attempts >= 2 && cnt <= defer_max && g_n + cnt <= kG) {
                    if (active && lane_live) {
                        const int slot = g_n + __popcll(ml & ((1ull << lane) - 1ull));
                        gl[0][slot] = __int_as_float((int)(e0 + lane));
#pragma unroll
                        for (int j = 0; j < kDyn; ++j) gl[1 + j][slot] = y[j];
                        gl[14][slot] = mass;
                        gl[15][slot] = L.t;
                        gl[16][slot] = L.h_abs;
#pragma unroll
                        for (int d = 0; d < dp45::D; ++d) gl[17 + d][slot] = L.K0[d];
                        gl[26][slot] = __uint_as_float(L.ctl);
                        gl[27][slot] = c.tbx;
                        gl[28][slot] = c.tby;
                        gl[29][slot] = c.tbz;
                        gl[30][slot] = park[wave][3][lane];
                        gl[31][slot] = c.mdot;
                        gl[32][slot] = Jl.inv_jt;
                        gl[33][slot] = Jl.jdiff;
                    }
                    if (active) deferred = true;
                    g_pend |= (cnt > 0 ? ((1u << cnt) - 1u) : 0u) << g_n;
                    g_n += cnt;
                    break;
                }
                // ---- guests: pending entry k of the list rides in the k-th lane of this attempt that has nothing of its own to do ----
                int my_entry = -1;
                const uint32_t pend0 = g_pend;
                if (pend0 != 0u) {
                    uint32_t pe = pend0;
                    uint64_t im = ~m;
                    while (pe != 0u && im != 0ull) {
                        const int e = __builtin_ctz(pe), l = __builtin_ctzll(im);
                        pe &= pe - 1u;
                        im &= im - 1ull;
                        if (lane == l) my_entry = e;
                    }
                    if (my_entry >= 0) {
#pragma unroll
                        for (int j = 0; j < kDyn; ++j) sl[j][my_entry] = y[j];
                        sl[13][my_entry] = mass;
                        sl[14][my_entry] = L.t;
                        sl[15][my_entry] = __uint_as_float(L.ctl);
                        sl[16][my_entry] = c.tbx;
                        sl[17][my_entry] = c.tby;
                        sl[18][my_entry] = c.tbz;
                        sl[19][my_entry] = c.thrust;
                        sl[20][my_entry] = c.mdot;
                        sl[21][my_entry] = Jl.inv_jt;
                        sl[22][my_entry] = Jl.jdiff;
#pragma unroll
                        for (int j = 0; j < kDyn; ++j) y[j] = gl[1 + j][my_entry];
                        mass = gl[14][my_entry];
                        L.t = gl[15][my_entry];
                        L.h_abs = gl[16][my_entry];
#pragma unroll
                        for (int d = 0; d < dp45::D; ++d) L.K0[d] = gl[17 + d][my_entry];
                        L.ctl = __float_as_uint(gl[26][my_entry]);
                        c.tbx = gl[27][my_entry];
                        c.tby = gl[28][my_entry];
                        c.tbz = gl[29][my_entry];
                        c.thrust = gl[30][my_entry];
                        c.mdot = gl[31][my_entry];
                        Jl.inv_jt = gl[32][my_entry];
                        Jl.jdiff = gl[33][my_entry];
                    }
                }
                if (active || my_entry >= 0) rk45_attempt(y, mass, c, Jl, p.dt, L);
                if (pend0 != 0u) {
                    const uint64_t fin = __ballot(my_entry >= 0 && !rk45_active(L, p.dt));
                    if (my_entry >= 0) {
#pragma unroll
                        for (int j = 0; j < kDyn; ++j) gl[1 + j][my_entry] = y[j];
                        gl[15][my_entry] = L.t;
                        gl[16][my_entry] = L.h_abs;
#pragma unroll
                        for (int d = 0; d < dp45::D; ++d) gl[17 + d][my_entry] = L.K0[d];
                        gl[26][my_entry] = __uint_as_float(L.ctl);
#pragma unroll
                        for (int j = 0; j < kDyn; ++j) y[j] = sl[j][my_entry];
                        mass = sl[13][my_entry];
                        L.t = sl[14][my_entry];
                        L.ctl = __float_as_uint(sl[15][my_entry]);
                        c.tbx = sl[16][my_entry];
                        c.tby = sl[17][my_entry];
                        c.tbz = sl[18][my_entry];
                        c.thrust = sl[19][my_entry];
                        c.mdot = sl[20][my_entry];
                        Jl.inv_jt = sl[21][my_entry];
                        Jl.jdiff = sl[22][my_entry];
                    }
                    // the guests that arrived leave the pending set: the pairing above again, on scalars
                    uint32_t pe = pend0;
                    uint64_t im = ~m;
                    while (pe != 0u && im != 0ull) {
                        const int e = __builtin_ctz(pe), l = __builtin_ctzll(im);
                        pe &= pe - 1u;
                        im &= im - 1ull;
                        if ((fin >> l) & 1ull) g_pend &= ~(1u << e);
                    }
                }
                ++attempts;
            }
            rk45_end(y, mass, c, L);
            status = rk45_status(L);
            nfev = rk45_nfev(L);
            asm volatile("" ::: "memory");
            v0 = park[wave][0][lane];
            ep_ret = park[wave][1][lane];
            ep_step = __float_as_uint(park[wave][2][lane]);
            c.thrust = park[wave][3][lane];
        } else {
            status = integrate<INTEG, true, ONE>(y, mass, c, J, p.dt, p.n_substeps);
            nfev = (INTEG == R6_RK4 ? 4 : 1) * (ONE ? 1 : p.n_substeps);
        }
#endif
        const bool live = lane_live && !deferred;   // a deferred lane's rows of this tile are written by the straggler pass

        // `ln` = the lane id again, opaque to the optimiser: everything addressed after the integration is
        // re-derived from it here instead of being hoisted out of the tile loop and kept in VGPRs across
        // the RK4 stages (where the register budget of 5 waves/SIMD is decided).
        // (masked to 6 bits after the laundering: with a known-small unsigned index the stores below take the
        // SGPR-base + 32-bit-VGPR-offset form instead of a 64-bit VALU address add each.)
        unsigned lnu = lane;
        asm volatile("" : "+v"(lnu));
        const int ln = (INTEG == R6_EULER) ? (int)lnu : (int)(lnu & 63u);   // Euler: A/B says the VALU-address form (+0.5 us with the mask)

        // ---- termination (env:213, :591-593), reward (env:215, :228-229, :317-402),
        //      fused ClipReward / TimeLimit / Monitor (main_6DOF.py:33-51).  In the RK4 instances the
        //      ~40 parameters of this phase are re-read from the kernarg segment here (scalar loads, cache
        //      hits) instead of living in SGPRs across the whole loop, where they spill (A/B: 43.8 -> 42.3
        //      us); the short Euler loop has too little work to hide that latency and keeps them resident. --
        bool bnd, done, truncated = false, landed;
        float rew;
        {
            const R6Params pr = [&]() __attribute__((always_inline)) {
                if constexpr (INTEG == R6_RK4) return load_params();
                else return p;
            }();
            bnd = bounds_violation(pr, y[0], y[1], y[2]);
            done = (status != 0) || bnd;
#ifdef R6_ABLATE_REWARD  // timing-only build: reward skipped
            RewardOut rw;
            for (int q = 0; q < R6_NUM_TERMS; ++q) rw.terms[q] = 0.0f;
            rw.sum = y[0] * pr.alfa + c.thrust * pr.beta + mass + v0;
            rw.landed = false;
#else
            RewardOut rw = reward<SHAPING>(pr, y, mass, c, v0);
#endif
            landed = rw.landed;
            rew = finalize_reward(pr, rw, bnd, fmaf(2.0f * c.thrust, frcp(pr.max_thrust), -1.0f), y);   // a2 back from the thrust
            if (live && opt_flags_terms) {   // per-term rewards (info["rewards_dict"]) leave here, so they die here
                const R6Buffers bo = load_buffers();
                if (bo.terms) {
                    float* trow = bo.terms + e0;
#pragma unroll
                    for (int q = 0; q < R6_NUM_TERMS; ++q) {
                        uniform_ptr(trow)[ln] = rw.terms[q];
                        trow += bo.ld;
                    }
                }
            }
            if (pr.clip_reward) rew = fminf(fmaxf(rew, pr.clip_lo), pr.clip_hi);
            ep_step += 1u;
            ep_ret += rew;
            if (pr.max_episode_steps > 0 && ep_step >= (uint32_t)pr.max_episode_steps) {
                truncated = !done;
                done = true;
            }
        }

        // ---- outputs: reward, done, state rows, counters (RK4: pointers re-read from kernarg, see above) --
        if (live) {
            const R6Buffers bs = [&]() __attribute__((always_inline)) {
                if constexpr (INTEG == R6_RK4) return load_buffers();
                else return b;
            }();
            stream_store<NT>(uniform_ptr(bs.reward + e0) + ln, rew);
            stream_store<NT>(uniform_ptr(bs.done + e0) + ln, done ? 1 : 0);
            float* row = bs.state + e0;
#pragma unroll
            for (int j = 0; j < kDyn; ++j) {
                uniform_ptr(row)[ln] = y[j];
                row += bs.ld;
            }
            uniform_ptr(row)[ln] = mass;
            if (has_ep) {
                uniform_ptr(bs.ep_step + e0)[ln] = ep_step;
                uniform_ptr(bs.ep_return + e0)[ln] = ep_ret;
            }
            if (opt_flags_terms) {   // optional outputs: their pointers are fetched only when present
                const R6Buffers bo = load_buffers();
                if (bo.flags)
                    uniform_ptr(bo.flags + e0)[ln] = (uint8_t)((status == 1 ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) |
                                                                 (truncated ? R6_FLAG_TRUNCATED : 0));
                if (bo.nfev) uniform_ptr(bo.nfev + e0)[ln] = (uint8_t)(nfev > 255 ? 255 : nfev);
            }
        }

        // ---- observation (env:503-504): registers -> LDS row-major tile ---------------------------------
#pragma unroll
        for (int j = 0; j < kDyn; ++j) {
            if (j < OD) tile[ln * OD + j] = y[j] * lds_inv_norm[j];
        }
        if (OD == 14) tile[ln * OD + (OD - 1)] = mass * lds_inv_norm[13];

        // ---- finished episodes: terminal outputs now; the re-sampling (env:180-199) is deferred to
        //      the dense drain after the loop, which overwrites state / obs / counters of these envs ---
        if (live && done) {
            const R6Buffers bb = load_buffers();
            if (bb.stats) {
                stats_add_episode(blk_stats, ep_ret, ep_step, status == 1, bnd, truncated, landed);
            }
            if (p.auto_reset) {
                const int64_t i = e0 + ln;
                if (bb.final_obs) {
#pragma unroll
                    for (int j = 0; j < OD; ++j) bb.final_obs[i * OD + j] = tile[ln * OD + j];
                }
                if (bb.fin_return) bb.fin_return[i] = ep_ret;
                if (bb.fin_length) bb.fin_length[i] = ep_step;
                const int slot = atomicAdd(&rq_count, 1);
                if (slot < kResetQueueCap) rq[slot] = (int)i;
                else overflow = true;   // queue full (mass termination): re-sampled by the tile loop itself
            }
        }

        // ---- LDS tile -> contiguous 16 B-per-ln stores of the (64, OD) observation block ---------------
        wave_lds_sync();
        {
            const gptr<float> dst = uniform_ptr(((INTEG == R6_RK4) ? load_buffers().obs : b.obs) + e0 * OD);
            if constexpr (FULL) {
                // 64 * OD / 4 float4 chunks: three full rounds of 64 lanes (all LDS reads first, then the stores) and
                // a tail of 32 (OD 14) or 16 (OD 13) lanes
                constexpr int n4 = 64 * OD / 4, kRounds = n4 / 64, kTail = n4 - 64 * kRounds;
#if R6_OBS_BATCH
                f4 v[kRounds];
#pragma unroll
                for (int k = 0; k < kRounds; ++k) v[k] = ld4(&tile[4 * (ln + 64 * k)]);
#pragma unroll
                for (int k = 0; k < kRounds; ++k) stream_store<NT>((gptr<f4>)(dst + 4 * (ln + 64 * k)), v[k]);
#else
#pragma unroll
                for (int k = 0; k < kRounds; ++k) stream_store<NT>((gptr<f4>)(dst + 4 * (ln + 64 * k)), ld4(&tile[4 * (ln + 64 * k)]));
#endif
                if (kTail > 0 && ln < kTail) {
                    const int q = ln + 64 * kRounds;
                    stream_store<NT>((gptr<f4>)(dst + 4 * q), ld4(&tile[4 * q]));
                }
            } else {
                for (int k = ln; k < (int)rem * OD; k += 64) dst[k] = tile[k];
            }
        }
        wave_lds_sync();  // before the next tile overwrites the LDS tile
        return overflow;
    };

    // ---- adaptive instances: what becomes of a wave's guests (see `gst` above).  Every guest that has arrived gets what process_tile
    //      gives a tile's lanes -- termination, reward, outputs, episode end -- with per-lane addresses, and leaves the list.
    //      final: whoever still integrates is finished first (rare: a guest rides in every attempt after its hand-over that has
    //      a free lane); otherwise (the list is nearly full in the middle of the tile loop) the guests still under way move to the
    //      front of the list and stay. ----------------------------------------------------------------------------------------
    auto flush_guests = [&](const bool final) __attribute__((always_inline)) {
        if constexpr (kAdaptive) {
            if (g_n > 0) {
                // the observation block of a tile is stored whole, a guest's row included: those stores come first
                __builtin_amdgcn_s_waitcnt(0);
                const bool mine = lane < g_n;
                const int e = mine ? lane : 0;
                float(*const gl)[kG] = gst[wave];
                const R6Buffers bb = load_buffers();
                const int64_t ld = bb.ld;
                const int64_t i = (int64_t)__float_as_int(gl[0][e]);
                float y[kDyn];
#pragma unroll
                for (int j = 0; j < kDyn; ++j) y[j] = gl[1 + j][e];
                float mass = gl[14][e];
                Rk45Lane L;
                L.t = mine ? gl[15][e] : p.dt;    // an idle lane: nothing to integrate
                L.h_abs = gl[16][e];
#pragma unroll
                for (int d = 0; d < dp45::D; ++d) L.K0[d] = gl[17 + d][e];
                L.ctl = __float_as_uint(gl[26][e]);
                Control c;
                c.tbx = gl[27][e];
                c.tby = gl[28][e];
                c.tbz = gl[29][e];
                c.thrust = gl[30][e];
                c.mdot = gl[31][e];
                Inertia J;
                J.inv_jt = gl[32][e];
                J.jdiff = gl[33][e];
                if (final) {
#pragma unroll 1
                    while (__any(rk45_active(L, p.dt))) {
                        if (rk45_active(L, p.dt)) rk45_attempt(y, mass, c, J, p.dt, L);
                    }
                }
                const bool under_way = mine && rk45_active(L, p.dt);       // (never with `final`)
                const bool valid = mine && !under_way;
                const uint64_t keep = __ballot(under_way);
                if (under_way) {   // to the front of the list (every lane has read its entry before anybody writes)
                    const int slot = __popcll(keep & ((1ull << lane) - 1ull));
                    gl[0][slot] = __int_as_float((int)i);
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) gl[1 + j][slot] = y[j];
                    gl[14][slot] = mass;
                    gl[15][slot] = L.t;
                    gl[16][slot] = L.h_abs;
#pragma unroll
                    for (int d = 0; d < dp45::D; ++d) gl[17 + d][slot] = L.K0[d];
                    gl[26][slot] = __uint_as_float(L.ctl);
                    gl[27][slot] = c.tbx;
                    gl[28][slot] = c.tby;
                    gl[29][slot] = c.tbz;
                    gl[30][slot] = c.thrust;
                    gl[31][slot] = c.mdot;
                    gl[32][slot] = J.inv_jt;
                    gl[33][slot] = J.jdiff;
                }
                g_n = __popcll(keep);
                g_pend = g_n > 0 ? ((1u << g_n) - 1u) : 0u;
                float v0 = 0.0f;
                if constexpr (SHAPING == R6_SHAPING_VELOCITY) v0 = bb.epconst[ld + i];
                uint32_t ep_step = 0u;
                float ep_ret = 0.0f;
                if (has_ep) {
                    ep_step = bb.ep_step[i];
                    ep_ret = bb.ep_return[i];
                }
                rk45_end(y, mass, c, L);
                const int status = rk45_status(L), nfev = rk45_nfev(L);
                // termination, reward, fused wrappers, outputs: process_tile's sequence, operation for operation
                const bool bnd = bounds_violation(p, y[0], y[1], y[2]);
                bool done = (status != 0) || bnd, truncated = false;
                RewardOut rw = reward<SHAPING>(p, y, mass, c, v0);
                const bool landed = rw.landed;
                float rew = finalize_reward(p, rw, bnd, fmaf(2.0f * c.thrust, frcp(p.max_thrust), -1.0f), y);
                if (valid && bb.terms) {
#pragma unroll
                    for (int q = 0; q < R6_NUM_TERMS; ++q) bb.terms[q * ld + i] = rw.terms[q];
                }
                if (p.clip_reward) rew = fminf(fmaxf(rew, p.clip_lo), p.clip_hi);
                ep_step += 1u;
                ep_ret += rew;
                if (p.max_episode_steps > 0 && ep_step >= (uint32_t)p.max_episode_steps) {
                    truncated = !done;
                    done = true;
                }
                bool overflow = false;
                if (valid) {
                    bb.reward[i] = rew;
                    bb.done[i] = done ? 1 : 0;
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) bb.state[j * ld + i] = y[j];
                    bb.state[kDyn * ld + i] = mass;
                    if (has_ep) {
                        bb.ep_step[i] = ep_step;
                        bb.ep_return[i] = ep_ret;
                    }
                    if (bb.flags)
                        bb.flags[i] = (uint8_t)((status == 1 ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) | (truncated ? R6_FLAG_TRUNCATED : 0));
                    if (bb.nfev) bb.nfev[i] = (uint8_t)(nfev > 255 ? 255 : nfev);
                    float ob[OD];
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) {
                        if (j < OD) ob[j] = y[j] * lds_inv_norm[j];
                    }
                    if (OD == 14) ob[OD - 1] = mass * lds_inv_norm[13];
#pragma unroll
                    for (int j = 0; j < OD; ++j) bb.obs[i * OD + j] = ob[j];
                    if (done) {
                        if (bb.stats) stats_add_episode(blk_stats, ep_ret, ep_step, status == 1, bnd, truncated, landed);
                        if (p.auto_reset) {
                            if (bb.final_obs) {
#pragma unroll
                                for (int j = 0; j < OD; ++j) bb.final_obs[i * OD + j] = ob[j];
                            }
                            if (bb.fin_return) bb.fin_return[i] = ep_ret;
                            if (bb.fin_length) bb.fin_length[i] = ep_step;
                            const int slot = atomicAdd(&rq_count, 1);
                            if (slot < kResetQueueCap) rq[slot] = (int)i;
                            else overflow = true;
                        }
                    }
                }
                if (__builtin_expect(__any(overflow), 0)) {
                    __builtin_amdgcn_s_waitcnt(0);
                    if (overflow) reset_env_global<OD>(load_params(), bb, i, seed, step_index, env_offset);
                }
            }
        }
    };

    // Tile loop.  (Measured alternatives that did not pay at N = 2^20, where the loop is VALU-bound:
    // register prefetch of the next tile -- +30 VGPRs, one wave per SIMD less, no gain; deferring
    // ground-event envs to the dense pass -- its serial tail per workgroup costs more than the
    // divergence it removes.)
    //
    // Wave priorities.  The SIMD arbiter serves its oldest wave first, so at equal priority the waves of the
    // workgroups dispatched first run ahead and finish at ~60 % of the launch, and the last quarter of the
    // launch runs on the few waves that were starved until then, with nothing left to hide their latencies
    // behind (wall-clock stamps per wave: scripts/phase_timing.py).  Each wave therefore lowers its own
    // priority as it completes tiles: laggards go first, the waves of a SIMD advance together and the tail
    // shrinks (A/B at N = 2^20: 42.3 -> 40.2 us).  (Priority from the tiles still to do instead of the tiles done, so
    // that a wave with one tile more runs ahead: 0.4-1.6 us slower.)
    //
    // What the per-wave stamps of round 2 show (scripts/trace_timeline.py): first data reaches a
    // SIMD ~4 us after kernel entry (1 us dispatch + kernarg, 3 us for the 24 MB of first-tile loads); from 6 to
    // ~28 us 95-100 % of the SIMDs have a wave in its compute phase; then the SIMDs run dry between 30 and 38 us -- the
    // mean SIMD is done at 0.84 of the launch.  Which SIMD is late changes from launch to launch (no correlation
    // between launches, workgroups land on different CUs each time); the one systematic part is the XCD parity
    // handled by the tile split above.  Pulling tiles from shared global counters balances nothing here: device-scope
    // atomics cross the 8 XCDs' L2s and cost ~140 ns each when 5120 waves use them (840 us per launch, measured);
    // pulling them from an LDS counter inside a workgroup helps the Euler instances only (see below).
    const int64_t n_full = n >> 6;  // tiles [0, n_full) are full; tile n_full (if any) is partial
    int tile_no = 0;
    constexpr bool kPrefetch = (INTEG == R6_EULER) || (R6_PREFETCH_ALL != 0);   // R6_PREFETCH_ALL: A/B (with R6_EARLY_LOAD=0): the RK4 instances prefetch into registers too
    TileIn nxt;
    bool have_nxt = false;
    // Tile hand-out inside a workgroup (Euler instances).  The four waves of a workgroup sit on the four SIMDs of a
    // CU, and per-wave stamps (scripts/trace_timeline.py) show the SIMDs of one CU finishing equal tile lists 3 us
    // apart on average (which one is late changes from launch to launch).  With kWgDyn the workgroup's tiles --
    // groups of kWaves consecutive tiles, one group per round of the grid -- are claimed through an LDS counter: a
    // wave that is ahead takes the next tile of its workgroup (claimed one tile ahead, which is what the register
    // prefetch needs anyway).  The first tile of every wave is fixed.  List index j -> pool tile
    // (j / kWaves) * nw + first tile of the workgroup + (j % kWaves).  Measured at N = 2^20: Euler 35.1 -> 34.0 us;
    // the RK4 instances lose with it (38.1 -> 38.7 claimed ahead, 40.0 claimed late: a wave that takes a tile late
    // in the launch finishes it alone) and keep their fixed per-wave lists.
    constexpr bool kWgDyn = (R6_WG_DYNAMIC != 0) && kPrefetch;   // (the adaptive instances with it AND the register prefetch it implied: 108.8 -> 122.8 us, round 4)
    // kWgList: the workgroup-wide tile list without the register prefetch, claimed at the bottom of the rotated loop (adaptive instances, A/B)
    constexpr bool kWgLate = kAdaptive && (R6_RK45_WG_DYNAMIC != 0) && !kWgDyn;
    constexpr bool kWgList = kWgDyn || kWgLate;
    const int64_t wg_first = t_first - wave;
    // Order of a wave's (workgroup's) tile list, chosen per launch by the host (launch_step_inst has the numbers).
    // sweep bit 0: the list is walked backwards -- the tile a wave wrote last in the previous launch is the one it reads
    // first in this one (workgroup i always runs on XCD i % 8, so those rows were last touched through the same L2).
    // It pays for the memory-bound Euler instances only; the RK4 instances measure 0.8-1.5 us FASTER with one fixed
    // order at N = 2^20 (their first loads wait 2.5 us after a forward launch and 3.3 us after a reversed one).
    // sweep bit 1 (dispatcher-balanced grid, N >= 3 M envs): the whole tile range is mirrored instead, what the previous
    // step wrote last is read first while it is still in the 256 MB Infinity Cache.
    const bool rev = (sweep & 1u) != 0u;
    const int n_my = t_first < pool_tiles ? (int)((pool_tiles - 1 - t_first) / nw) + 1 : 0;   // fixed list: tiles of this wave
    int n_wg = 0;                                                                               // kWgList: tiles of this workgroup
    if constexpr (kWgList) {
#pragma unroll 1
        for (int64_t g0 = wg_first; g0 < pool_tiles; g0 += nw) n_wg += (pool_tiles - g0 < kWaves) ? (int)(pool_tiles - g0) : kWaves;
    }
    auto list_tile = [&](int pos) __attribute__((always_inline)) -> int64_t {
        if constexpr (kWgList) {
            const int jj = rev ? n_wg - 1 - pos : pos;
            return wg_first + (int64_t)(jj / kWaves) * nw + (jj % kWaves);
        } else {
            return t_first + (int64_t)(rev ? n_my - 1 - pos : pos) * nw;
        }
    };
    const int n_pos = kWgList ? n_wg : n_my;
    int wg_jn = -1;
    if constexpr (kEarly) {
        // Rotated tile loop: a tile's inputs are requested at the bottom of the previous iteration, the first tile's
        // before the set-up barrier.
        auto tile_at = [&](int pos) __attribute__((always_inline)) -> int64_t {
            const int64_t ts = pool_base + list_tile(pos);
            return (sweep & 2u) ? (n_tiles - 1 - ts) : ts;
        };
        TileIn cur;
        int64_t t = 0;
        int pos = kWgLate ? wave : 0;     // kWgLate: list entries 0..kWaves-1 are the waves' fixed first tiles (wg_next starts behind them)
        if (pos < n_pos) {
            t = tile_at(pos);
            if (t < n_full) load_tile<SHAPING, true>(b, has_ep, t * 64, lane, cur, tile);
            else load_tile<SHAPING, false>(b, has_ep, t * 64, lane, cur);
        }
        __syncthreads();
#pragma unroll 1
        while (pos < n_pos) {
            if (sweep & 4u) {   // A/B: no priority ladder (oldest wave first)
            } else if (tile_no == 0) __builtin_amdgcn_s_setprio(3);
            else if (tile_no == 1) __builtin_amdgcn_s_setprio(2);
            else if (tile_no == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
            ++tile_no;
            bool overflow;
            // a wave's stragglers ride along in its LATER tiles: none are handed over in its last one (with the workgroup-wide list
            // a wave does not know its last tile: the last kWaves entries count as such)
            const bool may_defer = defer_max > 0 && (kWgLate ? (pos + kWaves < n_pos) : (pos + 1 < n_pos));
            if (t < n_full) overflow = process_tile(t * 64, cur, std::true_type{}, may_defer);
            else overflow = process_tile(t * 64, cur, std::false_type{}, may_defer);
            if (__builtin_expect(__any(overflow), 0)) {
                __builtin_amdgcn_s_waitcnt(0);
                if (overflow) reset_env_global<OD>(load_params(), load_buffers(), t * 64 + lane, seed, step_index, env_offset);
            }
            if constexpr (kWgLate) {   // the wave that is done first takes the workgroup's next tile
                int nj = 0;
                if (lane == 0) nj = atomicAdd(&wg_next, 1);
                pos = __builtin_amdgcn_readfirstlane(nj);
            } else {
                ++pos;
            }
            if constexpr (kAdaptive) {   // the guests: all of them after the wave's last tile, the arrived ones when the list runs out of room
                const bool last = !(pos < n_pos);
                if (last || g_n + defer_max > kG) flush_guests(last);
            }
            if (pos < n_pos) {
                t = tile_at(pos);
                if (t < n_full) load_tile<SHAPING, true>(b, has_ep, t * 64, lane, cur, tile);
                else load_tile<SHAPING, false>(b, has_ep, t * 64, lane, cur);
            }
        }
    } else {
#pragma unroll 1
    for (int pos = kWgDyn ? wave : 0; pos < n_pos;) {
        const int64_t tp = list_tile(pos);
        const int64_t ts = pool_base + tp;
        const int64_t t = (sweep & 2u) ? (n_tiles - 1 - ts) : ts;
        if (sweep & 4u) {   // A/B: no priority ladder (oldest wave first)
        } else if (tile_no == 0) __builtin_amdgcn_s_setprio(3);
        else if (tile_no == 1) __builtin_amdgcn_s_setprio(2);
        else if (tile_no == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        ++tile_no;
        TileIn cur;
        bool overflow;
#ifdef R6_TRACE   // dev build (scripts/trace_timeline.py): per wave and tile, 100 MHz stamps around the load / compute / store phases
        uint32_t* trc = reinterpret_cast<uint32_t*>(b.stats) + ((int64_t)blockIdx.x * kWaves + wave) * 24;
        const uint32_t tr_a = (uint32_t)wall_clock64();
#endif
        if (t < n_full) {
            if constexpr (kPrefetch) {
                // The Euler step has too little arithmetic to cover its own loads: the next tile's inputs are
                // requested into registers (18 more, which the 4-waves-per-SIMD budget of this instance has)
                // before this tile is computed: 37.0 -> 34.7 us.  The same costs the RK4 instance 1.2 us.
                if (have_nxt) cur = nxt;
                else load_tile<SHAPING, true>(b, has_ep, t * 64, lane, cur);
                int pos_n = pos + 1;
                if constexpr (kWgDyn) {   // the tile after this one is claimed now, so that its inputs can be requested
                    int nj = 0;
                    if (lane == 0) nj = atomicAdd(&wg_next, 1);
                    wg_jn = __builtin_amdgcn_readfirstlane(nj);
                    pos_n = wg_jn;
                }
                const int64_t tsn = pool_base + list_tile(pos_n < n_pos ? pos_n : pos);
                const int64_t tn = (sweep & 2u) ? (n_tiles - 1 - tsn) : tsn;
                have_nxt = pos_n < n_pos && tn < n_full;
                if (have_nxt) load_tile<SHAPING, true>(b, has_ep, tn * 64, lane, nxt);
            } else {
                load_tile<SHAPING, true>(b, has_ep, t * 64, lane, cur, tile);
            }
#ifdef R6_TRACE
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t tr_b = (uint32_t)wall_clock64();
#endif
            overflow = process_tile(t * 64, cur, std::true_type{}, false);
#ifdef R6_TRACE
            const uint32_t tr_c = (uint32_t)wall_clock64();
            __builtin_amdgcn_s_waitcnt(0);
            const uint32_t tr_d = (uint32_t)wall_clock64();
            if (b.stats && lane == 0 && tile_no <= 5) {
                if (tile_no == 1) {
                    trc[0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
                    trc[1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
                    trc[2] = (uint32_t)pt0;
                }
                trc[3] = tile_no;
                trc[4 * tile_no + 0] = tr_a;
                trc[4 * tile_no + 1] = tr_b;
                trc[4 * tile_no + 2] = tr_c;
                trc[4 * tile_no + 3] = tr_d;
            }
#endif
        } else {
            have_nxt = false;
            load_tile<SHAPING, false>(b, has_ep, t * 64, lane, cur);
            overflow = process_tile(t * 64, cur, std::false_type{}, false);
        }
        if (__builtin_expect(__any(overflow), 0)) {
            // Reset queue full (mass termination, e.g. the first landing wave of envs started in lockstep):
            // these lanes re-sample their env right here, once the tile's terminal outputs have left the wave.
            __builtin_amdgcn_s_waitcnt(0);
            if (overflow) reset_env_global<OD>(load_params(), load_buffers(), t * 64 + lane, seed, step_index, env_offset);
        }
        if constexpr (kWgDyn) {
            if (wg_jn >= 0) {   // claimed ahead by the prefetch
                pos = wg_jn;
                wg_jn = -1;
            } else {            // partial tile / no prefetch issued: claim now (lane 0 asks, the wave reads the answer as a scalar)
                int nj = 0;
                if (lane == 0) nj = atomicAdd(&wg_next, 1);
                pos = __builtin_amdgcn_readfirstlane(nj);
            }
        } else {
            ++pos;
        }
    }
    }   // !kEarly
    __builtin_amdgcn_s_setprio(0);

#ifdef R6_PHASE_TIMING
    const uint64_t pt1 = wall_clock64();
#endif
    // Every wave waits for its own outstanding stores before the barrier (the workgroup-scope fence of __syncthreads()
    // does not emit a vmcnt wait on gfx9, so the wait is explicit): the drain below overwrites the terminal state /
    // obs of the finished envs, and those stores must have been acknowledged by L2 first.
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
#ifdef R6_PHASE_TIMING
    const uint64_t pt2 = wall_clock64();
#endif

    const int n_queued = rq_count;
    if (n_queued > 0) {
        const R6Buffers bb = load_buffers();
#if !defined(R6_PHASE_TIMING) && !defined(R6_TRACE) && !defined(R6_CLOCK_PROBE)
        if (bb.stats && tid < R6_NUM_STATS) bb.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += stats_value(blk_stats, tid);
#endif

        // ---- dense drain: Rocket6DOF.reset (env:180-199) for every episode this workgroup finished ---
        const int n_rec = n_queued < kResetQueueCap ? n_queued : kResetQueueCap;   // the rest was handled in the loop
        // One env per quad of lanes for the Euler instances (measured on one set of buffers, two sessions: 34.97 -> 34.27 and 34.89 ->
        // 34.16 us); the RK4 instances keep one env per lane: no gain there (37.25 | 37.31 us), and the quad form moved nine more
        // scalar-register reloads into their tile loop (the drain shares the kernel's register allocation).
        if constexpr ((R6_RESET_QUAD != 0) && INTEG == R6_EULER) {
            for (int base = wave * 16; base < n_rec; base += kWaves * 16) {   // 16 envs per wave and pass
                const int e = base + (lane >> 2);
                const bool valid = e < n_rec;
                reset_env_quad<OD>(bb, lds_inv_norm, (int64_t)rq[valid ? e : base], valid, lane, seed, step_index, env_offset);
            }
        } else {
            const R6Params pp = load_params();
            for (int k = tid; k < n_rec; k += kBlock)
                reset_env_global<OD>(pp, bb, (int64_t)rq[k], seed, step_index, env_offset);
        }
    } else if (!p.auto_reset) {
        const R6Buffers bb = load_buffers();
#if !defined(R6_PHASE_TIMING) && !defined(R6_TRACE) && !defined(R6_CLOCK_PROBE)
        if (bb.stats && tid < R6_NUM_STATS && blk_stats[0] != 0ull)
            bb.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += stats_value(blk_stats, tid);
#endif
    }
#ifdef R6_CLOCK_PROBE
    if (b.stats && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t cp_s1 = __builtin_amdgcn_s_memtime();
        const uint64_t cp_r1 = wall_clock64();
#ifdef R6_CLOCK_RING      // a ring of R6_CLOCK_RING launches (keyed by the step index) instead of one: the host reads it every R6_CLOCK_RING launches
        double* o = b.stats + ((int64_t)(step_index % R6_CLOCK_RING) * gridDim.x + blockIdx.x) * R6_NUM_STATS;
#else
        double* o = b.stats + (int64_t)blockIdx.x * R6_NUM_STATS;
#endif
        o[0] = (double)(cp_s1 - cp_s0);
        o[1] = (double)(cp_r1 - cp_r0);
        o[2] = (double)(uint32_t)cp_r0;
        o[3] = (double)(uint32_t)cp_r1;
    }
#endif
#ifdef R6_PHASE_TIMING
    if (b.stats && (tid & 63) == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const uint64_t pt3 = wall_clock64();
        double* o = b.stats + (int64_t)blockIdx.x * R6_NUM_STATS;   // stamps as exact integers in float64
        if (tid == 0) {
            o[0] = (double)(uint32_t)pt0;
            o[2] = (double)(uint32_t)pt2;
            o[3] = (double)(uint32_t)pt3;
            o[7] = (double)n_queued;
        }
        // per-wave end of the tile loop: slots 1 (wave 0), 4, 5, 6 (waves 1..3)
        o[wave == 0 ? 1 : 3 + wave] = (double)(uint32_t)pt1;
    }
#endif
}

// -------------------------------------------------------------------------------------------------
// K consecutive Rocket6DOF.step calls in ONE launch (r6_step_k).
//
// The rollout loop of the reference (main_6DOF.py:136-139: model.learn -> env.step per timestep) pays one kernel
// launch per env step; at N = 65 536 (BASELINE configs[1]) a step is 1.5 us of memory time behind ~10 us of launch
// latency.  Here a wave keeps the state of its 64 envs in registers across K steps: per step it reads the 3-float
// action of step k and writes that step's observation / reward / done (and the optional outputs); state rows and
// episode counters are read once before and written once after the K steps.  Finished episodes are re-sampled in
// the lane itself (the same Philox draw, keyed by (seed, global env, global step), that the single-step kernel's
// dense drain makes), so K fused steps produce bit for bit what K r6_step launches produce.
//
// Per-step arrays carry a leading K dimension with the strides of R6StepK (0 = every step overwrites the same
// buffer, i.e. only the last step's outputs survive -- R6_K_LAST_ONLY skips the earlier stores altogether).
// -------------------------------------------------------------------------------------------------
template <int INTEG, int SHAPING, int OD, bool ONE>
__global__ __launch_bounds__(kBlock, (INTEG == R6_RK45 ? R6_RK45_WAVES : 4)) void step_k_kernel(const R6Params p, const R6Buffers b, const R6StepK sk,
                                                                               const uint64_t seed, const uint64_t step_index,
                                                                               const uint64_t env_offset) {
    __shared__ __attribute__((aligned(16))) float tiles[kWaves][64 * OD];
    __shared__ float lds_inv_norm[R6_STATE_DIM];
    __shared__ stat_t blk_stats[R6_NUM_STATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* tile = tiles[wave];
    const int64_t n = b.n, ld = b.ld;
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t nw = (int64_t)gridDim.x * kWaves;
    const bool has_ep = b.ep_step != nullptr && b.ep_return != nullptr;
    const int K = sk.k;
    const bool last_only = (sk.flags & R6_K_LAST_ONLY) != 0;
    const bool direct_obs = (sk.flags & R6_K_DIRECT_OBS_INTERNAL) != 0;

#ifdef R6_TRACE_K   // dev build (scripts/stepk_trace.py): per-wave 100 MHz stamps, written to the rows of `stats`
    const uint64_t tk0 = wall_clock64();
    uint64_t tk1 = 0, tk2 = 0, tk3 = 0;
    int tk_resets = 0;
    uint64_t tk_reg[4] = {0, 0, 0, 0}, tk_reg_d[4] = {0, 0, 0, 0};   // shader cycles per region: dynamics, reward, outputs + episode end, obs stores; all steps / steps with an episode end
#define TKSTAMP(v) __builtin_amdgcn_s_waitcnt(0xc07f); const uint64_t v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xc07f)
#else
#define TKSTAMP(v)
#endif
    if (tid < R6_NUM_STATS) blk_stats[tid] = 0ull;
    if (tid < R6_STATE_DIM) lds_inv_norm[tid] = p.inv_norm[tid];
    const uint64_t step_base = b.step_base ? *b.step_base : (uint64_t)0;   // constant during the launch: read once (a scalar load)
    __syncthreads();

#pragma unroll 1
    for (int64_t t = (int64_t)blockIdx.x * kWaves + wave; t < n_tiles; t += nw) {
        const int64_t e0 = t * 64;
        const int64_t rem = n - e0;
        const bool live = lane < rem;
        const bool full = rem >= 64;
        const int64_t i = e0 + lane;   // this lane's env (valid when live)
        const int lane_ld = live ? lane : 0;   // lane whose action a padding lane may read

        // ---- state of the tile: read once ------------------------------------------------------------------
        float y[kDyn], mass = 4.0e4f, m0 = 4.0e4f, v0 = 0.0f, ep_ret = 0.0f;
        uint32_t ep_step = 0u;
#pragma unroll
        for (int j = 0; j < kDyn; ++j) y[j] = (j == 0) ? 1000.0f : ((j == 6) ? 1.0f : 0.0f);   // benign padding lane
        if (live) {
            const float* row = b.state + e0;
#pragma unroll
            for (int j = 0; j < kDyn; ++j) {
                y[j] = uniform_ptr(row)[lane];
                row += ld;
            }
            mass = uniform_ptr(row)[lane];
            m0 = uniform_ptr(b.epconst + e0)[lane];
            v0 = uniform_ptr(b.epconst + (ld + e0))[lane];   // read for either shaping: a reset rewrites it, like r6_step's drain
            if (has_ep) {
                ep_step = uniform_ptr(b.ep_step + e0)[lane];
                ep_ret = uniform_ptr(b.ep_return + e0)[lane];
            }
        }
        float a0 = 0.0f, a1 = 0.0f, a2 = -1.0f;
        if (live) {
            const gptr<const float> act = uniform_ptr(b.action + e0 * R6_ACTION_DIM);
            a0 = act[3 * lane + 0];
            a1 = act[3 * lane + 1];
            a2 = act[3 * lane + 2];
        }
        // Everything the K loop carries in registers is made "arrived" HERE (the empty asm statements read the registers, so
        // the compiler's s_waitcnt for the loads above lands in front of the loop).  gfx9 counts loads and stores in ONE
        // in-order counter (vmcnt): left to itself the compiler waits for the pre-loop loads at their first use inside the
        // loop, with the only count that is safe on the entry edge -- vmcnt(0) -- and that wait then runs in EVERY iteration,
        // right behind the request for the next step's action: the prefetch was waited for on the spot, together with all
        // stores of the previous step (round 2: 42 % of the wave's cycles in s_waitcnt, profiles/r3_pmc_stepk.json).
#pragma unroll
        for (int j = 0; j < kDyn; ++j) asm volatile("" : "+v"(y[j]));
        asm volatile("" : "+v"(mass), "+v"(m0), "+v"(v0), "+v"(ep_ret), "+v"(ep_step));
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
#ifdef R6_TRACE_K
        tk1 = wall_clock64();
#endif

#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            // next step's action is requested before this step is computed.  The request is unconditional -- a load inside
            // a lane-masked block ends in a register merge that waits for it on the spot -- so the last step re-reads its own
            // action and the padding lanes of a partial tile read lane 0's (valid addresses, values never used).
            float n0, n1, n2;
            {
                const int kn = (k + 1 < K) ? k + 1 : k;
                const gptr<const float> act = uniform_ptr(b.action + ((int64_t)kn * sk.action_stride + e0 * R6_ACTION_DIM));
                n0 = act[3 * lane_ld + 0];
                n1 = act[3 * lane_ld + 1];
                n2 = act[3 * lane_ld + 2];
            }
            const bool emit = !last_only || k + 1 == K;   // wave-uniform: this step's outputs leave the chip

            // ---- dynamics (sim:69-143) ---------------------------------------------------------------------
            TKSTAMP(ts0);
            const Control c = make_control_normalised(p, a0, a1, a2);
            const Inertia J = make_inertia(m0);
            int nfev = 0, status;
            if constexpr (INTEG == R6_RK45) {
                status = integrate_rk45(y, mass, c, J, p.dt, nfev);
            } else {
                status = integrate<INTEG, true, ONE>(y, mass, c, J, p.dt, p.n_substeps);
                nfev = (INTEG == R6_RK4 ? 4 : 1) * (ONE ? 1 : p.n_substeps);
            }
            TKSTAMP(ts1);

            // ---- termination, reward, fused wrappers: the same sequence as step_kernel ---------------------------
            const R6Params pr = load_params();   // (kept resident in SGPRs across the K loop instead: 3.61 -> 3.77 us per step at N = 65 536)
            const bool bnd = bounds_violation(pr, y[0], y[1], y[2]);
            bool done = (status != 0) || bnd, truncated = false;
            RewardOut rw = reward<SHAPING>(pr, y, mass, c, v0);
            const bool landed = rw.landed;
            float rew = finalize_reward(pr, rw, bnd, fmaf(2.0f * c.thrust, frcp(pr.max_thrust), -1.0f), y);
            if (live && emit && b.terms) {
                float* trow = b.terms + ((int64_t)k * sk.terms_stride + i);
#pragma unroll
                for (int q = 0; q < R6_NUM_TERMS; ++q) {
                    *trow = rw.terms[q];
                    trow += ld;
                }
            }
            if (pr.clip_reward) rew = fminf(fmaxf(rew, pr.clip_lo), pr.clip_hi);
            ep_step += 1u;
            ep_ret += rew;
            if (pr.max_episode_steps > 0 && ep_step >= (uint32_t)pr.max_episode_steps) {
                truncated = !done;
                done = true;
            }

            TKSTAMP(ts2);
            // The next step's action has arrived by now (a whole step of arithmetic after its request), and so have the
            // previous step's stores: the vmcnt(0) the compiler needs before this read costs nothing HERE, in front of this
            // step's stores -- at the top of the next iteration it would wait for them (see above).  The "memory" clobber
            // keeps the stores below from being scheduled in front of it.
            asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2) : : "memory");

            // ---- per-step outputs -----------------------------------------------------------------------------------
            if (live && emit) {
                uniform_ptr(b.reward + ((int64_t)k * sk.reward_stride + e0))[lane] = rew;
                uniform_ptr(b.done + ((int64_t)k * sk.done_stride + e0))[lane] = done ? 1 : 0;
                if (b.flags)
                    uniform_ptr(b.flags + ((int64_t)k * sk.flags_stride + e0))[lane] =
                        (uint8_t)((status == 1 ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) | (truncated ? R6_FLAG_TRUNCATED : 0));
                if (b.nfev) uniform_ptr(b.nfev + ((int64_t)k * sk.nfev_stride + e0))[lane] = (uint8_t)(nfev > 255 ? 255 : nfev);
            }
            // observation (env:503-504): registers -> LDS row-major tile
            // Two forms of the AoS edge (wave-uniform choice of the launcher, R6_K_DIRECT_OBS).  Transposing: through the wave's
            // LDS tile, stored as contiguous 16 B per lane -- what a bandwidth-bound launch wants.  Direct: every lane stores its
            // own row as dwords straight from registers -- 14 scattered store instructions instead of 4 wide ones, but no LDS
            // write -> read round trip and no wave-level syncs in the step's critical path, which is what counts when a SIMD
            // holds a single wave (BASELINE configs[1]: 1024 tiles on 1024 SIMDs; per-wave stamps put the two LDS phases at
            // 1.5 k of a step's 4.9 k cycles).
            float ob[OD];
#pragma unroll
            for (int j = 0; j < kDyn; ++j) {
                if (j < OD) ob[j] = y[j] * lds_inv_norm[j];
            }
            if (OD == 14) ob[OD - 1] = mass * lds_inv_norm[13];
            if (!direct_obs) {
#pragma unroll
                for (int j = 0; j < OD; ++j) tile[lane * OD + j] = ob[j];
            }

            // ---- finished episodes: terminal outputs, statistics, and Rocket6DOF.reset (env:180-199) in the lane -----
#ifdef R6_TRACE_K
            if (__any(live && done)) ++tk_resets;
#endif
            TKSTAMP(ts2b);
            if (live && done) {
                if (b.stats) {
                    stats_add_episode(blk_stats, ep_ret, ep_step, status == 1, bnd, truncated, landed);
                }
                if (p.auto_reset) {
                    if (emit) {
                        if (b.final_obs) {
                            float* fo = b.final_obs + ((int64_t)k * sk.final_obs_stride + i * OD);
#pragma unroll
                            for (int j = 0; j < OD; ++j) fo[j] = ob[j];
                        }
                        if (b.fin_return) b.fin_return[(int64_t)k * sk.fin_stride + i] = ep_ret;
                        if (b.fin_length) b.fin_length[(int64_t)k * sk.fin_stride + i] = ep_step;
                    }
                }
            }
            TKSTAMP(ts2c);
            // Rocket6DOF.reset (env:180-199) for the envs that finished in this step, one env at a time with the whole wave
            // working on its draw (sample_ic_wave): typically ONE lane of the 64 needs it, and the in-lane form -- every
            // lane its own four Philox blocks, 63 of them masked off -- cost 1.4 us per reset at one wave per SIMD and made
            // a launch as long as its unluckiest wave (per-wave stamps, scripts/stepk_trace.py).  Same values, bit for bit.
            if (p.auto_reset) {
                uint64_t fin = __ballot(live && done);
                if (fin != 0ull) {
                    const R6Params pp = load_params();
                    do {
                        const int j = __builtin_ctzll(fin);
                        fin &= fin - 1ull;
                        float ic[R6_STATE_DIM];
                        sample_ic_wave(pp, seed, env_offset + (uint64_t)(e0 + j), step_index + (uint64_t)k + step_base, lane, ic);
                        if (lane == j) {
#pragma unroll
                            for (int q = 0; q < kDyn; ++q) y[q] = ic[q];
                            mass = m0 = ic[13];
                            v0 = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
                            ep_step = 0u;
                            ep_ret = 0.0f;
#pragma unroll
                            for (int q = 0; q < OD; ++q) ob[q] = ic[q] * lds_inv_norm[q];   // the fresh episode's first observation
                            if (!direct_obs) {
#pragma unroll
                                for (int q = 0; q < OD; ++q) tile[lane * OD + q] = ob[q];
                            }
                        }
                    } while (fin != 0ull);
                }
            }

            // ---- LDS tile -> contiguous 16 B-per-lane stores of this step's (64, OD) observation block -----------------
            TKSTAMP(ts3);
            if (direct_obs) {
                if (emit && live) {
                    float* dst = b.obs + ((int64_t)k * sk.obs_stride + i * OD);
#pragma unroll
                    for (int j = 0; j < OD; ++j) dst[j] = ob[j];
                }
            } else {
            wave_lds_sync();
            if (emit) {
                const gptr<float> dst = uniform_ptr(b.obs + ((int64_t)k * sk.obs_stride + e0 * OD));
                if (full) {
                    constexpr int n4 = 64 * OD / 4;
#pragma unroll
                    for (int r = 0; r < (n4 + 63) / 64; ++r) {
                        const int q = lane + 64 * r;
                        if (q < n4) *(gptr<f4>)(dst + 4 * q) = ld4(&tile[4 * q]);
                    }
                } else {
                    for (int q = lane; q < (int)rem * OD; q += 64) dst[q] = tile[q];
                }
            }
            wave_lds_sync();
            }
#ifdef R6_TRACE_K
            {
                TKSTAMP(ts4);
                const bool any_done = __any(live && done);
                const uint64_t d[4] = {ts2b - ts2, ts2c - ts2b, ts3 - ts2c, ts4 - ts3};   // per-step outputs + obs tile | terminal outputs | reset | obs stores
                for (int q = 0; q < 4; ++q) {
                    tk_reg[q] += d[q];
                    if (any_done) tk_reg_d[q] += d[q];
                }
            }
#endif
            a0 = n0;
            a1 = n1;
            a2 = n2;
        }

#ifdef R6_TRACE_K
        tk2 = wall_clock64();
#endif
        // ---- state of the tile: written once -----------------------------------------------------------------
        if (live) {
            float* row = b.state + e0;
#pragma unroll
            for (int j = 0; j < kDyn; ++j) {
                uniform_ptr(row)[lane] = y[j];
                row += ld;
            }
            uniform_ptr(row)[lane] = mass;
            uniform_ptr(b.epconst + e0)[lane] = m0;
            uniform_ptr(b.epconst + (ld + e0))[lane] = v0;
            if (has_ep) {
                uniform_ptr(b.ep_step + e0)[lane] = ep_step;
                uniform_ptr(b.ep_return + e0)[lane] = ep_ret;
            }
        }
    }

#ifdef R6_TRACE_K
    __builtin_amdgcn_s_waitcnt(0);
    tk3 = wall_clock64();
    if (b.stats && lane == 0) {
        double* o = b.stats + ((int64_t)blockIdx.x * kWaves + wave) * R6_NUM_STATS;   // needs grid * 4 <= R6_MAX_BLOCKS rows
        o[0] = (double)(uint32_t)tk0;
        o[1] = (double)(uint32_t)tk1;
        o[2] = (double)(uint32_t)tk2;
        o[3] = (double)(uint32_t)tk3;
        o[4] = (double)tk_resets;
        o[5] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 4);
        o[6] = (double)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        double* o2 = b.stats + ((int64_t)gridDim.x * kWaves + (int64_t)blockIdx.x * kWaves + wave) * R6_NUM_STATS;
        for (int q = 0; q < 4; ++q) {
            o2[q] = (double)tk_reg[q];
            o2[4 + q] = (double)tk_reg_d[q];
        }
    }
    return;
#endif
    __syncthreads();
    if (b.stats && tid < R6_NUM_STATS && blk_stats[R6_STAT_EPISODES] != 0ull)
        b.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += stats_value(blk_stats, tid);
}

// -------------------------------------------------------------------------------------------------
// r6_step_k for batches of at most one 64-env tile per SIMD (BASELINE configs[1]: 65 536 envs = 1024 tiles on 1024 SIMDs),
// as TWO waves per tile.
//
// With one wave on a SIMD nothing hides that wave's own latencies: per-wave stamps of step_k_kernel (scripts/stepk_trace.py) put
// a plain Euler step at 1.0 k cycles of dynamics + 2.2 k of reward + 1.3 k of outputs for ~610 VALU instructions -- a dependent
// instruction every ~6.6 cycles, half of the issue slots empty -- and nothing in the state's evolution needs the reward.  So the
// step is cut along that line and the halves run as three waves of one workgroup, placed on the same SIMD (waves w, w + 4, w + 8
// of a 768-thread workgroup), connected by a ring of slots in LDS:
//   wave A (w < 4):   action -> dynamics -> bounds / time limit -> done;  publishes the post-step state, the thrust vector, v0,
//                     the flag bits and the step count;  done / flag stores, terminal observation, Rocket6DOF.reset of finished
//                     envs (wave-wide Philox), observation -- everything the NEXT step needs, and nothing else;
//   waves B_j (w + 4, w + 8): reward (the t_go quartic: the long dependent chain) of the steps with seq % 2 == j, fused reward
//                     wrappers, reward / term stores; then, in step order (a counter in LDS), the running episode return,
//                     episode records and statistics.  Two reward waves, because the reward is the longer half: one B wave took
//                     2.0 us per step against A's 1.2 us of work.
// (Per-wave stamps, scripts/stepk_pair_trace.py: B takes 2.0 us per step, A 1.2 us of work + waiting.  Moving the reward's cheap
// half (limits, landing conditions, six of the seven terms) to A balanced the two waves at 1.9 / 2.0 us and changed nothing: the
// pair is bound by what one SIMD issues for both; and products formed in one wave cannot contract with the sum formed in the
// other, which costs the bit-exactness.  Not kept.)
// A never waits for B's results, only for a free slot; B's loads are LDS reads.  Same device functions, same order of
// operations per quantity as step_k_kernel: bit for bit the same outputs (tests/test_gpu_step_k.py runs both forms).
// Every wait is bounded: a wave that does not see its partner's counter move for ~0.1 s raises the workgroup's abort word and
// both leave (outputs then incomplete -- a partner that never arrives means the launch was broken anyway); the grid always drains.
// -------------------------------------------------------------------------------------------------
#ifndef R6_PAIR_SPIN_LOG2
#define R6_PAIR_SPIN_LOG2 21   // polls of a bounded wait (about 0.1 s); a dev build with 3 provokes the abort path (scripts/stepk_abort_check.py)
#endif
#ifndef R6_PAIR_RING
#define R6_PAIR_RING 4
#endif
#ifndef R6_PAIR_NB
#define R6_PAIR_NB 2   // reward waves per tile: wave B_j takes the steps with seq % NB == j (the reward is the longer half of a step)
#endif
#ifndef R6_PAIR_OBS_WAVE
#define R6_PAIR_OBS_WAVE 1   // a fourth wave per tile (C) takes every per-step store of the dynamics wave: observation (through an LDS tile, as wide
                             // stores), terminal observation, done / flag bytes.  0 = round 3's three waves (A stores them itself, lane by lane)
#endif
#ifndef R6_PAIR_A_LAST
#define R6_PAIR_A_LAST 0   // A/B switch (round 5): 1 = the dynamics wave computes the reward of its tile's LAST step itself -- it has nothing left to do then and
                           // the reward waves are a step or two behind.  Bit-identical (tests/test_gpu_step_k.py passes with it), and no gain: K = 4 / 16 / 64
                           // 3.91 / 2.27 / 1.61 -> 4.02 / 2.29 / 1.58 us per env step -- the reward inlined into the dynamics wave's K loop costs that wave
                           // 20 more spilled scalar registers on every step
#endif
#ifndef R6_PAIR_PRIO_A
#define R6_PAIR_PRIO_A 0   // wave priorities of the roles (s_setprio; A/B switches -- all equal by default: the arbiter serves the oldest wave, A, first)
#define R6_PAIR_PRIO_B 0
#define R6_PAIR_PRIO_C 0
#endif
constexpr int kPairTiles = 4, kPairNB = R6_PAIR_NB, kPairC = R6_PAIR_OBS_WAVE ? 1 : 0, kPairBlock = 64 * kPairTiles * (1 + kPairNB + kPairC),
              kRing = R6_PAIR_RING, kSlotWords = 21;
static_assert(kRing >= 2 * kPairNB, "the ring holds at least two steps per reward wave");
static_assert(kPairBlock <= 1024, "one workgroup");

// LDS words shared by the two waves of a pair.  The pointers carry the LDS address space explicitly: through a generic pointer
// the poll becomes a flat load, which counts in vmcnt and would make every poll wait for the wave's outstanding global loads / stores.
typedef __attribute__((address_space(3))) int* lds_word_t;
__device__ __forceinline__ bool pair_wait(lds_word_t counter, int need, lds_word_t abort_word) {
#pragma unroll 1
    for (int spin = 0; spin < (1 << R6_PAIR_SPIN_LOG2); ++spin) {
        const int v = __builtin_amdgcn_readfirstlane(__atomic_load_n(counter, __ATOMIC_RELAXED));
        if (v >= need) return true;
        if (__builtin_amdgcn_readfirstlane(__atomic_load_n(abort_word, __ATOMIC_RELAXED)) != 0) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    __atomic_store_n(abort_word, 1, __ATOMIC_RELAXED);
    return false;
}

template <int INTEG, int SHAPING, int OD>
__global__ __launch_bounds__(kPairBlock, 1) void step_k_pair_kernel(const R6Params p, const R6Buffers b, const R6StepK sk, const uint64_t seed,
                                                                    const uint64_t step_index, const uint64_t env_offset, int* host_abort) {
    static_assert(INTEG == R6_EULER || INTEG == R6_RK4, "fixed-step integrators (one sub-step per env step)");
    __shared__ float ring[kPairTiles][kRing][kSlotWords][64];
    __shared__ float ep_run[kPairTiles][64];                               // running episode return of a tile's envs, handed from reward wave to reward wave
    __shared__ int produced[kPairTiles], consumed[kPairTiles][kPairNB], returned[kPairTiles], abort_word;
    // wave C (R6_PAIR_OBS_WAVE): the first observation of the episodes wave A started in a step (rows of the reset lanes only), published
    // behind A's reset with `produced2`; C's own transposing tile; C's release of a slot.
    // (Also measured: C drawing Rocket6DOF.reset's initial conditions for EVERY env two steps ahead -- the draw depends on (seed, env,
    // step) only -- so that A merely picks them up: correct, and 2.43 -> 3.01 us per step at K = 16: ~300 instructions per step on the
    // SIMD the chain runs on, against ~70 for the draws actually needed.)
    __shared__ float rs_obs[kPairC ? kPairTiles : 1][kPairC ? kRing : 1][kPairC ? OD : 1][64];
    __shared__ __attribute__((aligned(16))) float ctile[kPairC ? kPairTiles : 1][kPairC ? 64 * OD : 4];
    __shared__ int produced2[kPairTiles], consumed_c[kPairTiles];
    __shared__ float lds_inv_norm[R6_STATE_DIM];
    __shared__ float lds_ic[2][16];     // Rocket6DOF.reset's distribution: low end, width (sample_ic_wave_lds)
    __shared__ stat_t blk_stats[R6_NUM_STATS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool role_c = kPairC != 0 && wave >= kPairTiles * (1 + kPairNB);
    const bool role_b = wave >= kPairTiles && !role_c;
    const int slot_of = wave % kPairTiles;                      // the tile slot within the workgroup (waves t, t + 4, t + 8 (, t + 12) share SIMD t)
    const int par = role_b ? wave / kPairTiles - 1 : 0;         // reward wave j of the tile
    const int64_t n = b.n, ld = b.ld;
    const int64_t n_tiles = (n + 63) >> 6;
    const int64_t nt = (int64_t)gridDim.x * kPairTiles;
    const bool has_ep = b.ep_step != nullptr && b.ep_return != nullptr;
    const int K = sk.k;
    const bool last_only = (sk.flags & R6_K_LAST_ONLY) != 0;

#ifdef R6_TRACE_K
    const uint64_t pk0 = wall_clock64();
    uint64_t pk1 = 0, pk2 = 0, pk_wait = 0;
    int pk_ends = 0;
#define PKWAIT(expr) ({ const uint64_t w0_ = __builtin_amdgcn_s_memtime(); const bool r_ = (expr); pk_wait += __builtin_amdgcn_s_memtime() - w0_; r_; })
#else
#define PKWAIT(expr) (expr)
#endif
    if (tid < R6_NUM_STATS) blk_stats[tid] = 0ull;
    if (tid < R6_STATE_DIM) lds_inv_norm[tid] = p.inv_norm[tid];
    if (tid < 16) {
        lds_ic[0][tid] = tid < R6_STATE_DIM ? p.ic_lo[tid < R6_STATE_DIM ? tid : 0] : 0.0f;
        lds_ic[1][tid] = tid < R6_STATE_DIM ? p.ic_range[tid < R6_STATE_DIM ? tid : 0] : 0.0f;
    }
    if (tid < kPairTiles) {
        produced[tid] = 0;
        returned[tid] = 0;
        produced2[tid] = 0;
        consumed_c[tid] = 0;
#pragma unroll
        for (int j = 0; j < kPairNB; ++j) consumed[tid][j] = 0;
    }
    if (tid == 0) abort_word = 0;
    const uint64_t step_base = b.step_base ? *b.step_base : (uint64_t)0;
    __syncthreads();

    if (R6_PAIR_PRIO_A | R6_PAIR_PRIO_B | R6_PAIR_PRIO_C) {
        if (role_c) __builtin_amdgcn_s_setprio(R6_PAIR_PRIO_C);
        else if (role_b) __builtin_amdgcn_s_setprio(R6_PAIR_PRIO_B);
        else __builtin_amdgcn_s_setprio(R6_PAIR_PRIO_A);
    }
    int seq = 0;   // steps this pair has gone through (its tiles one after the other): what the ring's counters count
    bool ok = true;   // false: the partner never arrived (abort): leave the loops, still meet the workgroup at its last barrier
#pragma unroll 1
    for (int64_t t = (int64_t)blockIdx.x * kPairTiles + slot_of; ok && t < n_tiles; t += nt) {
        const int64_t e0 = t * 64;
        const int64_t rem = n - e0;
        const bool live = lane < rem;
        const int64_t i = e0 + lane;
        const int lane_ld = live ? lane : 0;
        if (!role_b && !role_c) {
            // ================================= wave A: what the next step needs ==========================================
            // Each role re-reads the kernel arguments IT uses from the kernarg segment, inside its own branch (round 5): taken from
            // the function's parameters they are live through all three branches of the tile loop -- the scalar-register pressure
            // of the kernel is then the UNION of what the roles need, and the overflow came back as v_readlane / v_writelane
            // (vector instructions: 118 of the ~990 a tile's four waves issue per env step).
            const R6Params p = load_params();
            const R6Buffers b = load_buffers();
            const R6StepK sk = load_stepk();
            float y[kDyn], mass = 4.0e4f, m0 = 4.0e4f, v0 = 0.0f;
            uint32_t ep_step = 0u;
#pragma unroll
            for (int j = 0; j < kDyn; ++j) y[j] = (j == 0) ? 1000.0f : ((j == 6) ? 1.0f : 0.0f);   // benign padding lane
            if (live) {
                const float* row = b.state + e0;
#pragma unroll
                for (int j = 0; j < kDyn; ++j) {
                    y[j] = uniform_ptr(row)[lane];
                    row += ld;
                }
                mass = uniform_ptr(row)[lane];
                m0 = uniform_ptr(b.epconst + e0)[lane];
                v0 = uniform_ptr(b.epconst + (ld + e0))[lane];
                if (has_ep) ep_step = uniform_ptr(b.ep_step + e0)[lane];
            }
#if R6_PAIR_A_LAST
            float ep_ret_in = 0.0f;      // needed only when the tile's last step is also its first (K == 1)
            if (live && has_ep && K == 1) ep_ret_in = uniform_ptr(b.ep_return + e0)[lane];
#endif
            float a0, a1, a2;
            {
                const gptr<const float> act = uniform_ptr(b.action + e0 * R6_ACTION_DIM);
                a0 = act[3 * lane_ld + 0];
                a1 = act[3 * lane_ld + 1];
                a2 = act[3 * lane_ld + 2];
            }
#pragma unroll
            for (int j = 0; j < kDyn; ++j) asm volatile("" : "+v"(y[j]));
            asm volatile("" : "+v"(mass), "+v"(m0), "+v"(v0), "+v"(ep_step));
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
#ifdef R6_TRACE_K
            pk1 = wall_clock64();
#endif
#pragma unroll 1
            for (int k = 0; k < K; ++k, ++seq) {
                float n0, n1, n2;
                {
                    const int kn = (k + 1 < K) ? k + 1 : k;
                    const gptr<const float> act = uniform_ptr(b.action + ((int64_t)kn * sk.action_stride + e0 * R6_ACTION_DIM));
                    n0 = act[3 * lane_ld + 0];
                    n1 = act[3 * lane_ld + 1];
                    n2 = act[3 * lane_ld + 2];
                }
                const bool emit = !last_only || k + 1 == K;
                const Control c = make_control_normalised(p, a0, a1, a2);
                const Inertia J = make_inertia(m0);
                const int status = integrate<INTEG, true, true>(y, mass, c, J, p.dt, 1);
                const bool bnd = bounds_violation(p, y[0], y[1], y[2]);
                bool done = (status != 0) || bnd, truncated = false;
                ep_step += 1u;
                if (p.max_episode_steps > 0 && ep_step >= (uint32_t)p.max_episode_steps) {
                    truncated = !done;
                    done = true;
                }
                // ---- publish the step to wave B ----
                {   // the slot about to be overwritten held step seq - kRing: its reward wave must have read it
                    const int sp = seq - kRing;
                    if (sp >= 0 && !PKWAIT(pair_wait((lds_word_t)&consumed[slot_of][sp % kPairNB], sp / kPairNB + 1, (lds_word_t)&abort_word))) {
                        ok = false;
                        break;
                    }
                    if (kPairC && sp >= 0 && !PKWAIT(pair_wait((lds_word_t)&consumed_c[slot_of], sp + 1, (lds_word_t)&abort_word))) {
                        ok = false;
                        break;
                    }
                }
                {
                    float(*sl)[64] = ring[slot_of][seq % kRing];
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) sl[j][lane] = y[j];
                    sl[13][lane] = mass;
                    sl[14][lane] = c.tbx;
                    sl[15][lane] = c.tby;
                    sl[16][lane] = c.tbz;
                    sl[17][lane] = c.thrust;
                    sl[18][lane] = v0;
                    sl[19][lane] = __int_as_float((status == 1 ? 1 : 0) | (bnd ? 2 : 0) | (truncated ? 4 : 0) | (done ? 8 : 0));
                    sl[20][lane] = __uint_as_float(ep_step);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __atomic_store_n((lds_word_t)&produced[slot_of], seq + 1, __ATOMIC_RELAXED);

#if R6_PAIR_A_LAST
                if (k + 1 == K) {
                    // ---- the LAST step's reward and records: wave B's sequence, operation for operation, on the registers of this wave ----
                    const R6Params pr = load_params();
                    RewardOut rw = reward<SHAPING>(pr, y, mass, c, v0);
                    const bool landed = rw.landed;
                    float rew = finalize_reward(pr, rw, bnd, fmaf(2.0f * c.thrust, frcp(pr.max_thrust), -1.0f), y);
                    if (live && b.terms) {
                        float* trow = b.terms + ((int64_t)k * sk.terms_stride + i);
#pragma unroll
                        for (int q = 0; q < R6_NUM_TERMS; ++q) {
                            *trow = rw.terms[q];
                            trow += ld;
                        }
                    }
                    if (pr.clip_reward) rew = fminf(fmaxf(rew, pr.clip_lo), pr.clip_hi);
                    if (live) uniform_ptr(b.reward + ((int64_t)k * sk.reward_stride + e0))[lane] = rew;
                    if (!PKWAIT(pair_wait((lds_word_t)&returned[slot_of], seq, (lds_word_t)&abort_word))) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    float ep_ret = (k == 0) ? ep_ret_in : ep_run[slot_of][lane];
                    ep_ret += rew;
                    if (live && done) {
                        if (b.stats) stats_add_episode(blk_stats, ep_ret, ep_step, status == 1, bnd, truncated, landed);
                        if (pr.auto_reset) {
                            if (b.fin_return) b.fin_return[(int64_t)k * sk.fin_stride + i] = ep_ret;
                            if (b.fin_length) b.fin_length[(int64_t)k * sk.fin_stride + i] = ep_step;
                            ep_ret = 0.0f;
                        }
                    }
                    if (live && has_ep) uniform_ptr(b.ep_return + e0)[lane] = ep_ret;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __atomic_store_n((lds_word_t)&returned[slot_of], seq + 1, __ATOMIC_RELAXED);
                }
#endif

                if constexpr (kPairC != 0) {
                    // ---- wave C stores this step's outputs; A only starts the next episodes (Rocket6DOF.reset) and hands C their first
                    //      observation: no global store of A's inside the K loop ----
                    asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2));   // the prefetch has arrived
                    if (p.auto_reset) {
                        uint64_t fin = __ballot(live && done);
                        if (fin != 0ull) {
#ifdef R6_TRACE_K
                            ++pk_ends;
#endif
                            float(*rs)[64] = rs_obs[slot_of][seq % kRing];
#ifdef R6_PAIR_PRIO_RESET
                            __builtin_amdgcn_s_setprio(R6_PAIR_PRIO_RESET);   // the draw is the longest thing on the chain of the tile's K steps
#endif
                            do {
                                const int j = __builtin_ctzll(fin);
                                fin &= fin - 1ull;
                                float ic[R6_STATE_DIM];
                                sample_ic_wave_lds(lds_ic[0], lds_ic[1], seed, env_offset + (uint64_t)(e0 + j), step_index + (uint64_t)k + step_base, lane, ic);
                                if (lane == j) {
#pragma unroll
                                    for (int q = 0; q < kDyn; ++q) y[q] = ic[q];
                                    mass = m0 = ic[13];
                                    v0 = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
                                    ep_step = 0u;
#pragma unroll
                                    for (int q = 0; q < OD; ++q) rs[q][lane] = ic[q] * lds_inv_norm[q];
                                }
                            } while (fin != 0ull);
#ifdef R6_PAIR_PRIO_RESET
                            __builtin_amdgcn_s_setprio(R6_PAIR_PRIO_A);
#endif
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            if (lane == 0) __atomic_store_n((lds_word_t)&produced2[slot_of], seq + 1, __ATOMIC_RELAXED);
                        }
                    }
                } else {
                // ---- observation, terminal observation, reset ----
                float ob[OD];
#pragma unroll
                for (int j = 0; j < kDyn; ++j) {
                    if (j < OD) ob[j] = y[j] * lds_inv_norm[j];
                }
                if (OD == 14) ob[OD - 1] = mass * lds_inv_norm[13];
                asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2) : : "memory");   // the prefetch has arrived: waited for in front of the stores
                if (live && emit) {
                    uniform_ptr(b.done + ((int64_t)k * sk.done_stride + e0))[lane] = done ? 1 : 0;
                    if (b.flags)
                        uniform_ptr(b.flags + ((int64_t)k * sk.flags_stride + e0))[lane] =
                            (uint8_t)((status == 1 ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) | (truncated ? R6_FLAG_TRUNCATED : 0));
                    if (b.nfev) uniform_ptr(b.nfev + ((int64_t)k * sk.nfev_stride + e0))[lane] = (uint8_t)(INTEG == R6_RK4 ? 4 : 1);
                }
                if (live && done && p.auto_reset && emit && b.final_obs) {
                    float* fo = b.final_obs + ((int64_t)k * sk.final_obs_stride + i * OD);
#pragma unroll
                    for (int j = 0; j < OD; ++j) fo[j] = ob[j];
                }
                if (p.auto_reset) {
                    uint64_t fin = __ballot(live && done);
                    if (fin != 0ull) {
#ifdef R6_TRACE_K
                        ++pk_ends;
#endif
                        const R6Params pp = load_params();
                        do {
                            const int j = __builtin_ctzll(fin);
                            fin &= fin - 1ull;
                            float ic[R6_STATE_DIM];
                            sample_ic_wave(pp, seed, env_offset + (uint64_t)(e0 + j), step_index + (uint64_t)k + step_base, lane, ic);
                            if (lane == j) {
#pragma unroll
                                for (int q = 0; q < kDyn; ++q) y[q] = ic[q];
                                mass = m0 = ic[13];
                                v0 = sqrtf(fmaf(ic[3], ic[3], fmaf(ic[4], ic[4], ic[5] * ic[5])));
                                ep_step = 0u;
#pragma unroll
                                for (int q = 0; q < OD; ++q) ob[q] = ic[q] * lds_inv_norm[q];
                            }
                        } while (fin != 0ull);
                    }
                }
                if (emit && live) {
                    float* dst = b.obs + ((int64_t)k * sk.obs_stride + i * OD);
#pragma unroll
                    for (int j = 0; j < OD; ++j) dst[j] = ob[j];
                }
                }   // !kPairC
                a0 = n0;
                a1 = n1;
                a2 = n2;
            }
#ifdef R6_TRACE_K
            pk2 = wall_clock64();
#endif
            if (live) {
                float* row = b.state + e0;
#pragma unroll
                for (int j = 0; j < kDyn; ++j) {
                    uniform_ptr(row)[lane] = y[j];
                    row += ld;
                }
                uniform_ptr(row)[lane] = mass;
                uniform_ptr(b.epconst + e0)[lane] = m0;
                uniform_ptr(b.epconst + (ld + e0))[lane] = v0;
                if (has_ep) uniform_ptr(b.ep_step + e0)[lane] = ep_step;
            }
        } else if (role_c) {
            // ================================= wave C: this step's stores =================================================
            // done / flag bytes, terminal observation, observation (the fresh episode's first one for a lane wave A has reset) -- from
            // the state wave A published, through this wave's own LDS tile as contiguous 16-byte stores.  Nothing waits for C but the
            // slot's next use: the stores (round 3: lane by lane from wave A, 0.9 of a step's 2.4 us) are off the chain of the K steps.
            const R6Params p = load_params();
            const R6Buffers b = load_buffers();
            const R6StepK sk = load_stepk();
            float* const tile = ctile[slot_of];
            const bool full = rem >= 64;
#ifdef R6_TRACE_K
            pk1 = wall_clock64();
#endif
#pragma unroll 1
            for (int k = 0; k < K; ++k, ++seq) {
                const bool emit = !last_only || k + 1 == K;
                if (!PKWAIT(pair_wait((lds_word_t)&produced[slot_of], seq + 1, (lds_word_t)&abort_word))) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float ob[OD];
                int fl;
                {
                    const float(*sl)[64] = ring[slot_of][seq % kRing];
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) {
                        if (j < OD) ob[j] = sl[j][lane] * lds_inv_norm[j];
                    }
                    if (OD == 14) ob[OD - 1] = sl[13][lane] * lds_inv_norm[13];
                    fl = __float_as_int(sl[19][lane]);
                }
                const bool ground = (fl & 1) != 0, bnd = (fl & 2) != 0, truncated = (fl & 4) != 0, done = (fl & 8) != 0;
                if (live && emit) {
                    uniform_ptr(b.done + ((int64_t)k * sk.done_stride + e0))[lane] = done ? 1 : 0;
                    if (b.flags)
                        uniform_ptr(b.flags + ((int64_t)k * sk.flags_stride + e0))[lane] =
                            (uint8_t)((ground ? R6_FLAG_GROUND : 0) | (bnd ? R6_FLAG_BOUNDS : 0) | (truncated ? R6_FLAG_TRUNCATED : 0));
                    if (b.nfev) uniform_ptr(b.nfev + ((int64_t)k * sk.nfev_stride + e0))[lane] = (uint8_t)(INTEG == R6_RK4 ? 4 : 1);
                }
                if (live && done && p.auto_reset && emit && b.final_obs) {
                    float* fo = b.final_obs + ((int64_t)k * sk.final_obs_stride + i * OD);
#pragma unroll
                    for (int j = 0; j < OD; ++j) fo[j] = ob[j];
                }
                if (p.auto_reset && __any(live && done)) {   // wave A has reset these lanes: their row is the new episode's first observation
                    if (!PKWAIT(pair_wait((lds_word_t)&produced2[slot_of], seq + 1, (lds_word_t)&abort_word))) {
                        ok = false;
                        break;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    if (live && done) {
                        const float(*rs)[64] = rs_obs[slot_of][seq % kRing];
#pragma unroll
                        for (int q = 0; q < OD; ++q) ob[q] = rs[q][lane];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the slot's reads are done
                if (lane == 0) __atomic_store_n((lds_word_t)&consumed_c[slot_of], seq + 1, __ATOMIC_RELAXED);
#pragma unroll
                for (int j = 0; j < OD; ++j) tile[lane * OD + j] = ob[j];
                wave_lds_sync();
                if (emit) {
                    const gptr<float> dst = uniform_ptr(b.obs + ((int64_t)k * sk.obs_stride + e0 * OD));
                    if (full) {
                        constexpr int n4 = 64 * OD / 4;
#pragma unroll
                        for (int r = 0; r < (n4 + 63) / 64; ++r) {
                            const int q = lane + 64 * r;
                            if (q < n4) *(gptr<f4>)(dst + 4 * q) = ld4(&tile[4 * q]);
                        }
                    } else {
                        for (int q = lane; q < (int)rem * OD; q += 64) dst[q] = tile[q];
                    }
                }
                wave_lds_sync();
            }
#ifdef R6_TRACE_K
            pk2 = wall_clock64();
#endif
        } else {
            // ================================= wave B: reward, records =====================================================
            const R6Params p = load_params();
            const R6Buffers b = load_buffers();
            const R6StepK sk = load_stepk();
#ifdef R6_TRACE_K
            pk1 = wall_clock64();
#endif
            // The episode return the tile comes in with: requested and ARRIVED in front of the K loop (round 5).  Read inside the
            // loop (`if (k == 0)`) it left one s_waitcnt vmcnt(0) on the merged path of every step -- and gfx9 counts loads and
            // stores in ONE in-order counter: every step then waited for the acknowledgement of its own reward store (and the terms
            // stores), ~0.5 us of a reward wave's 2.0 us per step.
            float ep_ret_in = 0.0f;
            if (live && has_ep) ep_ret_in = uniform_ptr(b.ep_return + e0)[lane];
            asm volatile("" : "+v"(ep_ret_in));
#pragma unroll 1
            for (int k = 0; k < K; ++k, ++seq) {
                if (seq % kPairNB != par) continue;   // another reward wave's step
#if R6_PAIR_A_LAST
                if (k + 1 == K) {   // the dynamics wave's (see there): nothing to read, the slot is free as far as this wave is concerned
                    if (lane == 0) __atomic_store_n((lds_word_t)&consumed[slot_of][par], seq / kPairNB + 1, __ATOMIC_RELAXED);
                    continue;
                }
#endif
                const bool emit = !last_only || k + 1 == K;
                if (!PKWAIT(pair_wait((lds_word_t)&produced[slot_of], seq + 1, (lds_word_t)&abort_word))) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float y[kDyn], mass, v0;
                Control c;
                int fl;
                uint32_t ep_step;
                {
                    const float(*sl)[64] = ring[slot_of][seq % kRing];
#pragma unroll
                    for (int j = 0; j < kDyn; ++j) y[j] = sl[j][lane];
                    mass = sl[13][lane];
                    c.tbx = sl[14][lane];
                    c.tby = sl[15][lane];
                    c.tbz = sl[16][lane];
                    c.thrust = sl[17][lane];
                    c.mdot = 0.0f;   // not used by the reward
                    v0 = sl[18][lane];
                    fl = __float_as_int(sl[19][lane]);
                    ep_step = __float_as_uint(sl[20][lane]);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // the reads above are done (LDS operations of a wave complete in order)
                if (lane == 0) __atomic_store_n((lds_word_t)&consumed[slot_of][par], seq / kPairNB + 1, __ATOMIC_RELAXED);
                const bool ground = (fl & 1) != 0, bnd = (fl & 2) != 0, truncated = (fl & 4) != 0, done = (fl & 8) != 0;

                const R6Params pr = load_params();
                RewardOut rw = reward<SHAPING>(pr, y, mass, c, v0);
                const bool landed = rw.landed;
                float rew = finalize_reward(pr, rw, bnd, fmaf(2.0f * c.thrust, frcp(pr.max_thrust), -1.0f), y);
                if (live && emit && b.terms) {
                    float* trow = b.terms + ((int64_t)k * sk.terms_stride + i);
#pragma unroll
                    for (int q = 0; q < R6_NUM_TERMS; ++q) {
                        *trow = rw.terms[q];
                        trow += ld;
                    }
                }
                if (pr.clip_reward) rew = fminf(fmaxf(rew, pr.clip_lo), pr.clip_hi);
                if (live && emit) uniform_ptr(b.reward + ((int64_t)k * sk.reward_stride + e0))[lane] = rew;
                // ---- the running episode return: a chain through the steps, handed from reward wave to reward wave in LDS ----
                if (!PKWAIT(pair_wait((lds_word_t)&returned[slot_of], seq, (lds_word_t)&abort_word))) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                float ep_ret = (k == 0) ? ep_ret_in : ep_run[slot_of][lane];
                ep_ret += rew;
                if (live && done) {
                    if (b.stats) stats_add_episode(blk_stats, ep_ret, ep_step, ground, bnd, truncated, landed);
                    if (p.auto_reset) {
                        if (emit) {
                            if (b.fin_return) b.fin_return[(int64_t)k * sk.fin_stride + i] = ep_ret;
                            if (b.fin_length) b.fin_length[(int64_t)k * sk.fin_stride + i] = ep_step;
                        }
                        ep_ret = 0.0f;
                    }
                }
                if (k + 1 == K) {
                    if (live && has_ep) uniform_ptr(b.ep_return + e0)[lane] = ep_ret;
                } else {
                    ep_run[slot_of][lane] = ep_ret;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) __atomic_store_n((lds_word_t)&returned[slot_of], seq + 1, __ATOMIC_RELAXED);
            }
#ifdef R6_TRACE_K
            pk2 = wall_clock64();
#endif
        }
    }
#ifdef R6_TRACE_K
    __builtin_amdgcn_s_waitcnt(0);
    if (b.stats && lane == 0) {
        double* o = b.stats + ((int64_t)blockIdx.x * (kPairBlock / 64) + wave) * R6_NUM_STATS;
        o[0] = (double)(uint32_t)pk0;
        o[1] = (double)(uint32_t)pk1;
        o[2] = (double)(uint32_t)pk2;
        o[3] = (double)(uint32_t)wall_clock64();
        o[4] = (double)pk_ends;
        o[5] = (double)pk_wait;
        o[6] = role_c ? 2.0 : (role_b ? 1.0 : 0.0);
    }
    return;
#endif

    __syncthreads();
    if (b.stats && tid < R6_NUM_STATS && blk_stats[R6_STAT_EPISODES] != 0ull)
        b.stats[(int64_t)blockIdx.x * R6_NUM_STATS + tid] += stats_value(blk_stats, tid);
    // an abort leaves this workgroup's outputs incomplete: say so where the host sees it (a word of mapped host memory, read by
    // r6_device_status(), r6_last_error() and the next r6_step / r6_step_k / r6_rollout call) -- never silently
    if (tid == 0 && host_abort && abort_word != 0) __hip_atomic_store(host_abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

#ifdef R6_TEST_HOOKS
__global__ __launch_bounds__(kPairBlock, 1) void wave_placement_kernel(uint32_t* out) {
    __shared__ float pad[kPairTiles][kRing][kSlotWords][64];   // the pair kernel's LDS footprint: one workgroup per CU, as there
    if (threadIdx.x == 0) pad[0][0][0][0] = 0.0f;
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        out[((int64_t)blockIdx.x * (kPairBlock / 64) + wave) * 2 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        out[((int64_t)blockIdx.x * (kPairBlock / 64) + wave) * 2 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
}
#endif

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
    reset_env_global<OD>(p, b, i, seed, step_index, env_offset);
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
    int st, nfev = 0;
    if constexpr (INTEG == R6_RK45) st = integrate_rk45(y, mass, c, J, p.dt, nfev);
    else st = integrate<INTEG, true>(y, mass, c, J, p.dt, p.n_substeps);
#pragma unroll
    for (int j = 0; j < kDyn; ++j) state[j * ld + i] = y[j];
    state[13 * ld + i] = mass;
    if (status) status[i] = (uint8_t)(st == 1 ? 1 : (st != 0 ? 255 : 0));   // 255 = solver failure (scipy status -1)
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
    rhs13(y, 1.0f / mass, c, J, dy);
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
    if (b->n >= (int64_t)1 << 31) return fail(R6_E_BADARG, "n must be < 2^31 per shard");
    if (p->obs_dim != 13 && p->obs_dim != 14) return fail(R6_E_BADARG, "obs_dim must be 13 or 14");
    if (p->integrator != R6_EULER && p->integrator != R6_RK4 && p->integrator != R6_RK45) return fail(R6_E_BADARG, "bad integrator");
    if (p->n_substeps < 1 || p->n_substeps > 4096) return fail(R6_E_BADARG, "n_substeps out of range");
    if (p->shaping != R6_SHAPING_ACCELERATION && p->shaping != R6_SHAPING_VELOCITY)
        return fail(R6_E_BADARG, "bad shaping");
    if (!(p->dt > 0.0f)) return fail(R6_E_BADARG, "dt must be > 0");
    return 0;
}

inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + r6::kBlock - 1) / r6::kBlock)); }

// Resident capacity (workgroups) of the current device for one step-kernel instance, cached.
template <typename K>
int resident_blocks(K kernel, int slot) {
    static std::atomic<int> cache[16][48];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int v = cache[dev][slot].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, r6::kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    (void)hipGetLastError();
    v = per_cu * cus;
    if (const char* e = std::getenv("R6_GRID_BLOCKS")) {  // tuning knob
        const int o = std::atoi(e);
        if (o > 0) v = o;
    }
    if (v > R6_MAX_BLOCKS) v = R6_MAX_BLOCKS;
    cache[dev][slot].store(v, std::memory_order_relaxed);
    return v;
}

// True when the current device shows all 8 XCDs of an MI355X as one device (SPX mode: 256 CUs), which is what the
// XCD-parity split assumes (workgroup i on XCD i % 8).  A partitioned device (CPX: 32 CUs = one XCD) gets equal shares.
bool device_has_eight_xcds() {
    static std::atomic<int> cache[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        (void)hipGetLastError();
        v = (cus == 256) ? 1 : -1;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v > 0;
}

// Share of the tiles handed to the workgroups on even XCDs, in 1/65536 (0 = no split).  Defaults from a sweep at
// N = 2^20 (0.50 / 0.52 / 0.54 / 0.56: RK4 39.5 / 38.8 / 38.3 / 38.6 us, fused stack 43.6 / 42.8 / 41.8 / 41.5, Euler 35.2 /
// 35.3 / 35.2 / 34.1): the even XCDs of an MI355X finish equal shares ~1.19x sooner (load latency, see step_kernel).
// R6_EVEN_XCD_SHARE=<fraction> overrides (0 or 0.5 = equal shares); results do not depend on it, only which wave
// computes which tile.
#ifndef R6_EVEN_SHARE_DEFAULT
#define R6_EVEN_SHARE_DEFAULT -1.0   // < 0: per-integrator default (RK4 0.54, Euler 0.56)
#endif
// ---- launch shape of r6_step: tuning state (defaults from measurements; R6_* environment variables and r6_tune()
//      override them -- results never depend on any of it, only which wave computes which tile) ----------------------------
struct StepTune {
    std::atomic<double> even_share{R6_EVEN_SHARE_DEFAULT};   // < 0: per-integrator default
    std::atomic<int> grid_blocks{0};                         // > 0: pins the persistent grid (R6_GRID_BLOCKS)
    std::atomic<int> sweep{R6_SWEEP_DEFAULT};
    std::atomic<int> sweep_flags{0};                         // A/B bit passed to the kernel: 4 = no priority ladder
    std::atomic<int> balanced_tpw{0};                        // tiles per wave of the dispatcher-balanced grid (large shards); 0 = by shard size
    std::atomic<int> nt_mode{-1};                            // -1: streaming stores by shard size (default), 0: never, 1: always
    std::atomic<int> rk45_defer{R6_RK45_DEFER_DEFAULT};      // adaptive instances: a wave hands at most this many unfinished lanes to the straggler pass (0 = never)
    StepTune() {
        if (const char* e = std::getenv("R6_EVEN_XCD_SHARE")) even_share = std::atof(e);
        if (const char* e = std::getenv("R6_GRID_BLOCKS")) grid_blocks = std::atoi(e);
        if (const char* e = std::getenv("R6_SWEEP")) sweep = std::atoi(e);
        if (const char* e = std::getenv("R6_BALANCED_TPW")) balanced_tpw = std::atoi(e);
        if (const char* e = std::getenv("R6_RK45_DEFER")) rk45_defer = std::atoi(e);
    }
};
StepTune& tune() {
    static StepTune t;
    return t;
}

uint32_t even_xcd_share_q16(int integrator) {
    const double forced = tune().even_share.load(std::memory_order_relaxed);
    // (the adaptive instances are bound by arithmetic, not by load latency: equal shares -- 117.5 -> 109.9 us at 2^20 envs, round 4)
    const double f = forced >= 0.0 ? forced : (integrator == R6_EULER ? 0.56 : (integrator == R6_RK45 ? 0.5 : 0.54));
    if (!(f > 0.0) || f == 0.5 || f >= 1.0) return 0u;
    return (uint32_t)(f * 65536.0 + 0.5);
}

// One launch of the step kernel over a whole shard.
template <int INTEG, int SHAPING, int OD>
void launch_step_inst(const R6Params& p, const R6Buffers& b, uint64_t seed, uint64_t step, uint64_t off, hipStream_t s) {
    auto kernel = r6::step_kernel<INTEG, SHAPING, OD>;
    if constexpr (INTEG != R6_RK45) {
        // ONE: n_substeps == 1 (the reference's configuration) gets the straight-line integrator instance;
        // NT: non-temporal output stores for working sets far beyond the Infinity Cache
        const int nt_mode = tune().nt_mode.load(std::memory_order_relaxed);
        // ... and for the window in which the rows the next step re-reads (state, m0: 128 B per env with the action) fit the
        // 256 MB Infinity Cache only if the outputs do not pass through it: measured on one set of buffers (round 4,
        // profiles/r4_experiments.json; plain | streaming stores, us per step): 1.31 M envs 44.5 | 42.9, 1.57 M 59.9 | 51.2,
        // 1.84 M 76.1 | 61.1, 2^21 81.2 | 73.0 -- and 2.62 M 98.9 | 106.6, 3.15 M 110.6 | 113.8, 2^22 146.1 | 148.7, 5.2 M and up equal.
        const bool nt = nt_mode >= 0 ? nt_mode != 0 : (b.n >= R6_NT_MIN_ENVS || (b.n >= R6_NT_WINDOW_LO && b.n < R6_NT_WINDOW_HI));
        if (p.n_substeps == 1) kernel = nt ? r6::step_kernel<INTEG, SHAPING, OD, true, true> : r6::step_kernel<INTEG, SHAPING, OD, false, true>;
        else if (nt) kernel = r6::step_kernel<INTEG, SHAPING, OD, true, false>;
    }
    // occupancy cache slot: the sub-stepped RK4 instances have their own launch bounds
    const int slot = INTEG * 4 + SHAPING * 2 + (OD == 14 ? 1 : 0) + ((INTEG == R6_RK4 && p.n_substeps != 1) ? 32 : 0);
    const int64_t n_tiles = (b.n + 63) / 64;
    int64_t blocks = (n_tiles + r6::kWaves - 1) / r6::kWaves;
    int64_t cap = resident_blocks(kernel, slot);
    const int pinned_blocks = tune().grid_blocks.load(std::memory_order_relaxed);
    if (pinned_blocks > 0) cap = pinned_blocks;
    if (cap > R6_MAX_BLOCKS) cap = R6_MAX_BLOCKS;
    bool balanced = false;   // dispatcher-balanced grid (below)
    if (blocks > cap) {
        // Up to ~3 M envs the grid is the resident capacity (persistent waves, ~3-10 tiles each).  Beyond that the
        // hardware dispatcher balances better than a fixed tile list per wave: two tiles per wave, as many
        // workgroups as the per-workgroup statistics rows allow (N = 2^22: 4.6 -> 5.1 TB/s algorithmic, 2^23:
        // 3.8 -> 4.6, 2^24: 3.8 -> 5.1; at 2^20 / 2^21 the resident grid is 4-13 % faster).
        // two tiles per wave; four from 6.3 M envs on (round 4, one set of buffers, tiles per wave 2 | 3 | 4 | 5 | 8, us per step:
        // 2^22 envs 156.7 | 157.2 | 165.1 | 162.9 | 176.7; 6.3 M 228.5 | 230.1 | 229.0 | 233.6 | 236.1; 2^23 298.3 | 297.8 | 292.2 |
        // 303.8 | 309.6 (three sessions: -3.8, -9.4, -6.1 us for four); 12.6 M 472.8 | 473.6 | 474.2 | 485.9 | 493.6)
        int tpw = tune().balanced_tpw.load(std::memory_order_relaxed);
        // (adaptive instances: four at every size -- a wave's stragglers ride in its later tiles: 4 M envs 433.4 -> 397.9 us, round 5)
        if (tpw < 1) tpw = (INTEG == R6_RK45 || b.n >= R6_NT_MIN_ENVS) ? 4 : 2;
        const int64_t big = n_tiles / (tpw * r6::kWaves);
        balanced = pinned_blocks <= 0 && n_tiles >= 49152 && big > cap;
        blocks = balanced ? (big < R6_MAX_BLOCKS ? big : R6_MAX_BLOCKS) : cap;
    }
    const bool persistent = blocks == cap && n_tiles >= 2 * blocks * r6::kWaves;
    // XCD-aware split (see step_kernel): only for the persistent grid, where every wave walks several tiles
    uint32_t even_share = 0;
    if (persistent && (blocks % 8) == 0 && device_has_eight_xcds()) even_share = even_xcd_share_q16(INTEG);
    // Order of the tiles on odd steps (see step_kernel).  Measured in one process at N = 2^20 / 2^21 / 2^22 (us per step;
    // as even steps | every wave's list backwards | whole range mirrored): RK4 38.0 | 39.4 | 38.8, 85.2 | 90.8 | 90.1,
    // 154 | 159 | 162 (the last size runs the dispatcher-balanced grid, where only "mirrored" exists: its three numbers are
    // the same code and show the +-3 % between allocations); Euler 38.0 | 36.8 | 36.5, 84 | 94 | 94, 160 | 150 | 154.
    // So: the RK4 / RK45 instances keep one order on the resident grid, the Euler instances walk their lists backwards,
    // and the dispatcher-balanced grid mirrors the range (N = 2^23: 328 -> 291 us with the streaming stores, round 1).
    // R6_SWEEP=0|1|2 overrides the resident-grid choice.
    const int sweep_env = tune().sweep.load(std::memory_order_relaxed);
    const int sweep_mode = sweep_env >= 0 ? sweep_env : (INTEG == R6_EULER ? 1 : 0);
    uint32_t sweep = 0;
    if (step & 1) sweep = (blocks > cap || sweep_mode == 2) ? 2u : (sweep_mode == 1 ? 1u : 0u);
    sweep |= (uint32_t)tune().sweep_flags.load(std::memory_order_relaxed) & 4u;
    if constexpr (INTEG == R6_RK45) {   // bits 8..15: the straggler hand-over threshold (see step_kernel)
        int dm = tune().rk45_defer.load(std::memory_order_relaxed);
        dm = dm < 0 ? 0 : (dm > 64 ? 64 : dm);
        sweep |= (uint32_t)dm << 8;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(r6::kBlock), 0, s, p, b, seed, step, off, even_share, sweep);
}

// The device error word: step_k_pair_kernel raises it when one of its bounded waits ran out.  Mapped host memory, allocated by the
// first stepping entry point (r6_reset / r6_step / r6_step_k / r6_rollout) that runs outside a stream capture (an allocation
// would break the capture); a pair launch that would have to go without the word -- captured before any eager call -- uses the
// one-wave-per-tile kernel instead, which has no waits.  Read by r6_device_status(), r6_last_error() and at the top of every
// entry point that steps envs; cleared only by r6_clear_device_error().
static std::atomic<int*> g_pair_abort_host{nullptr};
static std::atomic<int*> g_pair_abort_dev{nullptr};
static int* pair_abort_word(hipStream_t s) {
    int* d = g_pair_abort_dev.load(std::memory_order_acquire);
    if (d) return d;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return nullptr;
    }
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    d = g_pair_abort_dev.load(std::memory_order_acquire);
    if (d) return d;
    int* h = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&h), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !h) {
        (void)hipGetLastError();
        return nullptr;
    }
    *h = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d), h, 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        (void)hipHostFree(h);
        return nullptr;
    }
    g_pair_abort_host.store(h, std::memory_order_release);
    g_pair_abort_dev.store(d, std::memory_order_release);
    return d;
}
static bool pair_abort_raised() {
    int* h = g_pair_abort_host.load(std::memory_order_acquire);
    return h && *reinterpret_cast<volatile int*>(h) != 0;
}
static const char kDeviceErrorText[] =
    "a kernel of an earlier call of this process gave up a bounded wait between the waves of a workgroup (step_k_pair_kernel): "
    "its outputs are incomplete and the envs it stepped are in an unknown state; re-initialise them (r6_reset) and call "
    "r6_clear_device_error()";
// Top of every entry point that steps envs: makes the word exist (eager calls) and refuses to step on after an abort.
static int device_error_guard(const char* who, hipStream_t s) {
    (void)pair_abort_word(s);
    if (!pair_abort_raised()) return 0;
    std::snprintf(g_err, sizeof(g_err), "%s: %s", who, kDeviceErrorText);
    return R6_E_DEVICE;
}

template <int INTEG, int SHAPING, int OD>
void launch_step_k_inst(const R6Params& p, const R6Buffers& b, const R6StepK& sk, uint64_t seed, uint64_t step, uint64_t off, hipStream_t s) {
    if constexpr (INTEG != R6_RK45) {
        // at most one 64-env tile per SIMD: the two-waves-per-tile form (step_k_pair_kernel).  R6_K_PAIR=0|1 overrides.
        static const int pair_env = [] { const char* e = std::getenv("R6_K_PAIR"); return e ? std::atoi(e) : -1; }();
        const int64_t n_tiles = (b.n + 63) / 64;
        // Not the pair form when several steps would write ONE reward / terms buffer (stride 0 without R6_K_LAST_ONLY: "all steps
        // write the same buffer, as K r6_step calls would"): the stores of steps k and k + 1 come from two different reward
        // waves, unordered against each other, and the buffer could end up holding step K - 2.  The one-wave kernel is in order.
        const bool same_buffer = sk.k > 1 && !(sk.flags & R6_K_LAST_ONLY) && (sk.reward_stride == 0 || (b.terms && sk.terms_stride == 0));
        int* const abort_word = pair_abort_word(s);    // nullptr: captured before any eager call -- no pair form without the word
        const bool pair = p.n_substeps == 1 && !same_buffer && abort_word != nullptr &&
                          (pair_env >= 0 ? pair_env != 0 : n_tiles <= R6_K_DIRECT_MAX_TILES);
        if (pair) {
            int64_t blocks = (n_tiles + r6::kPairTiles - 1) / r6::kPairTiles;
            if (blocks > R6_MAX_BLOCKS) blocks = R6_MAX_BLOCKS;
            R6StepK skl = sk;
            skl.flags = sk.flags & ~R6_K_DIRECT_OBS_INTERNAL;
            hipLaunchKernelGGL((r6::step_k_pair_kernel<INTEG, SHAPING, OD>), dim3((unsigned)blocks), dim3(r6::kPairBlock), 0, s, p, b, skl, seed, step, off,
                               abort_word);
            return;
        }
    }
    auto kernel = r6::step_k_kernel<INTEG, SHAPING, OD, false>;
    if constexpr (INTEG != R6_RK45) {
        if (p.n_substeps == 1) kernel = r6::step_k_kernel<INTEG, SHAPING, OD, true>;
    }
    const int slot = 16 + INTEG * 4 + SHAPING * 2 + (OD == 14 ? 1 : 0);
    const int64_t n_tiles = (b.n + 63) / 64;
    int64_t blocks = (n_tiles + r6::kWaves - 1) / r6::kWaves;
    const int cap = resident_blocks(kernel, slot);
    if (blocks > cap) blocks = cap;   // persistent waves: a tile's K steps are serial, tiles are spread over the resident grid
    // few waves per SIMD: the observation leaves straight from registers (see step_k_kernel).  R6_K_DIRECT=0|1 overrides.
    static const int direct_env = [] { const char* e = std::getenv("R6_K_DIRECT"); return e ? std::atoi(e) : -1; }();
    R6StepK skl = sk;
    const bool direct = direct_env >= 0 ? direct_env != 0 : n_tiles <= R6_K_DIRECT_MAX_TILES;
    skl.flags = (sk.flags & ~R6_K_DIRECT_OBS_INTERNAL) | (direct ? R6_K_DIRECT_OBS_INTERNAL : 0);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(r6::kBlock), 0, s, p, b, skl, seed, step, off);
}

template <int INTEG, int SHAPING>
void launch_step_k_od(const R6Params& p, const R6Buffers& b, const R6StepK& sk, uint64_t seed, uint64_t step, uint64_t off, hipStream_t s) {
    if (p.obs_dim == 14)
        launch_step_k_inst<INTEG, SHAPING, 14>(p, b, sk, seed, step, off, s);
    else
        launch_step_k_inst<INTEG, SHAPING, 13>(p, b, sk, seed, step, off, s);
}

template <int INTEG, int SHAPING>
void launch_step_od(const R6Params& p, const R6Buffers& b, uint64_t seed, uint64_t step, uint64_t off, hipStream_t s) {
    if (p.obs_dim == 14)
        launch_step_inst<INTEG, SHAPING, 14>(p, b, seed, step, off, s);
    else
        launch_step_inst<INTEG, SHAPING, 13>(p, b, seed, step, off, s);
}

}  // namespace

extern "C" {

// shared with r6_policy.hip (not part of the public header, and hidden from the dynamic symbol table)
__attribute__((visibility("hidden"))) void r6_internal_set_error(const char* what) { std::snprintf(g_err, sizeof(g_err), "%s", what); }
// shared with r6_rollout.hip (not part of the public header, hidden): the guard of the stepping entry points
__attribute__((visibility("hidden"))) int r6_internal_device_error_guard(const char* who, void* stream) { return device_error_guard(who, static_cast<hipStream_t>(stream)); }
#ifdef R6_TEST_HOOKS
// test hook (tests/test_gpu_step_k.py; libr6dof_testhooks.so only -- the release library does not export it): raises the device
// error word as a kernel would; returns 0 when the word does not exist yet (no stepping call so far).
int r6_internal_test_pair_abort(int raise) {
    int* h = g_pair_abort_host.load(std::memory_order_acquire);
    if (!h) return 0;
    *reinterpret_cast<volatile int*>(h) = raise ? 1 : 0;
    return 1;
}
// test hook: where the hardware puts the waves of a workgroup of step_k_pair_kernel's shape.  out[(block * waves + w) * 2] = HW_ID,
// [.. + 1] = XCC_ID of wave w.  The pair kernel runs waves w, w + 4, w + 8 of a 768-thread workgroup as the dynamics and reward waves
// of ONE tile and counts on their sharing a SIMD (HW_ID.simd_id) for speed -- never for a result; tests/test_gpu_step_k.py watches it.
int r6_internal_test_wave_placement(uint32_t* out, int blocks, void* stream) {
    if (!out || blocks < 1) return R6_E_BADARG;
    hipLaunchKernelGGL(r6::wave_placement_kernel, dim3((unsigned)blocks), dim3(r6::kPairBlock), 0, static_cast<hipStream_t>(stream), out);
    return check_launch("r6_internal_test_wave_placement");
}
#endif
size_t r6_policy_net_size(void) { return sizeof(R6PolicyNet); }

int r6_abi_version(void) { return R6_ABI_VERSION; }
size_t r6_params_size(void) { return sizeof(R6Params); }
size_t r6_buffers_size(void) { return sizeof(R6Buffers); }
const char* r6_last_error(void) {
    // a raised device error word outranks whatever the last call of this thread said: it is never reported late or not at all
    if (pair_abort_raised() && !std::strstr(g_err, "bounded wait")) std::snprintf(g_err, sizeof(g_err), "device error: %s", kDeviceErrorText);
    return g_err;
}
int r6_device_status(void) { return pair_abort_raised() ? R6_E_DEVICE : 0; }
int r6_clear_device_error(void) {
    const int was = r6_device_status();
    if (int* h = g_pair_abort_host.load(std::memory_order_acquire)) *reinterpret_cast<volatile int*>(h) = 0;
    return was;
}

int r6_tune(const char* key, double value) {
    if (!key) return fail(R6_E_BADARG, "r6_tune: null key");
    StepTune& t = tune();
    if (!std::strcmp(key, "even_xcd_share")) t.even_share = value;
    else if (!std::strcmp(key, "grid_blocks")) t.grid_blocks = (int)value;
    else if (!std::strcmp(key, "sweep")) t.sweep = (int)value;
    else if (!std::strcmp(key, "sweep_flags")) t.sweep_flags = (int)value;
    else if (!std::strcmp(key, "balanced_tiles_per_wave")) t.balanced_tpw = (int)value;
    else if (!std::strcmp(key, "streaming_stores")) t.nt_mode = (int)value;
    else if (!std::strcmp(key, "rk45_defer")) t.rk45_defer = (int)value;
    else return fail(R6_E_BADARG, "r6_tune: unknown key");
    return 0;
}

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
    if (int rc = device_error_guard("r6_step", s)) return rc;
    const R6Params& p = *params;
    if (p.integrator == R6_RK45) {
        if (p.shaping == R6_SHAPING_ACCELERATION)
            launch_step_od<R6_RK45, R6_SHAPING_ACCELERATION>(p, b, seed, step_index, env_offset, s);
        else
            launch_step_od<R6_RK45, R6_SHAPING_VELOCITY>(p, b, seed, step_index, env_offset, s);
    } else if (p.integrator == R6_RK4) {
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

size_t r6_stepk_size(void) { return sizeof(R6StepK); }

int r6_step_k(const R6Params* params, const R6Buffers* buf, const R6StepK* steps, uint64_t seed, uint64_t step_index,
              uint64_t env_offset, void* stream) {
    if (int rc = validate_common(params, buf)) return rc;
    if (!steps) return fail(R6_E_BADARG, "r6_step_k: null R6StepK");
    if (int rc = device_error_guard("r6_step_k", static_cast<hipStream_t>(stream))) return rc;
    const R6Buffers& b = *buf;
    const R6StepK& sk = *steps;
    if (sk.k < 1 || sk.k > 65536) return fail(R6_E_BADARG, "r6_step_k: k must be in 1..65536");
    if (b.n == 0) return 0;
    if (!b.state || !b.epconst || !b.action || !b.obs || !b.reward || !b.done)
        return fail(R6_E_BADARG, "r6_step_k: state/epconst/action/obs/reward/done are required");
    if (!aligned16(b.action) || !aligned16(b.obs))
        return fail(R6_E_ALIGN, "r6_step_k: action and obs must be 16-byte aligned");
    if (sk.action_stride < 0 || sk.obs_stride < 0 || sk.reward_stride < 0 || sk.done_stride < 0 || sk.flags_stride < 0 ||
        sk.nfev_stride < 0 || sk.terms_stride < 0 || sk.final_obs_stride < 0 || sk.fin_stride < 0)
        return fail(R6_E_BADARG, "r6_step_k: strides must be >= 0");
    if (sk.k > 1 && sk.action_stride < b.n * R6_ACTION_DIM) return fail(R6_E_BADARG, "r6_step_k: action_stride must be >= 3 n");
    if ((sk.obs_stride % 4) != 0) return fail(R6_E_ALIGN, "r6_step_k: obs_stride must be a multiple of 4 elements (16-byte stores)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const R6Params& p = *params;
    const bool acc = p.shaping == R6_SHAPING_ACCELERATION;
    if (p.integrator == R6_RK45) {
        if (acc) launch_step_k_od<R6_RK45, R6_SHAPING_ACCELERATION>(p, b, sk, seed, step_index, env_offset, s);
        else launch_step_k_od<R6_RK45, R6_SHAPING_VELOCITY>(p, b, sk, seed, step_index, env_offset, s);
    } else if (p.integrator == R6_RK4) {
        if (acc) launch_step_k_od<R6_RK4, R6_SHAPING_ACCELERATION>(p, b, sk, seed, step_index, env_offset, s);
        else launch_step_k_od<R6_RK4, R6_SHAPING_VELOCITY>(p, b, sk, seed, step_index, env_offset, s);
    } else {
        if (acc) launch_step_k_od<R6_EULER, R6_SHAPING_ACCELERATION>(p, b, sk, seed, step_index, env_offset, s);
        else launch_step_k_od<R6_EULER, R6_SHAPING_VELOCITY>(p, b, sk, seed, step_index, env_offset, s);
    }
    return check_launch("r6_step_k");
}

int r6_reset(const R6Params* params, const R6Buffers* buf, const uint8_t* mask, uint64_t seed, uint64_t step_index,
             uint64_t env_offset, void* stream) {
    if (int rc = validate_common(params, buf)) return rc;
    const R6Buffers& b = *buf;
    if (b.n == 0) return 0;
    if (!b.state || !b.epconst) return fail(R6_E_BADARG, "r6_reset: state/epconst are required");
    hipStream_t s = static_cast<hipStream_t>(stream);
    (void)pair_abort_word(s);   // the first call of a process is normally this one, eager: the device error word exists from here on
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
    if (params->integrator == R6_RK45)
        hipLaunchKernelGGL((r6::sim_step_kernel<R6_RK45>), grid_for(n), dim3(r6::kBlock), 0, s, *params, state, epconst, u_phys, status, n, ld);
    else if (params->integrator == R6_RK4)
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
