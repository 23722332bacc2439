/*
 * r6dof.h -- C ABI of the MI355X-native batched Rocket6DOF hot path (libr6dof.so).
 *
 * The reference (Tuxliri/RL_Rocket_6DOF) is pure Python and has no FFI seam; the seam this ABI
 * replaces is the Python class surface itself:
 *     my_environment/envs/rocket_env.py:201-231   Rocket6DOF.step        -> r6_step
 *     my_environment/envs/rocket_env.py:180-199   Rocket6DOF.reset       -> r6_reset
 *     my_environment/envs/rocket_env.py:523-524   Rocket6DOF._get_obs    -> r6_observe
 *     my_environment/utils/simulator.py:69-104    Simulator6DOF.step     -> r6_sim_step
 *     my_environment/utils/simulator.py:106-143   Simulator6DOF.RHS      -> r6_rhs
 *     main_6DOF.py:107-139 / montecarlo_script.py  SB3 policy.forward / predict (the caller of step) -> r6_policy_act
 *     main_6DOF.py:136-139  SB3 collect_rollouts: forward -> clip -> env.step, per timestep   -> r6_rollout (K steps, one launch)
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Every pointer in R6Buffers / the call arguments is a DEVICE pointer owned by the caller
 *     (e.g. a torch ROCm tensor's data_ptr); the library never allocates, frees or copies.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls are
 *     asynchronous, stream-ordered, stateless and re-entrant across streams and devices.
 *   - Return value: 0 = OK, negative = -(hipError_t) or R6_E_*; r6_last_error() gives the text
 *     (thread-local).  No C++ exception crosses the ABI.
 *   - SoA arrays are [rows][ld] float with ld >= n (ld*4 bytes should be a multiple of 16 for
 *     full-width accesses; torch allocations are; avoid ld a multiple of 2048: rows 8 KiB-multiples apart
 *     collide on HBM channels, +3 % per step).  AoS arrays are row-major (n, width).
 *   - State rows (simulator.py:14-28): 0-2 position x,y,z (x = height), 3-5 velocity,
 *     6-9 quaternion (leading scalar), 10-12 body rates, 13 mass.
 */
#ifndef R6DOF_H
#define R6DOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define R6_ABI_VERSION 7
#define R6_STATE_DIM 14
#define R6_ACTION_DIM 3
#define R6_NUM_TERMS 8 /* atarg|vtarg_tracking, thrust_penalty, eta, attitude_constraint,
                          goal_conditions, final_position, final_velocity (rocket_env.py:353-359),
                          vertical_attitude_reward (wrappers.py:128-155, 0 unless enabled) */

enum { R6_EULER = 0, R6_RK4 = 1, R6_RK45 = 2 /* adaptive Dormand-Prince, scipy solve_ivp step control */ };
enum { R6_SHAPING_ACCELERATION = 0, R6_SHAPING_VELOCITY = 1 };
enum { R6_REWARD_STANDARD = 0, R6_REWARD_ANNEALED = 1 };
/* bits of R6Buffers.flags[i] */
enum { R6_FLAG_GROUND = 1, R6_FLAG_BOUNDS = 2, R6_FLAG_TRUNCATED = 4 };
enum { R6_E_BADARG = -10001, R6_E_ALIGN = -10002, R6_E_DEVICE = -10003 /* a kernel of an earlier call reported a failed launch-internal wait: see r6_device_status */ };
/* columns of R6Buffers.stats */
#define R6_NUM_STATS 8
enum {
    R6_STAT_EPISODES = 0, /* finished episodes */
    R6_STAT_RETURN = 1,   /* sum of episode returns */
    R6_STAT_RETURN_SQ = 2,/* sum of squared episode returns */
    R6_STAT_LENGTH = 3,   /* sum of episode lengths */
    R6_STAT_GROUND = 4,   /* episodes ended by the ground event */
    R6_STAT_BOUNDS = 5,   /* episodes ended by a bounds violation */
    R6_STAT_TRUNCATED = 6,/* episodes ended by the time limit */
    R6_STAT_LANDED = 7    /* episodes whose last step met all landing conditions (goal term) */
};
#define R6_BLOCK 256      /* threads per workgroup (4 wave64) */
#define R6_MAX_BLOCKS 16384 /* upper bound of the step grid = rows of `stats` */

/* Kernel constants: everything config.yaml's env_config can change (rocket_env.py:22-178),
 * pre-digested on the host.  All members are 4 bytes wide (no padding anywhere). */
typedef struct R6Params {
    /* integrator */
    float dt;             /* env timestep [s] (config.yaml timestep) */
    int32_t integrator;   /* R6_EULER | R6_RK4 | R6_RK45 */
    int32_t n_substeps;   /* >= 1 fixed sub-steps per env step (ignored by R6_RK45) */
    int32_t shaping;      /* R6_SHAPING_* (reward_shaping_type) */
    /* action scaling, rocket_env.py:86-87, :509-521 */
    float max_gimbal;     /* rad */
    float max_thrust;     /* N */
    /* observation normaliser reciprocal, rocket_env.py:106-126, :503-504 */
    float inv_norm[R6_STATE_DIM];
    /* position bounds (inclusive float32 box), rocket_env.py:129-134, :591-593 */
    float bound_lo[3];
    float bound_hi[3];
    /* reward coefficients, config.yaml reward_coeff */
    float alfa, beta, eta, gamma, kappa, w_r_f, w_v_f, max_r_f, max_v_f;
    float bounds_penalty; /* -50, rocket_env.py:228-229 */
    /* Euler-zyx attitude limits, pre-transformed so that no inverse trig is needed:
     *   e0 = atan2(-M01, M00), e2 = atan2(-M12, M22):  |e| > L  <=>  x < cosL * hypot(x, y)
     *   e1 = asin(M02):                                |e| > L  <=>  |M02| > sinL
     * cos entries are -2 when L >= pi (never exceeded / always inside), sin entries are 2 when
     * L >= pi/2.  traj_* : trajectory_limits.attitude_limit (rocket_env.py:159, :366-369);
     * land_* : landing_params.landing_attitude_limit (rocket_env.py:165, :386-388). */
    float traj_c0, traj_s1, traj_c2;
    float land_c0, land_s1, land_c2;
    float land_radius;    /* landing_params.landing_radius, rocket_env.py:162 */
    float land_vmax;      /* landing_params.maximum_velocity, rocket_env.py:163 */
    float omega_lim[3];   /* hard-coded 0.2 in the reference, rocket_env.py:167 */
    float waypoint;       /* landing_params.waypoint (velocity shaping), rocket_env.py:168 */
    /* reset distribution U(ic_lo, ic_lo + ic_range), rocket_env.py:78-81 */
    float ic_lo[R6_STATE_DIM];
    float ic_range[R6_STATE_DIM];
    /* fused wrapper epilogue (main_6DOF.py:33-53): RemoveMassFromObs, ClipReward, TimeLimit */
    int32_t obs_dim;            /* 14 raw env, 13 = RemoveMassFromObs (wrappers.py:119-126) */
    int32_t clip_reward;        /* 0/1 ClipReward (main_6DOF.py:33-42) */
    float clip_lo, clip_hi;
    int32_t max_episode_steps;  /* 0 = no TimeLimit (main_6DOF.py:31,47-50) */
    int32_t auto_reset;         /* 0/1 re-sample done envs in-kernel (VecEnv semantics) */
    /* optional reward wrappers of the reference (my_environment/wrappers/wrappers.py) */
    int32_t reward_mode;        /* R6_REWARD_ANNEALED = RewardAnnealing (:39-61): reward = attitude_constraint +
                                   goal_conditions + final_position + final_velocity - xi * (action[2] + 1) */
    float xi;                   /* reward_coeff["xi"] (default 0.01) */
    int32_t vatt_enable;        /* 0/1 VerticalAttitudeReward (:128-155) */
    float vatt_threshold;       /* threshold_height (1e-3) */
    float vatt_weight;          /* weight (-0.5) */
    int32_t reserved0;          /* keeps sizeof(R6Params) a multiple of 8 */
} R6Params;

/* Device buffers of one shard of n environments. */
typedef struct R6Buffers {
    int64_t n;            /* environments in this shard */
    int64_t ld;           /* leading dimension of every SoA array (>= n) */
    float *state;         /* [14][ld]  in/out */
    float *epconst;       /* [2][ld]   in/out: row 0 = m0 (initial mass -> frozen inertia,
                                       simulator.py:42-50), row 1 = v0 = |IC velocity|
                                       (rocket_env.py:649-651, velocity shaping only) */
    const float *action;  /* (n,3)     in: normalised actions in [-1,1] (not clipped) */
    float *obs;           /* (n,obs_dim) out */
    float *reward;        /* [n]       out */
    uint8_t *done;        /* [n]       out: exactly 0 or 1 */
    /* ---- optional outputs / state: NULL to skip ---- */
    uint8_t *flags;       /* [n]       out: R6_FLAG_* bits */
    float *terms;         /* [8][ld]   out: per-term rewards (info["rewards_dict"]) */
    float *final_obs;     /* (n,obs_dim) out: rows of done envs only ("terminal_observation") */
    uint32_t *ep_step;    /* [n]       in/out: steps elapsed in the episode (TimeLimit/Monitor) */
    float *ep_return;     /* [n]       in/out: running episode return (Monitor) */
    float *fin_return;    /* [n]       out at done: episode return */
    uint32_t *fin_length; /* [n]       out at done: episode length */
    uint8_t *nfev;        /* [n]       out: RHS evaluations of the step (scipy sol.nfev), saturating at 255 */
    double *stats;        /* [R6_MAX_BLOCKS][R6_NUM_STATS] in/out: per-workgroup running sums over finished episodes (see
                                       R6_STAT_*).  Inside a workgroup they are exact integer / fixed-point sums (LDS integer
                                       atomics; the episode return and its square in units of 2^-16, returns clamped to
                                       +-32767), added to the workgroup's own float64 row (no global atomics); the caller
                                       sums the rows.  The totals are exact and therefore independent of arrival order,
                                       launch shape and sharding */
    const uint64_t *step_base; /* [1]  in (device memory) or NULL: added to the `step_index` argument where the reset
                                       RNG is keyed.  A launch recorded into a HIP graph carries its by-value arguments
                                       for ever; with the counter's base in device memory a replayed rollout still draws
                                       fresh initial conditions (the caller advances *step_base between replays) */
} R6Buffers;

int r6_abi_version(void);
size_t r6_params_size(void);   /* sizeof(R6Params), for binding self-checks */
size_t r6_buffers_size(void);  /* sizeof(R6Buffers) */
const char *r6_last_error(void);  /* text of the last failure on this thread; a raised device error word (below) outranks it */

/* Device-side error word (process-wide, sticky).  A kernel whose launch-internal bounded wait ran out (r6_step_k's waves that hand
 * each other work through LDS) sets a word of mapped host memory before it exits; from then on r6_step, r6_step_k and r6_rollout
 * return R6_E_DEVICE without launching (the envs that kernel stepped are in an unknown state) and r6_last_error() says why --
 * whichever call comes next, not only r6_step_k.  r6_device_status(): 0, or R6_E_DEVICE while the word is raised (also after the
 * LAST launch of a run, once the stream has been synchronised).  r6_clear_device_error(): the caller has re-initialised its envs
 * (r6_reset); clears the word and returns the status it had. */
int r6_device_status(void);
int r6_clear_device_error(void);

/* Launch-shape knobs of r6_step (process-wide; results never depend on them, only which wave computes which tile).  Keys:
 * "even_xcd_share" (share of the tiles given to the workgroups on even XCDs, < 0 = default), "grid_blocks" (> 0 pins the
 * persistent grid), "sweep" (tile order on odd steps, -1 = default), "sweep_flags" (4: no wave-priority ladder),
 * "balanced_tiles_per_wave" (large shards: tiles per wave of the dispatcher-balanced grid, 0 = by shard size), "streaming_stores"
 * (non-temporal output stores: -1 = by shard size, 0 = never, 1 = always), "rk45_defer" (integrator R6_RK45: a wave hands its last
 * <= this many unfinished lanes to the workgroup's straggler pass instead of following them; 0 = never; R6_RK45_DEFER).  The first
 * three are also read once from the environment: R6_EVEN_XCD_SHARE, R6_GRID_BLOCKS, R6_SWEEP. */
int r6_tune(const char *key, double value);

/* Rocket6DOF.step for n envs (rocket_env.py:201-231): denormalise action, integrate one env
 * step (fixed Euler/RK4 sub-steps, ground-event localisation, quaternion renormalise), bounds,
 * reward, done, observation; optional fused wrappers and in-kernel auto-reset.
 * `seed`, `step_index` (the caller's global step counter, >= 1) and `env_offset` (global index of
 * this shard's env 0) key the counter-based reset RNG, so results do not depend on sharding. */
int r6_step(const R6Params *params, const R6Buffers *buf, uint64_t seed, uint64_t step_index,
            uint64_t env_offset, void *stream);

/* K consecutive Rocket6DOF.step calls in one launch (the per-timestep loop of main_6DOF.py:136-139 /
 * montecarlo_script.py:33-51 without its per-step launch): actions for all K steps are given up front (open-loop
 * rollouts: random or pre-computed action sequences), the state stays in registers between the steps, finished
 * episodes are re-sampled in-kernel exactly as r6_step does it.  Results are bit for bit those of K r6_step calls with
 * step_index, step_index + 1, ...
 * Arrays that r6_step produces per step carry a leading K dimension here, addressed with the element strides below
 * (stride 0: all steps write the same buffer, as K r6_step calls on one R6Buffers would):
 *   action (k,n,3) in;  obs (k,n,obs_dim), reward (k,n), done (k,n), flags (k,n), nfev (k,n), terms (k,8,ld),
 *   final_obs (k,n,obs_dim), fin_return / fin_length (k,n) out.
 * state, epconst, ep_step, ep_return, stats are the same single arrays as in r6_step (read before / written after).
 * Batches of at most one 64-env tile per SIMD run as two kinds of waves per tile (dynamics / reward) that hand each other
 * work through LDS; their waits are bounded, and a wait that ever ran out raises the device error word (r6_device_status above):
 * every later stepping call of the process fails with R6_E_DEVICE until r6_clear_device_error() -- never silently.  (That form is
 * not used when k > 1 steps write ONE reward / terms buffer -- stride 0 without R6_K_LAST_ONLY -- where only the in-order kernel
 * leaves the last step's values.) */
enum { R6_K_LAST_ONLY = 1 /* only the last step's per-step outputs are stored (throughput rollouts, burn-in) */ };
typedef struct R6StepK {
    int32_t k;                /* env steps in this launch, 1..65536 */
    int32_t flags;            /* R6_K_* */
    int64_t action_stride;    /* elements from step j to step j+1 of `action` (n*3 when contiguous) */
    int64_t obs_stride;       /* ... of `obs` (n*obs_dim) */
    int64_t reward_stride;    /* ... of `reward` (n) */
    int64_t done_stride;      /* ... of `done` (n) */
    int64_t flags_stride;     /* ... of `flags` (n) */
    int64_t nfev_stride;      /* ... of `nfev` (n) */
    int64_t terms_stride;     /* ... of `terms` (8*ld) */
    int64_t final_obs_stride; /* ... of `final_obs` (n*obs_dim) */
    int64_t fin_stride;       /* ... of `fin_return` and `fin_length` (n) */
} R6StepK;
size_t r6_stepk_size(void);    /* sizeof(R6StepK) */
int r6_step_k(const R6Params *params, const R6Buffers *buf, const R6StepK *steps, uint64_t seed, uint64_t step_index,
              uint64_t env_offset, void *stream);

/* Rocket6DOF.reset for the envs with mask[i] != 0 (mask == NULL: all), rocket_env.py:180-199:
 * samples the initial condition, normalises the quaternion, sets m0/v0, writes obs, zeroes the
 * episode counters. */
int r6_reset(const R6Params *params, const R6Buffers *buf, const uint8_t *mask, uint64_t seed,
             uint64_t step_index, uint64_t env_offset, void *stream);

/* obs <- state / normaliser (rocket_env.py:503-504, :523-524) for all n envs. */
int r6_observe(const R6Params *params, const R6Buffers *buf, void *stream);

/* Simulator6DOF.step (simulator.py:69-104) only: state in/out, physical controls
 * u_phys (n,3) = [delta_y rad, delta_z rad, thrust N]; status[i] = 1 on ground event. */
int r6_sim_step(const R6Params *params, float *state, const float *epconst, const float *u_phys,
                uint8_t *status, int64_t n, int64_t ld, void *stream);

/* Simulator6DOF.RHS (simulator.py:106-143): dy[14][ld] <- f(y[14][ld], u_phys (n,3)). */
int r6_rhs(const R6Params *params, const float *y, const float *epconst, const float *u_phys,
           float *dy, int64_t n, int64_t ld, void *stream);

/* ---- rollout-side policy forward (the caller of r6_step in a PPO rollout) -------------------------------
 * SB3 1.6 `ActorCriticPolicy` as the reference builds it (`PPO("MlpPolicy", env, policy_kwargs=
 * {'net_arch': [128, 64]})`, main_6DOF.py:107-139; SB3 default net_arch (64, 64)): separate tanh MLPs for
 * policy and value, state-independent log_std, diagonal Gaussian.  Weights are device pointers to fp32
 * arrays in torch.nn.Linear layout ([out][in] row-major), not copied: the launch reads them. */
typedef struct R6PolicyNet {
    int32_t obs_dim;            /* 13 | 14 (= R6Params.obs_dim) */
    int32_t hidden1, hidden2;   /* (64, 64) or (128, 64) */
    int32_t shared;             /* 0: separate policy / value MLPs (SB3 default, net_arch=[dict(pi=.., vf=..)]);
                                   1: ONE trunk feeding both heads -- what SB3 1.6 builds for a plain list such as
                                   the reference's net_arch=[128, 64] (main_6DOF.py:113; its checkpoints hold
                                   mlp_extractor.shared_net.*): pi_w1/b1/w2/b2 are the trunk, vf_w1..vf_b2 unused */
    const float *pi_w1, *pi_b1; /* [hidden1][obs_dim], [hidden1] */
    const float *pi_w2, *pi_b2; /* [hidden2][hidden1], [hidden2] */
    const float *pi_w3, *pi_b3; /* [3][hidden2], [3]   (action mean) */
    const float *vf_w1, *vf_b1;
    const float *vf_w2, *vf_b2;
    const float *vf_w3, *vf_b3; /* [1][hidden2], [1]   (state value) */
    const float *log_std;       /* [3] */
} R6PolicyNet;

size_t r6_policy_net_size(void); /* sizeof(R6PolicyNet) */

/* SB3 `ActorCriticPolicy.forward(obs, deterministic)` followed by the clip to the action Box that
 * `OnPolicyAlgorithm.collect_rollouts` applies before `env.step`, for n envs in one launch:
 *   action (n,3)      clipped to [-1,1]: feed it to r6_step          (required)
 *   raw_action (n,3)  the unclipped sample: what the rollout buffer stores   (NULL to skip)
 *   value [n], logp [n]  state value, log-probability of raw_action          (NULL to skip)
 * Gaussian noise: Philox4x32-10 keyed by (seed, step_index, env_offset + i) + Box-Muller, so a rollout
 * does not depend on sharding; `deterministic` != 0 returns the mean action (`model.predict(...,
 * deterministic=True)`, montecarlo_script.py).  fp32-accurate arithmetic on the matrix cores: every fp32 operand is
 * split into an f16 hi + lo pair (v_mfma_f32_32x32x16_f16, three products per K-step, fp32 accumulation; agrees
 * with torch fp32 to 2e-5; |obs| and |weights| must stay below f16's 65504).
 *   step_base (device, [1], or NULL) is added to step_index inside the kernel (see R6Buffers.step_base: graph replays).
 *   row_flags [n] / flag_mask: when row_flags != NULL only 32-row tiles holding a row with (row_flags[i] & flag_mask) != 0
 *   are evaluated and written, the others are skipped without touching their outputs -- V(terminal observation) for the
 *   envs a TimeLimit truncated in this step (row_flags = R6Buffers.flags, flag_mask = R6_FLAG_TRUNCATED), decided on the
 *   device: a rollout loop needs no host round trip to know whether any env was truncated. */
int r6_policy_act(const R6PolicyNet *net, const float *obs, int64_t n, float *action, float *raw_action,
                  float *value, float *logp, uint64_t seed, uint64_t step_index, const uint64_t *step_base,
                  uint64_t env_offset, int deterministic, const uint8_t *row_flags, int flag_mask, void *stream);

/* ---- K closed-loop rollout steps in one launch ------------------------------------------------------------------------
 * The per-timestep loop of SB3's `OnPolicyAlgorithm.collect_rollouts` behind main_6DOF.py:136-139 -- policy.forward(obs), clip to
 * the Box, env.step(action), and V(terminal_observation) for the envs a TimeLimit truncated -- K times, with the state of an env in
 * registers throughout: the observation never leaves the CU between the env step and the forward that consumes it.
 * Results are bit for bit those of the launch sequence  r6_policy_act(step j) -> r6_step(step j) [-> masked r6_policy_act]  for
 * j = 0..k-1 with the same seeds and step counters.  All arrays are contiguous with a leading K dimension.
 * R6Buffers supplies state / epconst / ep_step / ep_return / stats / step_base (in/out as in r6_step) and, optionally, final_obs /
 * fin_return / fin_length (rows of finished envs, overwritten step after step as K r6_step calls would) and obs (the observation
 * after the LAST step, = obs_seq row k); its action / reward / done / flags members are not used.
 * final_obs / fin_return / fin_length are written under R6Params.auto_reset only, as in r6_step.
 * Supported: R6PolicyNet hidden (64,64) | (128,64), obs_dim 13 | 14 equal to R6Params.obs_dim; n_substeps 1; rk4 with either
 * reward shaping, euler with acceleration shaping (euler + velocity shaping has no instance: R6_E_BADARG). */
typedef struct R6Rollout {
    int32_t k;              /* env steps in this launch, 1..65536 */
    int32_t deterministic;  /* 0: sampled actions (training rollouts); 1: the mean action (evaluation, montecarlo_script.py) */
    float *obs_seq;         /* (k+1, n, obs_dim) or NULL: row j+1 <- the observation after step j.  Row 0 is the caller's (the
                               observation the first action is computed from: the kernel derives it from `state`, never reads it) */
    float *raw_action;      /* (k, n, 3) or NULL: unclipped samples (what SB3's rollout buffer stores) */
    float *value;           /* (k, n) or NULL */
    float *logp;            /* (k, n) or NULL */
    float *reward;          /* (k, n) */
    uint8_t *done;          /* (k, n): exactly 0 or 1 */
    uint8_t *flags;         /* (k, n) or NULL: R6_FLAG_* bits */
    float *boot_value;      /* (k, n) or NULL: V(terminal observation) of the envs truncated in step j; other entries untouched */
} R6Rollout;
size_t r6_rollout_size(void);   /* sizeof(R6Rollout) */
/* env_seed / env_step_index (+ *R6Buffers.step_base): the reset stream, as r6_step's seed / step_index of the first step;
 * policy_seed / policy_step_index (+ *policy_step_base): the action-noise stream, as r6_policy_act's of the first step. */
int r6_rollout(const R6Params *params, const R6Buffers *buf, const R6PolicyNet *net, const R6Rollout *rollout, uint64_t env_seed,
               uint64_t env_step_index, uint64_t policy_seed, uint64_t policy_step_index, const uint64_t *policy_step_base,
               uint64_t env_offset, void *stream);

/* SB3 `RolloutBuffer.compute_returns_and_advantage` (GAE(lambda); stable_baselines3/common/buffers.py, called from
 * OnPolicyAlgorithm.collect_rollouts behind main_6DOF.py:136-139) for T steps of n envs in one launch, one lane per env:
 *   delta_t = r_t + gamma V_{t+1} (1 - done_t) - V_t,  A_t = delta_t + gamma lambda (1 - done_t) A_{t+1},  ret_t = A_t + V_t
 * rewards / values / adv / ret: (T, n) float32; dones: (T, n) uint8, done_t = the episode ended AT step t; last_values [n].
 * Time-limit bootstrap (SB3 adds gamma * V(terminal_observation) to the reward of a truncated step): when flags and
 * boot_values are given, r_t += gamma * boot_values[t][i] where (flags[t][i] & flag_mask) != 0 -- boot_values entries of
 * other rows are never read (they may be uninitialised: see r6_policy_act's row_flags). */
int r6_gae(const float *rewards, const float *values, const uint8_t *dones, const float *last_values, int64_t T, int64_t n,
           float gamma, float lam, const uint8_t *flags, int flag_mask, const float *boot_values, float *adv, float *ret,
           void *stream);

/* SB3 1.6 `PPO.train` minibatch loss (stable_baselines3/ppo/ppo.py, the trainer behind main_6DOF.py:107-139) and
 * its gradient with respect to the policy outputs, for n samples in one launch:
 *   loss = -mean(min(A r, A clip(r, 1-c, 1+c))) + ent_coef * (-mean entropy) + vf_coef * mse(ret, value),
 *   r = exp(logp(action | mean, log_std) - old_logp),  A = (adv - adv_norm[0]) * adv_norm[1]  (adv_norm NULL: A = adv).
 * Outputs: grad_mean (n,3), grad_value [n]; grad_log_std [3] and stats [8] are ACCUMULATED (zero them first):
 * stats = {pg_loss, v_loss, entropy, clip_frac, approx_kl, n, -, -}.  All pointers are device pointers. */
int r6_ppo_loss_grad(const float *mean, const float *value, const float *log_std, const float *action,
                     const float *old_logp, const float *adv, const float *ret, const float *adv_norm, int64_t n,
                     float clip_range, float vf_coef, float ent_coef, float *grad_mean, float *grad_value,
                     float *grad_log_std, float *stats, void *stream);

/* ---- the whole PPO minibatch step's gradient in one launch ---------------------------------------------------------
 * SB3 1.6 `PPO.train` per minibatch (stable_baselines3/ppo/ppo.py, behind main_6DOF.py:107-139): evaluate_actions on the
 * minibatch rows, the clipped-surrogate / value / entropy loss of r6_ppo_loss_grad, and loss.backward() through the
 * actor-critic -- forward, loss and backward fused, on the matrix cores with fp32 accuracy: the layer products run on f16
 * hi + lo splits of both operands with fp32 accumulation (lo rounded to nearest; against float64 autograd the gradients are as
 * close as torch's own fp32 autograd), dW1 on v_mfma_f32_16x16x4_f32.  The splits put the observations
 * and the weights through f16: |obs|, |weights| must stay below 65504 (they are O(1) here: normalised observations).  Supported
 * nets (obs_dim 13 | 14): the reference's topology, ONE shared tanh trunk obs -> 128 -> 64 with a 3-output action head and a
 * value head (R6PolicyNet.shared = 1, hidden (128, 64)), and SB3's default, two separate tanh MLPs obs -> 64 -> 64
 * (shared = 0, hidden (64, 64); BASELINE configs[4] "SB3 PPO MLP(64,64)").
 *   obs (rows, obs_dim), action (rows, 3), old_logp / adv / ret [rows]: the rollout buffer; `index` [n] (int64, or NULL for
 *   rows 0..n-1) picks the minibatch, so no gathered copies are needed.  n_rows = rows of the rollout tensors: with an index it
 *   must be below 2^31 (row numbers are kept as 32-bit values one tile ahead of their use; R6_E_BADARG otherwise).  adv_norm = [mean, 1/(std + 1e-8)] or NULL.
 *   grad: WRITTEN (not accumulated), torch.nn.Linear layouts like the weights.  stats [16]: [0..7] written: pg_loss, v_loss,
 *   entropy, clip_frac, approx_kl, n; [8..15] += the same six (running sums over launches: the caller zeroes them when a new sum
 *   starts and reads means once per update).  scratch: r6_ppo_scratch_floats() floats of device memory owned by the caller. */
typedef struct R6PolicyGrad {
    float *w1, *b1;             /* trunk (or policy MLP) layer 1: [hidden1][obs_dim], [hidden1] */
    float *w2, *b2;             /* trunk (or policy MLP) layer 2: [hidden2][hidden1], [hidden2] */
    float *pi_w3, *pi_b3;       /* action head: [3][64], [3] */
    float *vf_w3, *vf_b3;       /* value head: [1][64], [1] */
    float *log_std;             /* [3] */
    float *vf_w1, *vf_b1;       /* value MLP of a net with separate MLPs (R6PolicyNet.shared = 0); NULL for a shared trunk */
    float *vf_w2, *vf_b2;
} R6PolicyGrad;
size_t r6_policy_grad_size(void);   /* sizeof(R6PolicyGrad) */
/* out2 = [mean, 1 / (std + 1e-8)] (unbiased std) of adv[index[0..n)] (index NULL: adv[0..n)): SB3's per-minibatch
 * advantage normalisation, the `adv_norm` input of the two loss entry points, without a gathered copy.  n_batches > 1: the
 * consecutive minibatches index[b n .. (b + 1) n) of an epoch's permutation in one call, out2[b][2].  scratch: 256 doubles
 * per minibatch, device memory owned by the caller. */
int r6_minibatch_adv_norm(const float *adv, const int64_t *index, int64_t n, int64_t n_batches, float *out2, double *scratch,
                          void *stream);
size_t r6_ppo_scratch_floats(void);
int r6_ppo_minibatch_grad(const R6PolicyNet *net, const float *obs, const float *action, const float *old_logp,
                          const float *adv, const float *ret, const int64_t *index, int64_t n, int64_t n_rows,
                          const float *adv_norm, float clip_range, float vf_coef, float ent_coef, const R6PolicyGrad *grad,
                          float *stats, float *scratch, size_t scratch_floats, void *stream);

/* torch.nn.utils.clip_grad_norm_(params, max_grad_norm) + torch.optim.Adam.step() (SB3 1.6's optimiser step, with its
 * eps 1e-5) for a policy whose parameters / gradients / Adam moments are FLAT device arrays of n floats, in one launch.
 * lr_dev (device scalar, optional) overrides lr; *step_dev (device int64, starts at 0) is the Adam step count and is
 * incremented; elements [clamp_off, clamp_off + clamp_n) are floored at clamp_min afterwards (log-std floor; clamp_n = 0: none).
 * max_grad_norm <= 0: no clipping. */
int r6_adam_clip_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float max_grad_norm,
                      float lr, const float *lr_dev, float beta1, float beta2, float eps, int64_t *step_dev,
                      int64_t clamp_off, int64_t clamp_n, float clamp_min, void *stream);

/* out[0..n) = a pseudo-random permutation of 0..n-1 keyed by (seed, counter + *counter_base): the minibatch order of an SB3
 * epoch (`RolloutBuffer.get`: np.random.permutation, stable_baselines3/common/buffers.py) as ONE elementwise launch -- a
 * keyed bijection with cycle walking, no sort.  counter_base (device, [1], or NULL) as in R6Buffers.step_base: a recorded
 * launch draws a new permutation at every graph replay once the caller advances it. */
int r6_permutation(int64_t *out, int64_t n, uint64_t seed, uint64_t counter, const uint64_t *counter_base, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* R6DOF_H */
