"""On-device PPO for ``BatchedRocket6DOF`` (BASELINE config 5).

The reference trains with stable-baselines3 1.6.0 ``PPO("MlpPolicy", env, policy_kwargs={'net_arch':
[128, 64]})`` on ONE env (reference main_6DOF.py:107-139; defaults recorded in its checkpoints:
n_steps 2048, batch 64, epochs 10, lr 3e-4, gamma 0.99, lambda 0.95, clip 0.2, ent 0, vf 0.5,
max-grad-norm 0.5).  SB3 is not a dependency here and its Python loop over a list of N info dicts cannot
serve N = 2^20 environments, so this is an own implementation of the same algorithm and defaults whose
rollout buffer, advantage computation and minibatching stay in HBM next to the environment state:

  * policy = SB3 ``MlpPolicy``: separate tanh MLPs for actor and critic, orthogonal init (gains sqrt 2,
    0.01 for the action head, 1 for the value head), state-independent ``log_std`` initialised to 0;
  * actions are clipped to the Box before ``env.step`` (SB3 ``OnPolicyAlgorithm.collect_rollouts``),
    the un-clipped sample is what the log-probability is taken of;
  * time-limit truncations bootstrap with gamma * V(terminal_observation) (SB3 >= 1.5);
  * GAE(lambda), advantage normalisation per minibatch, clipped surrogate, MSE value loss, global
    gradient-norm clipping -- as in SB3's ``PPO.train``.

Multi-GPU: envs are sharded per rank (``dist.make_sharded_env``); gradients are averaged with one
bucketed all-reduce per optimiser step (RCCL), episode summaries are all-gathered once per rollout.
PPO learning curves cannot be compared with the reference bit for bit ("parity unpinned" -- different
sampling, one env vs millions); the tests check the algebra (GAE, losses) against closed forms and that
the return improves.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist
import torch.nn as nn

from .params import FLAG_TRUNCATED


@dataclass
class PPOConfig:
    n_steps: int = 32                # rollout length per env (SB3 default 2048 for ONE env)
    batch_size: int = 1 << 16        # minibatch size (SB3 default 64)
    n_epochs: int = 10
    learning_rate: float = 3e-4
    final_learning_rate: Optional[float] = None   # linear schedule over learn()'s total_timesteps (SB3 accepts a
                                                  # callable of the remaining progress; None = constant, its default)
    gamma: float = 0.99
    gae_lambda: float = 0.95
    clip_range: float = 0.2
    ent_coef: float = 0.0
    vf_coef: float = 0.5
    max_grad_norm: float = 0.5
    normalize_advantage: bool = True
    net_arch: Sequence[int] = (64, 64)   # SB3 MlpPolicy default; the reference's training run used [128, 64]
    shared_trunk: bool = False           # True = SB3 1.6 semantics of a plain-list net_arch (one trunk, two heads)
    log_std_init: float = 0.0
    log_std_min: Optional[float] = None   # floor of the policy's log-std, applied after every optimiser step (SB3 has
                                          # none: with millions of samples per update the std collapses and the late
                                          # phase of training turns erratic)
    seed: int = 42
    fused_update: bool = True        # forward + loss + the whole backward pass of a minibatch as ONE HIP launch
                                     # (r6_ppo_minibatch_grad) when the net is the reference's shared trunk [128, 64]
    fused_loss: bool = True          # minibatch loss + its gradient w.r.t. the policy outputs as one HIP launch (GPU only)
    graph_update: bool = True        # replay the minibatch step as a HIP graph when it is launch-bound (<= 2^17 samples)
    fused_policy: bool = True        # rollouts through the fused HIP forward (policy.FusedActorCritic) when the
                                     # net shape is supported and the env is on a GPU; torch modules otherwise
    fast_permutation: bool = True    # minibatch order from r6_permutation (a keyed bijection, one elementwise launch, part of
                                     # the epoch's graph) instead of torch.randperm (a sort: 0.22 ms per 2^20 rows); GPU only
    fused_rollout: bool = True       # with the fused forward: the n_steps x (policy forward, env step, bootstrap value) of a rollout as
                                     # ONE kernel launch (r6_rollout: state in registers throughout) where the env supports it
    graph_rollout: bool = True       # with the fused forward: a whole rollout (n_steps x (policy, env step, time-limit
                                     # bootstrap) + GAE) is recorded once into a HIP graph and replayed -- three launches
                                     # per step and no host work; the RNG counters advance through a device-side base


class _FusedPPOLoss(torch.autograd.Function):
    """SB3's PPO minibatch loss with the gradient w.r.t. (action mean, value, log_std) from ONE HIP launch
    (``r6_ppo_loss_grad``) instead of ~60 elementwise / reduction kernels forward and backward.  Returns
    ``(loss, stats)``; stats = [pg_loss, v_loss, entropy, clip_frac, approx_kl, n, -, -] (no gradient)."""

    @staticmethod
    def forward(ctx, mean, value, log_std, action, old_logp, adv, ret, adv_norm, clip, vf_coef, ent_coef):
        import ctypes as C

        from . import _lib

        lib = _lib.load()
        mean, value = mean.contiguous(), value.contiguous()
        action, old_logp, adv, ret = action.contiguous(), old_logp.contiguous(), adv.contiguous(), ret.contiguous()
        gm, gv = torch.empty_like(mean), torch.empty_like(value)
        gls = torch.zeros(3, dtype=torch.float32, device=mean.device)
        stats = torch.zeros(8, dtype=torch.float32, device=mean.device)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(mean.device):
            rc = lib.r6_ppo_loss_grad(ptr(mean), ptr(value), ptr(log_std.detach().contiguous()), ptr(action), ptr(old_logp), ptr(adv),
                                      ptr(ret), ptr(adv_norm), mean.shape[0], float(clip), float(vf_coef), float(ent_coef),
                                      ptr(gm), ptr(gv), ptr(gls), ptr(stats),
                                      C.c_void_p(torch.cuda.current_stream(mean.device).cuda_stream))
        _lib.check(rc, "r6_ppo_loss_grad", lib)
        ctx.save_for_backward(gm, gv, gls)
        ctx.mark_non_differentiable(stats)
        return stats[0] - ent_coef * stats[2] + vf_coef * stats[1], stats

    @staticmethod
    def backward(ctx, g, _g_stats):
        gm, gv, gls = ctx.saved_tensors
        return g * gm, g * gv, g * gls, None, None, None, None, None, None, None, None


class FusedMinibatchGrad:
    """One SB3 ``PPO.train`` minibatch -- ``evaluate_actions``, the clipped-surrogate / value / entropy loss and
    ``loss.backward()`` -- as ONE HIP launch (``r6_ppo_minibatch_grad``, csrc/r6_ppo_update.hip): parameter gradients come out
    on the matrix cores with fp32 accuracy (f16 hi + lo operand splits, fp32 accumulation), no activation goes to HBM and the minibatch rows are picked by index (no gathered
    copies).  Supported: the reference's topology, one shared tanh trunk obs -> 128 -> 64 with both heads on it
    (``ActorCritic(net_arch=(128, 64), shared_trunk=True)``), and SB3's default, two separate tanh MLPs obs -> 64 -> 64
    (``ActorCritic(net_arch=(64, 64), shared_trunk=False)``; BASELINE configs[4]).  The gradient tensors are persistent and
    become the parameters' ``.grad``, so clipping, the all-reduce and Adam run unchanged behind it."""

    ARCH = (128, 64)             # the shared trunk
    ARCH_SEPARATE = (64, 64)     # two MLPs

    @staticmethod
    def supports(policy) -> bool:
        pi = [m for m in policy.pi if isinstance(m, nn.Linear)]
        vf = [m for m in policy.vf if isinstance(m, nn.Linear)]
        if len(pi) != 3 or len(vf) != 3 or pi[0].in_features not in (13, 14) or pi[2].out_features != 3 or vf[2].out_features != 1 \
                or not policy.log_std.is_cuda:
            return False
        arch = (pi[0].out_features, pi[1].out_features)
        if bool(getattr(policy, "shared_trunk", False)):
            return arch == FusedMinibatchGrad.ARCH and pi[0] is vf[0] and pi[1] is vf[1]
        return arch == FusedMinibatchGrad.ARCH_SEPARATE and (vf[0].out_features, vf[1].out_features) == arch and pi[0] is not vf[0]

    def __init__(self, policy, lib_path=None):
        import ctypes as C

        from . import _lib
        from .params import R6PolicyGrad, R6PolicyNet

        if not self.supports(policy):
            raise ValueError("FusedMinibatchGrad needs ActorCritic(net_arch=(128, 64), shared_trunk=True) or "
                             "ActorCritic(net_arch=(64, 64), shared_trunk=False) on a GPU")
        self.lib = _lib.load(lib_path)
        pi = [m for m in policy.pi if isinstance(m, nn.Linear)]
        vf = [m for m in policy.vf if isinstance(m, nn.Linear)]
        self.shared = bool(getattr(policy, "shared_trunk", False))
        self.arch = (pi[0].out_features, pi[1].out_features)
        self.params = [pi[0].weight, pi[0].bias, pi[1].weight, pi[1].bias, pi[2].weight, pi[2].bias, vf[2].weight, vf[2].bias,
                       policy.log_std]
        self._net_names = ["pi_w1", "pi_b1", "pi_w2", "pi_b2", "pi_w3", "pi_b3", "vf_w3", "vf_b3", "log_std"]
        self._grad_names = ["w1", "b1", "w2", "b2", "pi_w3", "pi_b3", "vf_w3", "vf_b3", "log_std"]
        if not self.shared:      # the value MLP's own trunk
            self.params += [vf[0].weight, vf[0].bias, vf[1].weight, vf[1].bias]
            self._net_names += ["vf_w1", "vf_b1", "vf_w2", "vf_b2"]
            self._grad_names += ["vf_w1", "vf_b1", "vf_w2", "vf_b2"]
        for t in self.params:
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("policy parameters must be contiguous float32 CUDA tensors")
        self.device = self.params[0].device
        self.obs_dim = pi[0].in_features
        if {id(t) for t in self.params} != {id(t) for t in policy.parameters()}:
            raise ValueError("the policy has parameters outside its trunk(s) and two heads")
        # Parameters, gradients and Adam moments as FLAT arrays: every parameter becomes a view into one buffer (same
        # nn.Parameter objects, same state_dict), so the optimiser step is one launch over n floats (r6_adam_clip_step) and a
        # multi-rank gradient all-reduce is one bucket with no packing.
        sizes = [t.numel() for t in self.params]
        self.flat_p = torch.cat([t.detach().reshape(-1) for t in self.params]).contiguous()
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
        self.adam_step_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.grads, off = [], 0
        for t, k in zip(self.params, sizes):
            t.data = self.flat_p[off:off + k].view_as(t)
            self.grads.append(self.flat_g[off:off + k].view_as(t))
            off += k
        self.log_std_offset = sum(sizes[:8])      # log_std is the ninth tensor of the flat arrays
        self.stats = torch.zeros(16, dtype=torch.float32, device=self.device)   # [0..7] last launch, [8..15] running sums
        self.norm = torch.zeros(2, dtype=torch.float32, device=self.device)
        self._norm_scratch = torch.zeros(256, dtype=torch.float64, device=self.device)
        self.scratch = torch.empty(int(self.lib.r6_ppo_scratch_floats()), dtype=torch.float32, device=self.device)
        self._net, self._grad, self._C = R6PolicyNet(), R6PolicyGrad(), C
        for name, t in zip(self._grad_names, self.grads):
            setattr(self._grad, name, t.data_ptr())

    def adam_step(self, lr, max_grad_norm: float, log_std_min=None, betas=(0.9, 0.999), eps: float = 1e-5):
        """``clip_grad_norm_(parameters, max_grad_norm)`` + ``Adam.step()`` (+ the optional log-std floor) on the flat arrays
        in one launch.  ``lr``: float, or a device scalar tensor (read by the kernel at run time: graph-safe schedules)."""
        C = self._C
        lr_dev = C.c_void_p(lr.data_ptr()) if torch.is_tensor(lr) else None
        with torch.cuda.device(self.device):
            rc = self.lib.r6_adam_clip_step(
                C.c_void_p(self.flat_p.data_ptr()), C.c_void_p(self.flat_g.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
                C.c_void_p(self.exp_avg_sq.data_ptr()), self.flat_p.numel(), float(max_grad_norm if max_grad_norm is not None else 0.0),
                0.0 if torch.is_tensor(lr) else float(lr), lr_dev, float(betas[0]), float(betas[1]), float(eps),
                C.c_void_p(self.adam_step_count.data_ptr()), self.log_std_offset, 3 if log_std_min is not None else 0,
                float(log_std_min if log_std_min is not None else 0.0), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        from . import _lib
        _lib.check(rc, "r6_adam_clip_step", self.lib)

    def adv_norm(self, adv, index, n_batches: int = 1):
        """[mean, 1 / (std + 1e-8)] of ``adv[index]`` (SB3's per-minibatch advantage normalisation) in two small launches.
        ``n_batches`` > 1: ``index`` holds that many consecutive minibatches of equal size (an epoch's permutation);
        returns a (n_batches, 2) tensor, one pair per minibatch."""
        C = self._C
        total = int(index.numel()) if index is not None else int(adv.numel())
        if n_batches < 1 or total % n_batches:
            raise ValueError("the index does not split into n_batches equal minibatches")
        if n_batches == 1:
            out, scratch = self.norm, self._norm_scratch
        else:
            # one pair of buffers per batch count, never freed: a recorded graph keeps reading the ones it was captured with
            cache = self.__dict__.setdefault("_norm_bufs", {})
            if n_batches not in cache:
                cache[n_batches] = (torch.zeros((n_batches, 2), dtype=torch.float32, device=self.device),
                                    torch.zeros(256 * n_batches, dtype=torch.float64, device=self.device))
            out, scratch = cache[n_batches]
        with torch.cuda.device(self.device):
            rc = self.lib.r6_minibatch_adv_norm(C.c_void_p(adv.data_ptr()), None if index is None else C.c_void_p(index.data_ptr()),
                                                total // n_batches, n_batches, C.c_void_p(out.data_ptr()), C.c_void_p(scratch.data_ptr()),
                                                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        from . import _lib
        _lib.check(rc, "r6_minibatch_adv_norm", self.lib)
        return out

    def __call__(self, obs, action, old_logp, adv, ret, index, adv_norm, clip, vf_coef, ent_coef):
        """Writes the gradients of the minibatch ``index`` (int64 rows of the rollout tensors; None = all rows) into the
        parameters' ``.grad`` and returns the stats tensor [pg_loss, v_loss, entropy, clip_frac, approx_kl, n, -, -] of this
        launch, followed by the running sums of the same eight over all launches since ``stats[8:]`` was last zeroed."""
        C, net = self._C, self._net
        net.obs_dim, net.hidden1, net.hidden2, net.shared = self.obs_dim, self.arch[0], self.arch[1], int(self.shared)
        for name, t in zip(self._net_names, self.params):
            setattr(net, name, t.data_ptr())
        n = int(index.numel()) if index is not None else int(obs.shape[0])
        n_rows = int(obs.shape[0])
        if index is not None and n_rows >= 1 << 31:
            raise ValueError(f"rollout buffer of {n_rows} rows: the fused minibatch gradient gathers rows through 32-bit row numbers "
                             "(< 2^31 rows); use a smaller n_steps x num_envs per rank")
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        with torch.cuda.device(self.device):
            rc = self.lib.r6_ppo_minibatch_grad(C.byref(net), ptr(obs), ptr(action), ptr(old_logp), ptr(adv), ptr(ret), ptr(index), n,
                                                n_rows, ptr(adv_norm), float(clip), float(vf_coef), float(ent_coef), C.byref(self._grad),
                                                ptr(self.stats), ptr(self.scratch), self.scratch.numel(),
                                                C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        from . import _lib
        _lib.check(rc, "r6_ppo_minibatch_grad", self.lib)
        for p_, g_ in zip(self.params, self.grads):
            p_.grad = g_
        return self.stats


class _TallLinearFn(torch.autograd.Function):
    """``F.linear`` whose weight gradient is a hand-split-K product.  In a PPO minibatch of 2^20 samples the
    weight gradient is a (64 x 2^20) @ (2^20 x 64) GEMM; the BLAS heuristics give it one tile column and
    1.5 ms (2.0 ms for the 3-row head), a batched product over 256 row chunks + a sum takes 0.11 ms
    (scripts/gemm_probe.py).  Forward and input gradient are the plain GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = g @ weight if ctx.needs_input_grad[0] else None
        rows = x.shape[0]
        c = 256
        while c > 1 and (rows % c or rows // c < 512):
            c //= 2
        if c > 1:
            gw = torch.bmm(g.reshape(c, rows // c, -1).transpose(1, 2), x.reshape(c, rows // c, -1)).sum(0)
        else:
            gw = g.t() @ x
        return gx, gw, g.sum(0)


class TallLinear(nn.Linear):
    """``nn.Linear`` (same parameters, same state_dict) for tall inputs, see ``_TallLinearFn``."""

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= 1 << 15 and x.is_cuda and torch.is_grad_enabled():
            return _TallLinearFn.apply(x, self.weight, self.bias)
        return super().forward(x)


def _mlp(sizes, out_dim, out_gain):
    layers, last = [], sizes[0]
    for h in sizes[1:]:
        lin = TallLinear(last, h)
        nn.init.orthogonal_(lin.weight, gain=math.sqrt(2))
        nn.init.zeros_(lin.bias)
        layers += [lin, nn.Tanh()]
        last = h
    head = TallLinear(last, out_dim)
    nn.init.orthogonal_(head.weight, gain=out_gain)
    nn.init.zeros_(head.bias)
    return nn.Sequential(*layers, head)


class ActorCritic(nn.Module):
    """SB3 1.6 ``ActorCriticPolicy`` with a diagonal Gaussian action distribution.

    ``shared_trunk=False``: separate tanh MLPs for policy and value -- SB3's default ``net_arch`` and the
    ``[dict(pi=[..], vf=[..])]`` form.  ``shared_trunk=True``: ONE trunk feeding both heads -- what SB3 1.6
    builds for a plain list, i.e. for the reference's ``policy_kwargs={'net_arch': [128, 64]}``
    (main_6DOF.py:113; its checkpoints hold ``mlp_extractor.shared_net.*``).  Either way ``self.pi`` /
    ``self.vf`` are the full obs -> mean / obs -> value stacks (sharing the trunk's modules when shared)."""

    def __init__(self, obs_dim: int, act_dim: int, net_arch=(64, 64), log_std_init: float = 0.0, shared_trunk: bool = False):
        super().__init__()
        self.shared_trunk = bool(shared_trunk)
        if self.shared_trunk:
            trunk = list(_mlp([obs_dim, *net_arch[:-1]], net_arch[-1], math.sqrt(2))) + [nn.Tanh()]
            pi_head, vf_head = TallLinear(net_arch[-1], act_dim), TallLinear(net_arch[-1], 1)
            nn.init.orthogonal_(pi_head.weight, gain=0.01)
            nn.init.orthogonal_(vf_head.weight, gain=1.0)
            nn.init.zeros_(pi_head.bias)
            nn.init.zeros_(vf_head.bias)
            self.pi = nn.Sequential(*trunk, pi_head)
            self.vf = nn.Sequential(*trunk, vf_head)      # same trunk modules: parameters are shared
        else:
            self.pi = _mlp([obs_dim, *net_arch], act_dim, 0.01)
            self.vf = _mlp([obs_dim, *net_arch], 1, 1.0)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std_init)))

    def _forward(self, obs):
        """(action mean, value); the trunk is evaluated once when it is shared."""
        if self.shared_trunk:
            latent = self.pi[:-1](obs)
            return self.pi[-1](latent), self.vf[-1](latent).squeeze(-1)
        return self.pi(obs), self.vf(obs).squeeze(-1)

    def value(self, obs):
        return self.vf(obs).squeeze(-1)

    def _logp(self, mean, actions):
        var = torch.exp(2 * self.log_std)
        return (-0.5 * ((actions - mean) ** 2 / var + 2 * self.log_std + math.log(2 * math.pi))).sum(-1)

    @torch.no_grad()
    def act(self, obs, generator=None, deterministic=False):
        mean, value = self._forward(obs)
        if deterministic:
            actions = mean
        else:
            noise = torch.randn(mean.shape, device=mean.device, dtype=mean.dtype, generator=generator)
            actions = mean + noise * torch.exp(self.log_std)
        return actions, value, self._logp(mean, actions)

    def evaluate(self, obs, actions):
        mean, value = self._forward(obs)
        entropy = (0.5 + 0.5 * math.log(2 * math.pi) + self.log_std).sum().expand(obs.shape[0])
        return value, self._logp(mean, actions), entropy


def compute_gae(rewards, values, dones, last_values, gamma: float, lam: float):
    """SB3 ``RolloutBuffer.compute_returns_and_advantage``.  rewards/values/dones: (T, N) with
    dones[t] = episode ended AT step t (so the value after it is not bootstrapped); last_values: (N,)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    last = torch.zeros_like(last_values)
    for t in reversed(range(T)):
        next_v = last_values if t == T - 1 else values[t + 1]
        nonterm = 1.0 - dones[t]
        delta = rewards[t] + gamma * next_v * nonterm - values[t]
        last = delta + gamma * lam * nonterm * last
        adv[t] = last
    return adv, adv + values


class PPO:
    def __init__(self, env, config: Optional[PPOConfig] = None, policy: Optional[ActorCritic] = None):
        self.env = env
        self.cfg = cfg = config or PPOConfig()
        self.device = env.device
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # the gradient all-reduce is issued whenever several ranks exist -- or one rank with R6_FORCE_PROCESS_GROUP=1 (a builder's
        # rehearsal of the multi-rank code path, collectives included, on the one GPU a box has)
        self._collectives = self.world > 1 or (dist.is_available() and dist.is_initialized() and os.environ.get("R6_FORCE_PROCESS_GROUP") == "1")
        self.graph_collective_error = None   # what RCCL / torch said if the all-reduce refused to be captured into the epoch graph
        torch.manual_seed(cfg.seed)          # identical initial weights on every rank
        self.policy = (policy or ActorCritic(env.obs_dim, 3, cfg.net_arch, cfg.log_std_init, cfg.shared_trunk)).to(self.device)
        capt = self.device.type == "cuda" and cfg.graph_update
        # a tensor learning rate is read by the (graphed) Adam step at replay time, so a schedule can move it
        lr = torch.tensor(float(cfg.learning_rate), device=self.device) if capt else cfg.learning_rate
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=lr, eps=1e-5, capturable=capt)
        self._graph, self._graph_key, self._eager_steps = None, None, 0
        self._perm_count, self._perm_base = 0, None
        rank = dist.get_rank() if self.world > 1 else 0
        self.gen = torch.Generator(device=self.device).manual_seed(cfg.seed + 1000003 * rank)
        self._perm_seed = (cfg.seed + 1000003 * rank) * 2654435761 + 97
        if self.device.type == "cuda":
            self._perm_base = torch.zeros((), dtype=torch.int64, device=self.device)
        n, T = env.num_envs, cfg.n_steps
        if getattr(env, "max_episode_steps", 0) and (getattr(env, "flags", None) is None or getattr(env, "final_obs", None) is None):
            # without the truncation flag and the terminal observation the time-limit bootstrap gamma * V(s_T) cannot
            # be applied and truncated episodes would silently count as true terminations
            raise ValueError("PPO needs env.flags and env.final_obs when max_episode_steps > 0: build the env with "
                             "record_flags=True and auto_reset=True")
        # Shards may be ragged (dist.shard_range hands out counts that differ by one).  Everything a collective depends
        # on is agreed ONCE here: the number of minibatch steps per epoch (the minimum over ranks, so every rank issues
        # the same number of gradient all-reduces) and each rank's weight in the gradient average (its share of the
        # samples of a global minibatch).
        self.local_envs = n
        total = T * n
        self._bs = min(cfg.batch_size, total)
        self._steps_per_epoch = total // self._bs
        self.global_envs, self._grad_weight = n, 1.0
        if self.world > 1:
            t = torch.zeros((self.world, 3), dtype=torch.float64)
            t[rank] = torch.tensor([n, self._bs, self._steps_per_epoch], dtype=torch.float64)
            t = self._all_reduce(t)
            self.global_envs = int(t[:, 0].sum())
            self._grad_weight = float(self._bs / t[:, 1].sum())
            self._steps_per_epoch = int(t[:, 2].min())
        f32 = dict(dtype=torch.float32, device=self.device)
        # T + 1 observation rows: row t is what action t was computed from, row T the observation after the last step (the next
        # rollout's row 0); the fused rollout kernel writes rows 1..T in one launch
        self._buf_obs_ext = torch.zeros((T + 1, n, env.obs_dim), **f32)
        self.buf_obs = self._buf_obs_ext[:T]
        self.buf_act = torch.zeros((T, n, 3), **f32)
        self.buf_logp = torch.zeros((T, n), **f32)
        self.buf_val = torch.zeros((T, n), **f32)
        self.buf_rew = torch.zeros((T, n), **f32)
        self.buf_done = torch.zeros((T, n), dtype=torch.uint8, device=self.device)
        self.adv = torch.zeros((T, n), **f32)
        self.ret = torch.zeros((T, n), **f32)
        self._obs = None
        self.num_timesteps = 0
        self.fused_update = None
        if cfg.fused_update and self.device.type == "cuda" and FusedMinibatchGrad.supports(self.policy) and env.obs_dim in (13, 14):
            self.fused_update = FusedMinibatchGrad(self.policy)
        self.fused = None
        self._rollout_step = 0
        # launch-only rollout (set up below when the fused forward exists): time-limit bootstrap folded into r6_gae,
        # env outputs written straight into the rollout buffer, the whole rollout replayed as a HIP graph
        self._time_limit, self._direct, self._in_kernel_bootstrap = False, None, False
        self._one_launch_rollout = False
        self.buf_flags = self.buf_boot = None
        self._rollout_graph, self._rollouts = None, 0
        if cfg.fused_policy and self.device.type == "cuda":
            from .policy import FusedActorCritic

            if self._fusable() and env.obs_dim in (13, 14):
                self.fused = FusedActorCritic(self.policy, seed=cfg.seed, env_offset=getattr(env, "env_offset", 0))
                self._act_clip = torch.zeros((n, 3), **f32)
                self._boot_out = (torch.zeros((n, 3), **f32), torch.zeros((n, 3), **f32), torch.zeros((n,), **f32),
                                  torch.zeros((n,), **f32))
                self._last_val = torch.zeros((n,), **f32)
                # time-limit bootstrap on the device: per step the truncation flags and V(terminal observation) of the
                # flagged rows (written by a masked forward, read by r6_gae only where the flag is set)
                self._time_limit = bool(getattr(env, "max_episode_steps", 0))
                self.buf_flags = torch.zeros((T, n), dtype=torch.uint8, device=self.device) if self._time_limit else None
                self.buf_boot = torch.zeros((T, n), **f32) if self._time_limit else None
                # env outputs land directly in the rollout buffer when its rows are 16-byte aligned (BatchedRocket6DOF.step_into)
                self._direct = hasattr(env, "step_into") and (n * env.obs_dim) % 4 == 0 and not getattr(env, "numpy_io", False)
                # device-resident part of the step counters (graph replays): attached to the env only while the rollout's
                # launches are issued or recorded -- an eager env.step() between rollouts must not add a stale base
                self._step_base = torch.zeros((), dtype=torch.int64, device=self.device)
                # the whole rollout as one launch (r6_rollout) where env and net are among its instances
                p_ = getattr(env, "params", None)
                self._one_launch_rollout = bool(cfg.fused_rollout and self._direct and hasattr(env, "rollout") and p_ is not None
                                                and n <= self.ONE_LAUNCH_ROLLOUT_MAX_ENVS
                                                and int(p_.n_substeps) == 1 and int(p_.integrator) in (0, 1)
                                                and not (int(p_.integrator) == 0 and int(p_.shaping) == 1))

    # r6_rollout keeps one 32-env tile per wave in registers for the whole rollout (built for few waves per SIMD): measured on
    # MI355X, 64-step rollouts: 16 k envs 2.13 -> 0.46 ms, 65 k 2.27 -> 0.87, 262 k 3.84 -> 3.27, 2^20 envs 1.3x SLOWER than the
    # three launches per step (scripts/rollout_fused_bench.py) -- large batches keep the launch sequence
    ONE_LAUNCH_ROLLOUT_MAX_ENVS = 1 << 18

    def _all_reduce(self, t: torch.Tensor, op=None) -> torch.Tensor:
        """Sum (default) over ranks of a small host tensor: through the device for RCCL, on the host for gloo."""
        op = dist.ReduceOp.SUM if op is None else op
        if dist.get_backend() == "nccl":
            d = t.to(self.device)
            dist.all_reduce(d, op=op)
            return d.cpu()
        dist.all_reduce(t, op=op)
        return t

    def _fusable(self) -> bool:
        from .policy import SUPPORTED_ARCHS

        lin = [m for m in self.policy.pi if isinstance(m, nn.Linear)]
        return len(lin) == 3 and (lin[0].out_features, lin[1].out_features) in SUPPORTED_ARCHS

    # ------------------------------------------------------------------------------------------
    def _rollout_launches(self):
        """One rollout + GAE as a fixed sequence of launches with no host decision in it (so it can be recorded into a
        HIP graph): per step the fused policy forward, the env step writing straight into the rollout buffer, and -- with
        a time limit -- V(terminal observation) of the rows the step truncated (a masked forward: tiles without a
        truncated row are skipped on the device); then V(last obs) and ``r6_gae``."""
        env, cfg, T = self.env, self.cfg, self.cfg.n_steps
        base = self._step_base
        env.set_step_base(base)
        try:
            if self._one_launch_rollout:
                self.buf_obs[0].copy_(self._obs)
                env.rollout(self.fused, T, reward=self.buf_rew, done=self.buf_done, obs_seq=self._buf_obs_ext, raw_action=self.buf_act,
                            value=self.buf_val, logp=self.buf_logp, flags=self.buf_flags if self._time_limit else None,
                            boot_value=self.buf_boot if self._time_limit else None, policy_step_index=self._rollout_step + 1,
                            policy_step_base=base)
                self._rollout_step += T
                if self._obs.data_ptr() != self._buf_obs_ext[T].data_ptr():
                    self._obs.copy_(self._buf_obs_ext[T])
            else:
                self._rollout_steps(env, T, base)
        finally:
            env.set_step_base(None)
        bo = self._boot_out
        self.fused(self._obs, deterministic=True, step_index=0, out=(bo[0], bo[1], self._last_val, bo[3]))
        self._gae(self._last_val)

    def _rollout_steps(self, env, T, base):
        self.buf_obs[0].copy_(self._obs)
        for t in range(T):
            self._rollout_step += 1
            self.fused(self.buf_obs[t], step_index=self._rollout_step, step_base=base,
                       out=(self._act_clip, self.buf_act[t], self.buf_val[t], self.buf_logp[t]))
            env.step_into(self._act_clip, obs=self.buf_obs[t + 1] if t + 1 < T else self._obs, reward=self.buf_rew[t],
                          done=self.buf_done[t], flags=self.buf_flags[t] if self._time_limit else None)
            if self._time_limit:
                bo = self._boot_out
                self.fused(env.final_obs, deterministic=True, step_index=0, out=(bo[0], bo[1], self.buf_boot[t], bo[3]),
                           row_flags=self.buf_flags[t], flag_mask=FLAG_TRUNCATED)

    def _gae(self, last_values):
        """GAE(lambda) into ``self.adv`` / ``self.ret``: one launch of ``r6_gae`` on the GPU (time-limit bootstrap folded
        in when ``buf_flags`` / ``buf_boot`` exist), the torch loop ``compute_gae`` otherwise."""
        cfg, (T, n) = self.cfg, self.buf_rew.shape
        lib = getattr(self.fused, "lib", None) or getattr(self.fused_update, "lib", None)
        if lib is None or self.device.type != "cuda":
            adv, ret = compute_gae(self.buf_rew, self.buf_val, self.buf_done.to(torch.float32), last_values, cfg.gamma, cfg.gae_lambda)
            self.adv.copy_(adv)      # persistent buffers: the graphed update step reads these addresses
            self.ret.copy_(ret)
            return
        from . import _lib

        tl = self.fused is not None and self._time_limit and self._in_kernel_bootstrap
        with torch.cuda.device(self.device):
            rc = lib.r6_gae(self.buf_rew.data_ptr(), self.buf_val.data_ptr(), self.buf_done.data_ptr(), last_values.data_ptr(), T, n,
                            cfg.gamma, cfg.gae_lambda, self.buf_flags.data_ptr() if tl else None, FLAG_TRUNCATED,
                            self.buf_boot.data_ptr() if tl else None, self.adv.data_ptr(), self.ret.data_ptr(),
                            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(rc, "r6_gae", lib)

    @torch.no_grad()
    def collect_rollouts(self):
        env, cfg = self.env, self.cfg
        if self._obs is None:
            first = env.reset()
            if torch.is_tensor(first) and first.device == self.device:
                self._obs = self._buf_obs_ext[cfg.n_steps]      # lives in the rollout buffer's extra row
                self._obs.copy_(first)
            else:
                self._obs = first.clone()
        T = cfg.n_steps
        self._in_kernel_bootstrap = False
        if self.fused is not None and self._direct:
            self._in_kernel_bootstrap = True
            if cfg.graph_rollout and self._rollouts >= 1:        # the first rollout runs eagerly (one-time set-up in the launch paths)
                if self._rollout_graph is None:
                    # recording runs the Python side once and executes nothing: the launches keep the step counters they
                    # were recorded with (c0 + 1 .. c0 + T) for ever, and the host counters are put back afterwards
                    self._graph_c0, self._graph_r0 = env._step_count, self._rollout_step
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        self._rollout_launches()
                    self._rollout_graph = g
                    env._step_count, self._rollout_step = self._graph_c0, self._graph_r0
                # every replay continues the RNG streams through the device-side base, set from the HOST counter: eager
                # env.step / step_k / evaluation calls between rollouts advance that counter too, so no (env, step) key of the
                # reset stream is ever drawn twice, and the host counters (LazyInfos' staleness check reads them) stay true
                self._step_base.fill_(env._step_count - self._graph_c0)
                self._rollout_graph.replay()
                env._step_count += T
                self._rollout_step += T
            else:
                self._step_base.zero_()
                self._rollout_launches()
            self._rollouts += 1
            self.num_timesteps += cfg.n_steps * self.global_envs
            return
        self.buf_obs[0].copy_(self._obs)
        for t in range(T):
            if self.fused is not None:
                # one launch: sample, log-prob, value and the Box clip, written straight into the rollout buffer
                self._rollout_step += 1
                self.fused(self.buf_obs[t], step_index=self._rollout_step,
                           out=(self._act_clip, self.buf_act[t], self.buf_val[t], self.buf_logp[t]))
                clipped = self._act_clip
            else:
                actions, values, logp = self.policy.act(self.buf_obs[t], generator=self.gen)
                self.buf_act[t].copy_(actions)
                self.buf_val[t].copy_(values)
                self.buf_logp[t].copy_(logp)
                clipped = actions.clamp(-1.0, 1.0).contiguous()                  # SB3 clips to the Box
            obs, rew, done, _ = env.step(clipped)
            rew_t = self.buf_rew[t]
            rew_t.copy_(rew)                  # env.obs / rew / done are overwritten by the next step
            if env.max_episode_steps and env.flags is not None and env.final_obs is not None:
                trunc = (env.flags & FLAG_TRUNCATED) != 0          # TimeLimit.truncated -> bootstrap
                if self.fused is not None:
                    # no host decision (a .any() would drain the queue every step): V(final_obs) for every env in
                    # one fused launch, masked by the truncation flag; rows of envs that did not end are ignored
                    self.fused(env.final_obs, deterministic=True, step_index=0, out=self._boot_out)
                    rew_t.addcmul_(self._boot_out[2], trunc.to(torch.float32), value=cfg.gamma)
                elif bool(trunc.any()):
                    idx = trunc.nonzero(as_tuple=True)[0]
                    rew_t[idx] += cfg.gamma * self.policy.value(env.final_obs[idx])
            self.buf_done[t].copy_(done)
            (self.buf_obs[t + 1] if t + 1 < T else self._obs).copy_(obs)
        self.num_timesteps += cfg.n_steps * self.global_envs     # the whole job, ragged shards included
        self._gae(self.policy.value(self._obs).contiguous())

    def _sync_grads(self):
        if not self._collectives:
            return
        grads = [p.grad for p in self.policy.parameters() if p.grad is not None]
        # one bucket (~10^4 parameters); each rank's gradient (a mean over ITS minibatch) enters with its share of the
        # global minibatch's samples, so the result is the mean over all samples also when shards are ragged
        flat = torch.cat([g.reshape(-1) for g in grads]) * self._grad_weight
        if dist.get_backend() != "nccl":
            flat = flat.cpu()
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat = flat.to(grads[0].device)
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def _minibatch_step(self, mb, flat, acc, norm=None):
        """One SB3 ``PPO.train`` minibatch: loss, backward, (all-reduce), grad-norm clip, Adam; stats into ``acc`` (the fused
        kernel keeps its own running sums: ``FusedMinibatchGrad.stats[8:]``).  ``norm``: this minibatch's advantage
        normalisation when the caller computed it for a whole epoch at once."""
        cfg = self.cfg
        obs, act, old_logp, adv, ret = flat
        if self.fused_update is not None:
            # forward + loss + backward of the whole minibatch in one launch; rows picked by index inside the kernel
            if norm is None and cfg.normalize_advantage and mb.numel() > 1:
                norm = self.fused_update.adv_norm(adv, mb)
            self.fused_update(obs, act, old_logp, adv, ret, mb, norm, cfg.clip_range, cfg.vf_coef, cfg.ent_coef)
            if self._collectives:     # one bucket, already flat: each rank's mean gradient enters with its share of the samples
                g = self.fused_update.flat_g
                g.mul_(self._grad_weight)
                if dist.get_backend() == "nccl":
                    dist.all_reduce(g, op=dist.ReduceOp.SUM)
                else:
                    h = g.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    g.copy_(h)
            # clip_grad_norm_ + Adam (SB3: eps 1e-5) + the log-std floor: one launch on the flat arrays
            self.fused_update.adam_step(self.opt.param_groups[0]["lr"], cfg.max_grad_norm, cfg.log_std_min,
                                        betas=self.opt.param_groups[0]["betas"], eps=self.opt.param_groups[0]["eps"])
            return
        a = adv[mb]
        fused = cfg.fused_loss and self.device.type == "cuda"
        if fused:
            norm = None
            if cfg.normalize_advantage and mb.numel() > 1:
                std, mean = torch.std_mean(a)                      # one Welford pass instead of two reductions
                norm = torch.stack([mean, 1.0 / (std + 1e-8)])
            mean, values = self.policy._forward(obs[mb])
            loss, st = _FusedPPOLoss.apply(mean, values, self.policy.log_std, act[mb], old_logp[mb], a, ret[mb], norm,
                                           cfg.clip_range, cfg.vf_coef, cfg.ent_coef)
        else:
            if cfg.normalize_advantage and mb.numel() > 1:
                a = (a - a.mean()) / (a.std() + 1e-8)
            values, logp, entropy = self.policy.evaluate(obs[mb], act[mb])
            ratio = torch.exp(logp - old_logp[mb])
            pg = -torch.min(a * ratio, a * ratio.clamp(1 - cfg.clip_range, 1 + cfg.clip_range)).mean()
            v_loss = torch.nn.functional.mse_loss(ret[mb], values)
            ent = -entropy.mean()
            loss = pg + cfg.ent_coef * ent + cfg.vf_coef * v_loss
        self.opt.zero_grad(set_to_none=True)   # fresh gradient buffers: no fill + accumulate-add launch per parameter (graph-safe: replays reuse the pool addresses)
        loss.backward()
        self._sync_grads()
        torch.nn.utils.clip_grad_norm_(self.policy.parameters(), cfg.max_grad_norm)
        self.opt.step()
        with torch.no_grad():
            if cfg.log_std_min is not None:
                self.policy.log_std.clamp_(min=cfg.log_std_min)
            if fused:
                acc += st[:5]
            else:
                acc += torch.stack([pg, v_loss, -ent, ((ratio - 1).abs() > cfg.clip_range).float().mean(),
                                    ((ratio - 1) - (logp - old_logp[mb])).mean()])

    def _permute(self, out: torch.Tensor, base: Optional[torch.Tensor] = None):
        """``out[:] = `` a fresh pseudo-random permutation of 0..len(out)-1 (``r6_permutation``; host counter + device base key
        it, as for the rollout's RNG streams)."""
        from . import _lib

        lib = self.fused_update.lib
        self._perm_count += 1
        with torch.cuda.device(self.device):
            rc = lib.r6_permutation(out.data_ptr(), out.numel(), self._perm_seed, self._perm_count, None if base is None else base.data_ptr(),
                                    C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(rc, "r6_permutation", lib)
        return out

    def _graphed_steps(self, flat, bs, chunk, total=0):
        """``chunk`` consecutive minibatch steps captured once into a HIP graph (static index / stats buffers) and
        replayed: at minibatches of <= 2^17 samples the ~250 kernels of a torch step are launch-bound in eager mode
        (2.1 ms per step at 65 536 samples against ~0.4 ms of GPU work), and with the fused kernels (4 launches,
        ~0.09 ms per step) even one index copy + one graph launch per STEP is a fifth of the time -- so a whole
        epoch's steps (up to 32) share one replay and one copy of the permutation.

        Several ranks (round 5; round 4 kept multi-rank updates eager, 160 launch-bound steps per iteration in exactly the
        configuration BASELINE configs[4] names): the gradient all-reduce of every step is captured INTO the graph -- RCCL's
        collectives are stream-ordered kernel launches like any other; torch's process group forks to its own stream and joins
        back inside the capture.  Every rank captures the same sequence (the step count per epoch is agreed in __init__), the
        communicator exists by then (the first three steps of a job always run eagerly).  If the stack refuses the capture, the
        error text is kept in ``graph_collective_error``, None is returned and the update stays eager for the rest of the job."""
        key = (bs, chunk, total) + tuple(t.data_ptr() for t in flat)
        if self._graph is not None and self._graph_key == key:
            return self._graph
        # total > 0: the graph also draws the epoch's permutation of the `total` rollout rows into the index buffer
        self._mb_idx = torch.zeros(max(chunk * bs, total), dtype=torch.int64, device=self.device)
        self._acc = torch.zeros(5, dtype=torch.float32, device=self.device)
        batched_norm = self.fused_update is not None and self.cfg.normalize_advantage and bs > 1
        if batched_norm:
            self.fused_update.adv_norm(flat[3], self._mb_idx[:chunk * bs], chunk)     # allocates its buffers outside the capture
        self._graph = torch.cuda.CUDAGraph()
        idx = self._mb_idx[:chunk * bs]
        perm_count0 = self._perm_count
        try:
            # thread_local: the process group's watchdog thread may query its events while this thread captures
            with torch.cuda.graph(self._graph, **({"capture_error_mode": "thread_local"} if self._collectives else {})):
                if total:
                    self._permute(self._mb_idx[:total], self._perm_base)
                    self._perm_base.add_(1)
                norms = None
                if batched_norm:
                    norms = self.fused_update.adv_norm(flat[3], idx, chunk).view(chunk, 2)   # the whole chunk's minibatches at once
                for c in range(chunk):
                    self._minibatch_step(self._mb_idx[c * bs:(c + 1) * bs], flat, self._acc, None if norms is None else norms[c])
        except Exception as e:   # noqa: BLE001
            if not self._collectives:
                raise
            self.graph_collective_error = f"{type(e).__name__}: {e}"[:600]
            self._graph, self._graph_key, self._perm_count = None, None, perm_count0
            torch.cuda.synchronize(self.device)
            return None
        self._graph_key = key
        return self._graph

    def _graph_chunk(self, steps: int) -> int:
        """Steps per graph replay: the largest divisor of the epoch's step count up to 32 (fused kernels only: a torch
        step is ~250 nodes and already amortises its launch)."""
        if self.fused_update is None:
            return 1
        return max(c for c in range(1, min(steps, 32) + 1) if steps % c == 0)

    def train(self) -> Dict[str, float]:
        cfg = self.cfg
        T, n = self.buf_rew.shape
        total = T * n
        flat = (self.buf_obs.reshape(total, -1), self.buf_act.reshape(total, 3), self.buf_logp.reshape(total),
                self.adv.reshape(total), self.ret.reshape(total))
        bs = self._bs
        keys = ("pg_loss", "v_loss", "entropy", "clip_frac", "approx_kl")
        # several ranks: only over RCCL (a gloo all-reduce is host code), and only while the capture has not been refused
        coll_ok = not self._collectives or (dist.get_backend() == "nccl" and self.graph_collective_error is None)
        use_graph = cfg.graph_update and self.device.type == "cuda" and coll_ok and bs <= (1 << 17)
        acc = torch.zeros(len(keys), dtype=torch.float32, device=self.device)   # summed on the device: no per-batch sync
        if self.fused_update is not None:
            self.fused_update.stats[8:].zero_()      # the gradient kernel keeps the running sums itself
        n_batches = 0
        steps = self._steps_per_epoch                              # the same count on every rank (agreed in __init__)
        chunk = self._graph_chunk(steps) if use_graph else 1
        fast = cfg.fast_permutation and self.fused_update is not None
        if fast and (getattr(self, "_perm_buf", None) is None or self._perm_buf.numel() != total):
            self._perm_buf = torch.zeros(total, dtype=torch.int64, device=self.device)
        for _ in range(cfg.n_epochs):
            if fast and use_graph and chunk == steps and self._eager_steps >= 3:
                g = self._graphed_steps(flat, bs, chunk, total)           # permutation + every minibatch step of the epoch: one replay
                if g is not None:
                    g.replay()
                    n_batches += steps
                    continue
                use_graph = False                                         # the capture was refused (several ranks): eager from here on
            perm = self._permute(self._perm_buf) if fast else torch.randperm(total, device=self.device, generator=self.gen)
            norms = None
            if self.fused_update is not None and cfg.normalize_advantage and bs > 1 and steps > 1:
                # eager steps (multi-rank runs, warm-up): the epoch's advantage normalisations in one batched call all the same
                norms = self.fused_update.adv_norm(flat[3], perm[:steps * bs], steps).view(steps, 2)
            s = 0
            while s < steps:
                # the first steps run eagerly (allocator / Adam state warm-up), and so does the rest of that epoch when
                # it is shorter than a chunk
                g = self._graphed_steps(flat, bs, chunk) if (use_graph and self._eager_steps >= 3 and s + chunk <= steps) else None
                if g is None and self.graph_collective_error is not None:
                    use_graph = False
                if g is not None:
                    self._mb_idx[:chunk * bs].copy_(perm[s * bs:(s + chunk) * bs])
                    g.replay()
                    s += chunk
                    n_batches += chunk
                else:
                    self._minibatch_step(perm[s * bs:(s + 1) * bs], flat, acc, None if norms is None else norms[s])
                    self._eager_steps += 1
                    s += 1
                    n_batches += 1
        if use_graph and self._graph is not None:
            acc += self._acc
            self._acc.zero_()
        if self.fused_update is not None:
            acc += self.fused_update.stats[8:13]
        out = (acc / max(n_batches, 1)).tolist()
        return dict(zip(keys, out))

    def learn(self, total_timesteps: int, log=None):
        """Alternates rollouts and updates until ``total_timesteps`` env steps (whole job) are done.
        Returns the per-iteration log (episode summaries are all-gathered across ranks)."""
        from .dist import gather_episode_stats, summarize

        history = []
        while self.num_timesteps < total_timesteps:
            if self.cfg.final_learning_rate is not None:
                frac = min(self.num_timesteps / max(total_timesteps, 1), 1.0)
                lr = self.cfg.learning_rate + (self.cfg.final_learning_rate - self.cfg.learning_rate) * frac
                for g in self.opt.param_groups:
                    if torch.is_tensor(g["lr"]):
                        g["lr"].fill_(lr)
                    else:
                        g["lr"] = lr
            self.collect_rollouts()
            info = self.train()
            if self.env.stats is not None:
                tot, _ = gather_episode_stats(self.env.episode_stats(), device=self.device)
                info.update(summarize(tot))
            info["timesteps"] = self.num_timesteps
            history.append(info)
            if log is not None:
                log(info)
        return history
