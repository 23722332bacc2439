"""CPU-only tests of the on-device PPO plumbing (rl_rocket_6dof_amd/ppo.py) on a tiny torch stand-in
environment: GAE against a naive reference, the Gaussian policy against torch.distributions, the
time-limit bootstrap, learning progress, and gradient averaging across 2 gloo ranks."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from rl_rocket_6dof_amd.params import FLAG_TRUNCATED
from rl_rocket_6dof_amd.ppo import PPO, ActorCritic, PPOConfig, compute_gae


class ToyEnv:
    """N points on a line: obs = (x, 0...), action[0] = velocity command; reward = -|x|; episodes end by the
    time limit only (so the truncation bootstrap is exercised)."""

    def __init__(self, n, obs_dim=13, T=20, seed=0):
        self.num_envs, self.obs_dim, self.device = n, obs_dim, torch.device("cpu")
        self.max_episode_steps = T
        self.g = torch.Generator().manual_seed(seed)
        self.flags = torch.zeros(n, dtype=torch.uint8)
        self.final_obs = torch.zeros((n, obs_dim))
        self.stats = None
        self.x = self.v = self.t = None

    def _obs(self):
        o = torch.zeros((self.num_envs, self.obs_dim))
        o[:, 0], o[:, 1] = self.x, self.v
        return o

    def reset(self):
        self.x = torch.rand(self.num_envs, generator=self.g) * 2 - 1
        self.v = torch.zeros(self.num_envs)
        self.t = torch.zeros(self.num_envs, dtype=torch.int64)
        return self._obs()

    def step(self, a):
        self.x = self.x + 0.2 * a[:, 0].clamp(-1, 1)
        self.t += 1
        rew = -self.x.abs()
        self.raw_reward_mean = float(rew.mean())
        done = self.t >= self.max_episode_steps
        self.flags = torch.where(done, torch.tensor(FLAG_TRUNCATED, dtype=torch.uint8), torch.tensor(0, dtype=torch.uint8))
        self.final_obs = self._obs()
        if done.any():
            idx = done.nonzero(as_tuple=True)[0]
            self.x[idx] = torch.rand(len(idx), generator=self.g) * 2 - 1
            self.v[idx] = 0
            self.t[idx] = 0
        return self._obs(), rew, done, None


def test_gae_matches_naive_reference():
    torch.manual_seed(0)
    T, n, gamma, lam = 7, 5, 0.99, 0.95
    r, v = torch.randn(T, n), torch.randn(T, n)
    d = (torch.rand(T, n) < 0.2).float()
    last = torch.randn(n)
    adv, ret = compute_gae(r, v, d, last, gamma, lam)
    want = np.zeros((T, n))
    for e in range(n):
        for t in range(T):
            a, coef = 0.0, 1.0
            for k in range(t, T):
                nv = last[e].item() if k == T - 1 else v[k + 1, e].item()
                nt = 1.0 - d[k, e].item()
                a += coef * (r[k, e].item() + gamma * nv * nt - v[k, e].item())
                coef *= gamma * lam * nt
                if nt == 0.0:
                    break
            want[t, e] = a
    np.testing.assert_allclose(adv.numpy(), want, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ret.numpy(), want + v.numpy(), rtol=1e-5, atol=1e-6)


def test_policy_matches_torch_distributions_and_sb3_init():
    torch.manual_seed(1)
    pol = ActorCritic(13, 3, (64, 64))
    assert [m.out_features for m in pol.pi if isinstance(m, torch.nn.Linear)] == [64, 64, 3]
    assert float(pol.log_std.abs().max()) == 0.0                       # SB3 log_std_init = 0
    w = pol.pi[-1].weight
    assert abs(float((w @ w.T)[0, 0]) - 0.01 ** 2) < 1e-6              # orthogonal, gain 0.01
    with torch.no_grad():
        pol.log_std.copy_(torch.tensor([0.1, -0.3, 0.2]))
    obs = torch.randn(32, 13)
    a, v, logp = pol.act(obs)
    dist_ = torch.distributions.Normal(pol.pi(obs), pol.log_std.exp())
    np.testing.assert_allclose(logp.numpy(), dist_.log_prob(a).sum(-1).detach().numpy(), rtol=1e-5, atol=1e-5)
    v2, logp2, ent = pol.evaluate(obs, a)
    np.testing.assert_allclose(ent.detach().numpy(), dist_.entropy().sum(-1).detach().numpy(), rtol=1e-5)
    np.testing.assert_allclose(v2.detach().numpy(), v.numpy(), rtol=1e-6)


def test_truncation_bootstrap_and_learning_progress():
    env = ToyEnv(256, T=20)
    cfg = PPOConfig(n_steps=40, batch_size=2048, n_epochs=6, learning_rate=3e-3, seed=3)
    algo = PPO(env, cfg)
    algo.collect_rollouts()
    # every env is truncated at t = 20 and t = 40; the stored reward there is r + gamma * V(terminal obs)
    assert algo.buf_done[19].all() and algo.buf_done[39].all() and algo.buf_done[:19].sum() == 0
    x_after = algo.buf_obs[19, :, 0] + 0.2 * algo.buf_act[19, :, 0].clamp(-1, 1)
    term_obs = torch.zeros((env.num_envs, env.obs_dim))
    term_obs[:, 0] = x_after
    with torch.no_grad():
        want = -x_after.abs() + cfg.gamma * algo.policy.value(term_obs)
    np.testing.assert_allclose(algo.buf_rew[19].numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    first = env.raw_reward_mean
    for _ in range(20):
        algo.collect_rollouts()
        info = algo.train()
        assert all(math.isfinite(v) for v in info.values())
    assert env.raw_reward_mean > 0.25 * first, (first, env.raw_reward_mean)   # |x| shrinks >4x as the policy learns


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _ddp_worker(rank, world, port, q, counts=(64, 64), batch_size=256):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_ppo_host import ToyEnv

    env = ToyEnv(counts[rank], T=10, seed=100 + rank)          # different data per rank
    algo = PPO(env, PPOConfig(n_steps=20, batch_size=batch_size, n_epochs=2, seed=5))
    for _ in range(3):
        algo.collect_rollouts()
        algo.train()
    flat = torch.cat([p.detach().reshape(-1) for p in algo.policy.parameters()])
    q.put((rank, flat.numpy(), algo.num_timesteps))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_averaging_keeps_replicas_identical():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])     # same init + averaged grads => same weights
    assert res[0][2] == 3 * 20 * 64 * 2                     # timesteps count the whole job


def test_two_rank_ragged_shards_do_not_hang_and_count_the_whole_job():
    """dist.shard_range hands out counts that differ by one.  26 vs 25 envs x 20 steps = 520 vs 500 samples with
    minibatches of 260: rank 0 could run 2 steps per epoch, rank 1 only 1 -- the step count (= number of gradient
    all-reduces) must be agreed or the job hangs; gradients are weighted by sample share; timesteps are global."""
    from rl_rocket_6dof_amd.dist import shard_range

    counts = tuple(shard_range(51, r, 2)[1] for r in range(2))
    assert counts == (26, 25)
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q, counts, 260)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(res[0][1], res[1][1])
    assert res[0][2] == res[1][2] == 3 * 20 * 51


def test_ppo_refuses_a_time_limited_env_without_truncation_flags():
    env = ToyEnv(8, T=5)
    env.flags = None
    with pytest.raises(ValueError, match="record_flags"):
        PPO(env, PPOConfig(n_steps=4, batch_size=16))


def test_sb3_checkpoint_tensors_map_onto_both_topologies():
    """policy.policy_from_sb3_state_dict on the reference's own checkpoints (tensors in golden G8): the plain-list
    net_arch=[128, 64] checkpoint is ONE shared trunk (10 311 parameters), the default one two separate MLPs; the
    rebuilt modules reproduce the deterministic actions the reference env saw (logged every 25 steps)."""
    import numpy as np
    import torch

    from rl_rocket_6dof_amd.policy import policy_from_sb3_state_dict
    from tests.conftest import load_golden

    G = load_golden("g8_closed_loop.npz")
    for tag, shared, n_par in (("2bo71j9m", True, 10311), ("2i3ipkag", False, 10375)):
        sd = {k[len(tag) + 3:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(tag + "_w_")}
        pol = policy_from_sb3_state_dict(sd, device="cpu")
        assert pol.shared_trunk == shared
        assert sum(p.numel() for p in pol.parameters()) == n_par
        obs, act = G[f"{tag}_obs_every25"], G[f"{tag}_act_every25"]
        live = np.abs(obs).sum(axis=2) > 0
        with torch.no_grad():
            mean = pol.pi(torch.as_tensor(obs[live])).clamp(-1, 1).numpy()
        np.testing.assert_allclose(mean, act[live], rtol=0, atol=2e-6)


def test_load_sb3_policy_reads_a_checkpoint_zip_tensor_only(tmp_path):
    """policy.load_sb3_policy on a zip laid out like SB3's ``model.save`` (policy.pth inside), built here from the
    reference checkpoint tensors of golden G8; the loader uses torch.load(weights_only=True)."""
    import io
    import zipfile

    import numpy as np
    import torch

    from rl_rocket_6dof_amd.policy import load_sb3_policy
    from tests.conftest import load_golden

    G = load_golden("g8_closed_loop.npz")
    tag = "2bo71j9m"
    sd = {k[len(tag) + 3:]: torch.as_tensor(G[k]) for k in G.files if k.startswith(tag + "_w_")}
    buf = io.BytesIO()
    torch.save(sd, buf)
    path = tmp_path / "best_model.zip"
    with zipfile.ZipFile(path, "w") as z:
        z.writestr("policy.pth", buf.getvalue())
        z.writestr("data", "{}")
    pol = load_sb3_policy(str(path), device="cpu")
    assert pol.shared_trunk
    obs = G[f"{tag}_obs_every25"][:, 0]
    with torch.no_grad():
        got = pol.pi(torch.as_tensor(obs)).clamp(-1, 1).numpy()
    np.testing.assert_allclose(got, G[f"{tag}_act_every25"][:, 0], rtol=0, atol=2e-6)


def test_shipped_policy_lands_in_the_cpu_restatement_of_the_reference():
    """No GPU needed: the policy trained on the HIP env (examples/policies/) flown through oracle_np.OracleEnv --
    the numpy/scipy restatement of the reference's env (adaptive RK45 + terminal event + np.roots reward), from the
    initial conditions of golden G9 -- ends every episode on the step the REFERENCE ended it, landed."""
    import json
    import os

    import numpy as np
    import torch

    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd.policy import load_policy
    from tests.conftest import load_golden

    G = load_golden("g9_shipped_policy_in_reference.npz")
    cfg = json.loads(str(G["config"]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pol = load_policy(os.path.join(root, "examples", "policies", "landing_policy_shared_128_64.pt"), device="cpu")
    assert pol.shared_trunk
    for ep in range(4):
        env = onp.OracleEnv(cfg)
        obs = env.reset(G["ic"][ep])
        for k in range(1500):
            with torch.no_grad():
                a = pol.pi(torch.as_tensor(obs[:13], dtype=torch.float32)).clamp(-1, 1).numpy()
            obs, rew, done, info = env.step(a)
            if done:
                break
        assert k + 1 == int(G["n_steps"][ep])
        assert info["terms"][4] != 0.0, "goal_conditions: the episode must end landed"
        np.testing.assert_allclose(env.state[:6], G["terminal"][ep][:6], rtol=0, atol=2e-3)
