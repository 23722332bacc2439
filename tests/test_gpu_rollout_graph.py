"""GPU: the launch-only rollout of ppo.PPO (fused forward -> r6_step into the rollout buffer -> masked time-limit bootstrap ->
r6_gae), its HIP-graph replay, and the pieces of the C ABI it stands on (R6Buffers.step_base, r6_policy_act's step_base /
row_flags, r6_gae).  The references are the per-step torch loop of the same class and SB3's formulas restated in torch."""
import pytest

pytestmark = pytest.mark.gpu


def _env(n, seed=7, **kw):
    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

    kw.setdefault("remove_mass", True)
    kw.setdefault("clip_reward", (-1, 100))
    return BatchedRocket6DOF(n, default_env_config(), seed=seed, **kw)


@pytest.mark.parametrize("time_limit", [False, True])
def test_r6_gae_matches_the_torch_recurrence(time_limit):
    import ctypes as C

    import torch

    from rl_rocket_6dof_amd import _lib
    from rl_rocket_6dof_amd.params import FLAG_TRUNCATED
    from rl_rocket_6dof_amd.ppo import compute_gae

    lib = _lib.load()
    torch.manual_seed(1)
    T, n, gamma, lam = 37, 5000, 0.99, 0.95
    dev = "cuda"
    rew, val = torch.randn(T, n, device=dev), torch.randn(T, n, device=dev)
    done = (torch.rand(T, n, device=dev) < 0.05).to(torch.uint8)
    last = torch.randn(n, device=dev)
    flags = (done * torch.randint(1, 8, (T, n), device=dev, dtype=torch.uint8)).contiguous()   # truncation bit only where done
    boot = torch.randn(T, n, device=dev)
    boot[(flags & FLAG_TRUNCATED) == 0] = float("nan")          # must never be read there
    adv, ret = torch.empty(T, n, device=dev), torch.empty(T, n, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.r6_gae(p(rew), p(val), p(done), p(last), T, n, gamma, lam, p(flags) if time_limit else None, FLAG_TRUNCATED,
                    p(boot) if time_limit else None, p(adv), p(ret), None)
    _lib.check(rc, "r6_gae", lib)
    r = rew.clone()
    if time_limit:
        tr = (flags & FLAG_TRUNCATED) != 0
        r[tr] += gamma * boot[tr]
    a_ref, r_ref = compute_gae(r, val, done.float(), last, gamma, lam)
    assert bool(torch.isfinite(adv).all())
    assert float((adv - a_ref).abs().max()) <= 2e-5 * float(a_ref.abs().max())
    assert float((ret - r_ref).abs().max()) <= 2e-5 * float(r_ref.abs().max())


def test_policy_step_base_and_row_mask():
    import torch

    from rl_rocket_6dof_amd.policy import FusedActorCritic
    from rl_rocket_6dof_amd.ppo import ActorCritic

    torch.manual_seed(0)
    n, od = 4096 + 17, 13
    pol = ActorCritic(od, 3, (128, 64), shared_trunk=True).cuda()
    f = FusedActorCritic(pol, seed=5)
    obs = torch.randn(n, od, device="cuda")
    a = f(obs, step_index=12)
    base = torch.tensor(7, dtype=torch.int64, device="cuda")
    b = f(obs, step_index=5, step_base=base)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                     # the same noise stream: index 5 + base 7
    base.add_(1)
    c = f(obs, step_index=5, step_base=base)
    assert not torch.equal(a[1], c[1])               # the base moved: fresh noise from the same recorded arguments
    # masked evaluation: flagged rows get exactly the unmasked result, tiles without a flagged row keep their old contents
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    rows = torch.tensor([3, 40, 41, 2000, n - 1], device="cuda")
    flags[rows] = 4
    flags[100] = 3                                    # other bits do not count
    full = f(obs, deterministic=True, step_index=0)
    out = tuple(torch.full_like(t, -77.0) for t in full)
    f(obs, deterministic=True, step_index=0, out=out, row_flags=flags, flag_mask=4)
    assert torch.equal(out[2][rows], full[2][rows]) and torch.equal(out[0][rows], full[0][rows])
    tiles = torch.zeros((n + 31) // 32, dtype=torch.bool, device="cuda")
    tiles[rows // 32] = True
    untouched = ~tiles.repeat_interleave(32)[:n]
    assert bool((out[2][untouched] == -77.0).all()) and int(untouched.sum()) > n - 5 * 32 - 1
    assert bool((out[2][~untouched] != -77.0).all())  # a selected tile is written whole


def test_env_step_base_and_step_into():
    import torch

    n = 2048
    a, b = _env(n, record_flags=True, max_episode_steps=6), _env(n, record_flags=True, max_episode_steps=6)
    base = torch.tensor(1, dtype=torch.int64, device="cuda")
    b.set_step_base(base)
    b._reset_count = -1                              # index (E - 1) + base 1 = E: the stream of a.reset()
    assert torch.equal(a.reset(), b.reset())
    b._step_count = -1                               # steps: (k - 1) + 1 = k
    g = torch.Generator().manual_seed(3)
    od = a.obs_dim
    obs = torch.empty((n, od), device="cuda")
    rew = torch.empty(n, device="cuda")
    done = torch.empty(n, dtype=torch.uint8, device="cuda")
    flags = torch.empty(n, dtype=torch.uint8, device="cuda")
    fobs = torch.zeros((n, od), device="cuda")
    truncated = 0
    for _ in range(14):                              # two time-limit truncations of every env: in-kernel resets keyed by step
        act = (torch.rand((n, 3), generator=g) * 2 - 1).cuda()
        o, r, d, _ = a.step(act)
        b.step_into(act, obs=obs, reward=rew, done=done, flags=flags, final_obs=fobs)
        assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(d.view(torch.uint8), done) and torch.equal(a.flags, flags)
        if bool(d.any()):
            sel = d.nonzero(as_tuple=True)[0]
            assert torch.equal(a.final_obs[sel], fobs[sel])
        truncated += int((flags & 4).ne(0).sum())
    assert truncated >= 2 * n - 64 and torch.equal(a.state, b.state)
    with pytest.raises(ValueError):
        b.step_into(act, obs=obs[:, :od - 1])
    with pytest.raises(ValueError):
        b.set_step_base(torch.zeros(1, device="cuda"))


@pytest.mark.parametrize("max_steps", [0, 9])
def test_graphed_rollout_is_the_eager_rollout(max_steps):
    """Same seeds, graph_rollout on / off: every rollout buffer of three consecutive rollouts must agree bit for bit -- the
    replay continues the RNG streams through the device-side base exactly where host counters would have been.  With a
    time limit of 9 steps every env is truncated during a 24-step rollout, so the masked bootstrap is exercised."""
    import torch

    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    runs = []
    for graph in (False, True):
        env = _env(4096, seed=11, record_flags=True, max_episode_steps=max_steps) if max_steps else _env(4096, seed=11)
        algo = PPO(env, PPOConfig(n_steps=24, batch_size=1 << 14, net_arch=(128, 64), seed=3, graph_rollout=graph))
        assert algo.fused is not None and algo._direct
        snaps = []
        for _ in range(3):
            algo.collect_rollouts()
            snaps.append([t.clone() for t in (algo.buf_obs, algo.buf_act, algo.buf_rew, algo.buf_done, algo.buf_val, algo.buf_logp,
                                              algo.adv, algo.ret, algo._obs, env.state)])
        assert (algo._rollout_graph is not None) == graph
        runs.append(snaps)
        stats = env.episode_stats()
    for k, (e, g) in enumerate(zip(*runs)):
        for i, (x, y) in enumerate(zip(e, g)):
            assert torch.equal(x, y), (k, i)
    assert not torch.equal(runs[0][1][1], runs[0][2][1])     # consecutive rollouts draw different noise
    if max_steps:
        assert stats["truncated"] > 0


def test_launch_only_rollout_matches_the_per_step_torch_loop():
    """The direct path (env outputs into the buffer, device-side bootstrap, r6_gae) against the loop it replaces (copies,
    a full V(final_obs) forward + addcmul per step, the torch GAE recurrence): same samples, advantages to fp32 rounding."""
    import torch

    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    res = []
    for direct in (True, False):
        env = _env(4096, seed=5, record_flags=True, max_episode_steps=7)
        algo = PPO(env, PPOConfig(n_steps=20, batch_size=1 << 14, net_arch=(128, 64), seed=3, graph_rollout=False))
        algo._direct = direct
        if not direct:
            env.set_step_base(None)
        algo.collect_rollouts()
        res.append([t.clone() for t in (algo.buf_obs, algo.buf_act, algo.buf_done, algo.buf_val, algo.adv, algo.ret)])
    for i in range(4):
        assert torch.equal(res[0][i], res[1][i]), i
    for i in (4, 5):
        assert float((res[0][i] - res[1][i]).abs().max()) <= 2e-5 * float(res[1][i].abs().max()), i


def _permutation_replica(n, seed, c):
    """numpy restatement of r6_permutation (keyed bijection of [0, 2^bits) + cycle walking): integer work, compared bit for bit."""
    import numpy as np

    M = (1 << 64) - 1

    def mix64(z):
        z &= M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    bits = (n - 1).bit_length() if n > 2 else 1
    mask = (1 << bits) - 1
    s1, s2 = (bits + 1) // 2, (bits // 3 if bits > 3 else 1)
    key = [mix64(seed + 0x9E3779B97F4A7C15 * (r + 1) + mix64(c + r * 0xD1342543DE82EF95)) for r in range(6)]
    x = np.arange(n, dtype=np.uint64)
    todo = np.ones(n, bool)
    while todo.any():
        y = x[todo]
        for r in range(6):
            y ^= np.uint64(key[r] & mask)
            y = (y * np.uint64((key[r] >> 17) | 1)) & np.uint64(mask)
            y ^= y >> np.uint64(s2 if r & 1 else s1)
        x[todo] = y
        todo = x >= n
    return x.astype(np.int64)


def test_r6_permutation_is_a_keyed_permutation():
    import numpy as np
    import torch

    from rl_rocket_6dof_amd import _lib

    lib = _lib.load()
    for n in (1, 2, 3, 7, 1000, 12345, 1 << 16, (1 << 20) - 3, 1 << 20):
        out = torch.full((n,), -1, dtype=torch.int64, device="cuda")
        _lib.check(lib.r6_permutation(out.data_ptr(), n, 42, 5, None, None), "r6_permutation", lib)
        p = out.cpu().numpy()
        assert np.array_equal(np.sort(p), np.arange(n)), n
        assert np.array_equal(p, _permutation_replica(n, 42, 5)), n
    n = 1 << 20
    base = torch.tensor(3, dtype=torch.int64, device="cuda")
    a, b, c = (torch.empty(n, dtype=torch.int64, device="cuda") for _ in range(3))
    lib.r6_permutation(a.data_ptr(), n, 9, 8, None, None)
    lib.r6_permutation(b.data_ptr(), n, 9, 5, base.data_ptr(), None)
    lib.r6_permutation(c.data_ptr(), n, 9, 6, base.data_ptr(), None)
    assert torch.equal(a, b) and int((a == c).sum()) < 32            # counter + base keys it; the next counter: unrelated order
    p = a.cpu().numpy().astype(np.float64)
    assert abs(np.corrcoef(np.arange(n), p)[0, 1]) < 5e-3 and abs(np.corrcoef(p[:-1], p[1:])[0, 1]) < 5e-3
    mb = p.reshape(16, -1).mean(1) / n                                # sixteen minibatches: each a fair sample of the rows
    assert np.abs(mb - 0.5).max() < 5e-3
