"""Full-size (BASELINE.json configs 2-3: N = 2^20) GPU checks through size-independent properties and a
sampled comparison with the C oracle.  Tolerances as in test_gpu_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N = 1 << 20


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch


@pytest.fixture(scope="module")
def cfg():
    from rl_rocket_6dof_amd import default_env_config

    return default_env_config()


def _rollout(torch, env, steps, seed=7):
    g = torch.Generator(device=env.device).manual_seed(seed)
    for _ in range(steps):
        env.step_async(torch.rand((env.num_envs, 3), generator=g, device=env.device) * 2 - 1)
    torch.cuda.synchronize()


@pytest.mark.parametrize("integ,dt", [("rk4", 0.1), ("euler", 0.04)])
def test_rollout_vs_c_oracle_sampled(torch_cuda, cfg, integ, dt):
    """260 auto-reset steps of 2^20 envs; at five steps 512 random envs are re-stepped by the C oracle
    from the kernel's own pre-state: post-state / obs / reward within 1e-5, done + flags bit-exact,
    reset state bit-exact with the oracle's Philox stream."""
    torch = torch_cuda
    from oracle import oracle_c as oc
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    P = onp.OracleParams.from_config(dict(cfg, timestep=dt))
    env = BatchedRocket6DOF(N, cfg, integrator=integ, timestep=dt, seed=42, record_flags=True, track_episodes=True)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(7)
    rng = np.random.RandomState(0)
    n_done = 0
    worst = 0.0
    m = 512
    for k in range(260):
        a = torch.rand((N, 3), generator=g, device=env.device) * 2 - 1
        check = k in (0, 60, 130, 200, 259)
        if check:
            idx = rng.choice(N, m, replace=False)
            sel = torch.as_tensor(idx, device=env.device)
            pre = env.get_state()[sel].cpu().numpy().astype(np.float64)
            m0 = env.epconst[0, sel].cpu().numpy().astype(np.float64)
        obs, rew, done, _ = env.step(a)
        n_done += int(done.sum().item())
        if check:
            ref = oc.BatchOracle(m, dict(cfg, timestep=dt), integrator=integ, dt=dt)
            ref.set_state(pre, m0=m0)
            ref.step(a[sel].cpu().numpy())
            d = done[sel].cpu().numpy()
            np.testing.assert_array_equal(d, ref.done.astype(bool))
            np.testing.assert_array_equal(env.flags[sel].cpu().numpy() & 3, ref.flags & 3)
            got_obs = np.where(d[:, None], env.final_obs[sel].cpu().numpy(), obs[sel].cpu().numpy())
            e = np.abs(got_obs - ref.obs) / np.maximum(np.abs(ref.obs), 1.0)
            worst = max(worst, float(e.max()))
            assert e.max() < 2e-6, f"obs vs C oracle: {e.max():.2e}"     # measured on MI355X: 2e-7 (rk4), events included up to 4e-7
            np.testing.assert_allclose(rew[sel].cpu().numpy(), ref.reward, rtol=1e-5, atol=1e-5)
            post = env.get_state()[sel].cpu().numpy()
            nd = ~d
            err = np.abs(post[nd] - ref.state[nd]) / np.maximum(np.abs(ref.state[nd]), P.normalizer)
            assert err.max() < 2e-6, f"post-state vs C oracle: {err.max():.2e}"
            for j in np.nonzero(d)[0][:8]:   # freshly reset envs: exactly the oracle's Philox sample
                want = onp.reset_sample(42, int(idx[j]), env._step_count, P)
                plain = [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]
                np.testing.assert_array_equal(post[j, plain], want[plain])
    st = env.get_state()
    assert bool(torch.isfinite(st).all())
    assert float((st[:, 6:10].norm(dim=1) - 1).abs().max()) < 1e-5
    stats = env.episode_stats()
    assert abs(stats["episodes"] - n_done) < 0.5 and n_done > 10000
    assert stats["ground"] + stats["bounds"] >= stats["episodes"]
    print(f"[fullsize {integ}] {n_done} episodes finished, worst scaled obs err vs C oracle {worst:.2e}, stats {stats}")


def test_determinism_and_sharding_invariance(torch_cuda, cfg):
    """Same seed => bit-identical rollouts; and two half-size shards with env_offset reproduce the
    single 2^20 batch bit for bit (the reset RNG is keyed by the global env index)."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    steps = 160   # long enough for the first wave of auto-resets
    full = BatchedRocket6DOF(N, cfg, seed=5)
    full.reset()
    again = BatchedRocket6DOF(N, cfg, seed=5)
    again.reset()
    lo = BatchedRocket6DOF(N // 2, cfg, seed=5, env_offset=0)
    hi = BatchedRocket6DOF(N // 2, cfg, seed=5, env_offset=N // 2)
    lo.reset()
    hi.reset()
    g = torch.Generator(device=full.device).manual_seed(3)
    for k in range(steps):
        a = (torch.rand((N, 3), generator=g, device=full.device) * 2 - 1)
        full.step_async(a)
        again.step_async(a)
        lo.step_async(a[: N // 2].contiguous())
        hi.step_async(a[N // 2:].contiguous())
    torch.cuda.synchronize()
    assert int(full.episode_stats(clear=False)["episodes"]) > 1000
    assert torch.equal(full.state[:, :N], again.state[:, :N]) and torch.equal(full.obs, again.obs)
    assert torch.equal(full.state[:, : N // 2], lo.state[:, : N // 2])
    assert torch.equal(full.state[:, N // 2: N], hi.state[:, : N // 2])
    assert torch.equal(full.obs[N // 2:], hi.obs) and torch.equal(full.reward[: N // 2], lo.reward)
    assert torch.equal(full.done[N // 2:], hi.done)


@pytest.mark.parametrize("n,groups,sync,wrapped", [(N, 2, "caller", False), (200_003, 3, "caller", True), (N // 2 + 77, 2, "none", True)])
def test_env_groups_on_their_own_streams_change_nothing(torch_cuda, cfg, n, groups, sync, wrapped):
    """BatchedRocket6DOF(groups=G): step_async issues G launches, one per contiguous env group, on streams the env owns (one
    group's ramp and tail under the other's bulk; round-4 review, Next 4).  Envs are independent and the reset RNG is keyed by the
    global env index, so every output, the state, the episode counters and the (exact) episode statistics equal the one-launch
    env's, bit for bit -- with the per-step fork / join against the caller's stream ("caller") and without ("none": the groups
    as independent pipelines, joined once at the end); odd sizes, the fused wrappers and the optional outputs included."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    kw = dict(seed=5, record_flags=True, record_terms=wrapped, track_episodes=True)
    if wrapped:
        kw.update(remove_mass=True, clip_reward=(-1, 100), max_episode_steps=40)
    one = BatchedRocket6DOF(n, cfg, **kw)
    grp = BatchedRocket6DOF(n, cfg, groups=groups, group_sync=sync, **kw)
    assert grp._group_bounds[0] == 0 and grp._group_bounds[-1] == n and all(b % 64 == 0 for b in grp._group_bounds[:-1])
    one.reset()
    grp.reset()
    g = torch.Generator(device=one.device).manual_seed(3)
    steps = 120
    acts = [(torch.rand((n, 3), generator=g, device=one.device) * 2 - 1) for _ in range(4)]
    torch.cuda.synchronize()      # "none": the action pool is complete before any group stream reads it
    for k in range(steps):
        a = acts[k % 4]
        one.step_async(a)
        grp.step_async(a)
        if sync == "caller":
            o1, r1, d1, _ = one.step_wait()
            o2, r2, d2, _ = grp.step_wait()
            if k % 40 == 7:       # ordered against the caller's stream: readable right here
                assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2)
    grp.join_groups()
    torch.cuda.synchronize()
    assert int(one.episode_stats(clear=False)["episodes"]) > (1000 if wrapped else 200)
    assert one.episode_stats(clear=False) == grp.episode_stats(clear=False)
    for name in ("state", "epconst", "obs", "reward", "done", "flags", "final_obs", "ep_step", "ep_return", "fin_return", "fin_length") + (("terms",) if wrapped else ()):
        assert torch.equal(getattr(one, name), getattr(grp, name)), name


@pytest.mark.parametrize("n", [3 << 21, 1 << 21])
def test_streaming_store_form_matches_the_plain_form(torch_cuda, cfg, n):
    """From 3 * 2^21 envs on, and (round 4) between 1.2 M and 2.36 M envs, the launcher picks the kernel form whose obs / reward /
    done stores carry the non-temporal hint (r6_kernels.hip, stream_store).  One batch of such a size against its two half-size
    shards (plain form: 3 * 2^20 resp. 2^20 envs; global env indices via env_offset): every output bit for bit, resets included."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    h = n // 2
    for integ, dt in (("rk4", 0.1), ("euler", 0.04)):
        big = BatchedRocket6DOF(n, cfg, integrator=integ, timestep=dt, seed=9)
        lo = BatchedRocket6DOF(h, cfg, integrator=integ, timestep=dt, seed=9, env_offset=0)
        hi = BatchedRocket6DOF(h, cfg, integrator=integ, timestep=dt, seed=9, env_offset=h)
        for e in (big, lo, hi):
            e.reset()
        g = torch.Generator(device=big.device).manual_seed(4)
        steps = 150 if integ == "rk4" else 60
        for k in range(steps):
            a = torch.rand((n, 3), generator=g, device=big.device) * 2 - 1
            big.step_async(a)
            lo.step_async(a[:h].contiguous())
            hi.step_async(a[h:].contiguous())
        torch.cuda.synchronize()
        if integ == "rk4":
            assert int(big.episode_stats(clear=False)["episodes"]) > 1000
        assert torch.equal(big.state[:, :h], lo.state[:, :h]) and torch.equal(big.state[:, h:n], hi.state[:, :h])
        assert torch.equal(big.obs[:h], lo.obs) and torch.equal(big.obs[h:], hi.obs)
        assert torch.equal(big.reward[:h], lo.reward) and torch.equal(big.reward[h:], hi.reward)
        assert torch.equal(big.done[:h], lo.done) and torch.equal(big.done[h:], hi.done)
        del big, lo, hi


def test_physical_invariants_over_a_rollout(torch_cuda, cfg):
    """Mass falls by exactly thrust*dt/(g0*Isp) per step (closed form, simulator.py:140-141), the roll
    rate omega_x is conserved (tau_x == 0 and Jy == Jz), observations are state/normaliser."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n = 1 << 18
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, seed=11)
    env.reset()
    g = torch.Generator(device=env.device).manual_seed(1)
    for k in range(30):
        a = torch.rand((n, 3), generator=g, device=env.device) * 2 - 1
        s0 = env.get_state()
        obs, rew, done, _ = env.step(a)
        s1 = env.get_state()
        alive = ~done
        thrust = (a[:, 2] + 1) * 0.5 * 981e3
        dm = thrust * 0.1 / (9.81 * 360.0)
        assert float(((s0[:, 13] - s1[:, 13]) - dm)[alive].abs().max()) < 2e-2          # fp32 ulp of 4e4 kg is 4e-3
        assert float((s1[:, 10] - s0[:, 10])[alive].abs().max()) == 0.0
        inv = torch.as_tensor(1.0 / env.state_normalizer, dtype=torch.float32, device=env.device)
        assert float((obs - s1 * inv).abs().max()) < 1e-6


def test_mass_termination_overflows_the_reset_queue(torch_cuda, cfg):
    """Every env of a 4 M batch ends its episode in the same step (all placed outside the position bounds):
    ~3300 finished episodes per workgroup, more than its LDS reset queue holds, so the overflow path of the
    tile loop re-samples the excess.  Every env must come out re-sampled exactly as the oracle's Philox
    stream prescribes for (seed, step_index, global env id), with the terminal observation preserved."""
    torch = torch_cuda
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n = 1 << 22
    env = BatchedRocket6DOF(n, cfg, seed=99, env_offset=12345, record_flags=True, track_episodes=True)
    env.reset()
    st = env.get_state()
    st[:, 1] = 1.0e6                       # far outside the y bounds -> done on the next step
    env.set_state(st)
    before = env.get_state()
    obs, rew, done, _ = env.step(torch.zeros((n, 3), device=env.device))
    torch.cuda.synchronize()
    assert bool(done.all())
    assert bool(((env.flags & 2) != 0).all())
    assert int(env.episode_stats(clear=False)["episodes"]) == n
    after = env.get_state()
    assert bool((after[:, 1].abs() < 1.0e5).all()), "an env was left in its terminal state"
    assert bool((env.ep_step == 0).all())
    assert bool((env.final_obs[:, 1] > 100.0).all()), "terminal observation must be the pre-reset one"
    assert bool((env.fin_length == 1).all())
    P = onp.OracleParams.from_config(cfg)
    a = after.cpu().numpy()
    plain = [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]
    rng = np.random.default_rng(0)
    for i in np.concatenate([rng.integers(0, n, 400), [0, 63, 64, n - 1]]):
        want = onp.reset_sample(99, 12345 + int(i), env._step_count, P)
        np.testing.assert_array_equal(a[i, plain], want[plain])
        np.testing.assert_allclose(a[i, 6:10], want[6:10], rtol=0, atol=1.2e-7)
    np.testing.assert_allclose(obs.cpu().numpy()[::4097], a[::4097] / P.normalizer, rtol=3e-7, atol=1e-9)
    assert not torch.equal(before, after)


@pytest.mark.parametrize("integ,shaping", [("rk4", "acceleration"), ("euler", "velocity")])
def test_fuzz_wide_state_ranges_vs_c_oracle(torch_cuda, cfg, integ, shaping):
    """One step from 200 000 states drawn far outside the distribution of the golden rollouts (whole bounds box
    and beyond, |v| up to 300 m/s, arbitrary attitudes, |omega| up to 1 rad/s, 25-45 t, actions in [-1.5, 1.5]
    -- the env does not clip) against the C oracle: post-state / obs within 1e-5 of the normaliser scale, done
    and flag bytes equal, reward within 1e-5 -- except for the few draws that sit on a discrete threshold
    (a float32 vs float64 straddle flips a whole reward term there), which are counted and bounded."""
    torch = torch_cuda
    import json

    from oracle import oracle_c as oc
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    c2 = json.loads(json.dumps(cfg))
    c2["reward_shaping_type"] = shaping
    if shaping == "velocity":
        c2["landing_params"]["waypoint"] = 40
    n = 200_000
    rng = np.random.default_rng(11)
    P = onp.OracleParams.from_config(c2)
    s = np.zeros((n, 14))
    s[:, 0] = rng.uniform(-50.0, 2600.0, n)
    s[: n // 4, 0] = rng.uniform(0.0, 12.0, n // 4)            # a quarter close to the ground: events, landing terms
    s[:, 1:3] = rng.uniform(-2600.0, 2600.0, (n, 2))
    s[: n // 4, 1:3] = rng.uniform(-60.0, 60.0, (n // 4, 2))
    s[:, 3:6] = rng.uniform(-300.0, 300.0, (n, 3))
    s[: n // 4, 3:6] = rng.uniform(-25.0, 25.0, (n // 4, 3))
    q = rng.normal(size=(n, 4))
    s[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
    s[:, 10:13] = rng.uniform(-1.0, 1.0, (n, 3))
    s[:, 13] = rng.uniform(25e3, 45e3, n)
    s = s.astype(np.float32).astype(np.float64)
    m0 = (s[:, 13] + rng.uniform(0.0, 3e3, n)).astype(np.float32).astype(np.float64)
    a = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)

    env = BatchedRocket6DOF(n, c2, integrator=integ, auto_reset=False, record_flags=True, seed=1)
    env.set_state(torch.as_tensor(s, dtype=torch.float32), m0=torch.as_tensor(m0, dtype=torch.float32))
    obs, rew, done, _ = env.step(torch.as_tensor(a))
    ref = oc.BatchOracle(n, c2, integrator=integ)
    ref.set_state(s, m0=m0)
    ref.step(a)

    post = env.get_state().cpu().numpy().astype(np.float64)
    assert np.isfinite(post).all()
    err = np.abs(post - ref.state) / np.maximum(np.abs(ref.state), P.normalizer)
    assert err.max() < 1e-5, f"post-state: worst scaled error {err.max():.3e} at {np.unravel_index(err.argmax(), err.shape)}"
    eo = np.abs(obs.cpu().numpy() - ref.obs) / np.maximum(np.abs(ref.obs), 1.0)
    assert eo.max() < 1e-5
    d_bad = int((done.cpu().numpy() != ref.done.astype(bool)).sum())
    f_bad = int(((env.flags.cpu().numpy() & 3) != (ref.flags & 3)).sum())
    r = rew.cpu().numpy().astype(np.float64)
    r_bad = int((np.abs(r - ref.reward) > 1e-5 * np.maximum(np.abs(ref.reward), 1.0)).sum())
    print(f"[fuzz {integ}/{shaping}] state err {err.max():.2e}, obs err {eo.max():.2e}, mismatches: done {d_bad}, flags {f_bad}, "
          f"reward {r_bad} of {n}; ground events {int((ref.flags & 1).sum())}, bounds {int(((ref.flags >> 1) & 1).sum())}")
    assert d_bad <= 2 and f_bad <= 2, "done / flag bytes differ on more than a threshold straddle"
    assert r_bad <= n // 5000, "reward differs beyond tolerance on more envs than threshold straddles explain"



def test_reset_against_the_references_own_draws():
    """a23, Rocket6DOF.reset (rocket_env.py:180-199, init_space :78-81): 2^20 HIP resets, as a sample, beside golden G12 -- 50 000
    initial conditions drawn by the reference's own reset() in the golden generator (two-sample KS per component, means, spreads,
    constant components, quaternion rule)."""
    from conftest import check_reset_sample_against_reference_draws
    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

    env = BatchedRocket6DOF(1 << 20, default_env_config(), seed=321)
    env.reset()
    w = check_reset_sample_against_reference_draws(env.get_state().cpu().numpy(), "HIP reset")
    print(f"[G12] HIP reset vs the reference's draws: worst KS distance / critical value {w:.2f}")
    # in-kernel auto-reset draws (another counter range of the same stream): the states of envs that have just been re-sampled
    env2 = BatchedRocket6DOF(1 << 18, default_env_config(), seed=5, max_episode_steps=1)     # every env ends after its first step
    env2.reset()
    env2.step(torch_zero_actions(env2))
    w2 = check_reset_sample_against_reference_draws(env2.get_state().cpu().numpy(), "HIP in-kernel reset")
    print(f"[G12] HIP in-kernel reset vs the reference's draws: {w2:.2f}")


def torch_zero_actions(env):
    import torch

    return torch.zeros((env.num_envs, 3), device=env.device)


def test_reset_distribution_matches_the_reference_rule():
    """Rocket6DOF.reset (rocket_env.py:180-199, init_space :78-81): every component ~ U(IC - ICRange/2, IC + ICRange/2)
    independently, quaternion normalised afterwards.  gym's RNG stream cannot be reproduced (reset
    parity is distributional), so the 2^20 draws of one reset are tested as a distribution: range, mean, variance and the
    Kolmogorov-Smirnov distance per component, pairwise correlations, and the quaternion rule."""
    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config

    cfg = default_env_config()
    n = 1 << 20
    env = BatchedRocket6DOF(n, cfg, seed=123)
    env.reset()
    st = env.get_state().double().cpu().numpy()
    ic, rng = np.float64(cfg["IC"]), np.float64(cfg["ICRange"])
    plain = [0, 1, 3, 4, 10, 11, 12, 13]                       # components with a non-zero range that are not normalised afterwards
    for j in range(14):
        if rng[j] == 0 and j not in (6, 7, 8, 9):
            assert np.all(st[:, j] == np.float32(ic[j])), f"component {j} has no range: every env gets IC[{j}]"
    u = (st[:, plain] - (ic[plain] - rng[plain] / 2)) / rng[plain]
    assert u.min() >= -1e-4 and u.max() <= 1.0 + 1e-4      # float32 states: x around 2000 m resolves 1.2e-4 m of a 10 m range
    se = 1.0 / np.sqrt(12 * n)
    assert np.all(np.abs(u.mean(0) - 0.5) < 5 * se), u.mean(0)
    assert np.all(np.abs(u.var(0) - 1 / 12) < 5 * np.sqrt(1 / 180 / n)), u.var(0)
    grid = (np.arange(n) + 0.5) / n
    for c in range(u.shape[1]):
        ks = np.abs(np.sort(u[:, c]) - grid).max()
        assert ks < 1.95 / np.sqrt(n), (plain[c], ks)           # KS critical value at alpha = 0.001
    corr = np.corrcoef(u.T)
    assert np.abs(corr - np.eye(len(plain))).max() < 5 / np.sqrt(n)
    q = st[:, 6:10]
    assert np.abs(np.linalg.norm(q, axis=1) - 1).max() < 1e-6   # rocket_env.py:194: normalised
    # before normalisation the quaternion components were uniform around IC[6:10]: the direction statistics follow
    raw = np.random.RandomState(0).uniform(ic[6:10] - rng[6:10] / 2, ic[6:10] + rng[6:10] / 2, (n, 4))
    raw /= np.linalg.norm(raw, axis=1, keepdims=True)
    assert np.abs(q.mean(0) - raw.mean(0)).max() < 5 * raw.std(0).max() / np.sqrt(n) * 2
    assert np.abs(q.std(0) - raw.std(0)).max() < 2e-4
    # m0 / v0 of the fresh episodes (simulator.py:42-50, rocket_env.py:649-651)
    assert torch.equal(env.epconst[0, :n], env.state[13, :n])
    np.testing.assert_allclose(env.epconst[1, :n].cpu().numpy(), np.linalg.norm(st[:, 3:6], axis=1), rtol=2e-7)


@pytest.mark.parametrize("integ,dt", [("rk4", 0.1), ("euler", 0.04), ("rk45", 0.1)])
def test_rows_beyond_4_gib_match_small_shards(torch_cuda, cfg, integ, dt):
    """Maximum sizes: 2^27 + 77 envs on one GPU (128x the headline batch; 7.5 GB of SoA state whose rows start up to 7 GiB past the
    base pointer, AoS observation / action blocks whose byte offsets pass 2^31 and 2^32) through the size-independent property the
    domain offers: envs are independent and the reset RNG is keyed by the GLOBAL env index, so any window of the big batch equals,
    bit for bit, a small env built with `env_offset` = the window's first env and stepped with the window's actions.  TimeLimit 6
    puts every env through the in-kernel reset (queue entries, final_obs, episode sums at those offsets) within the run."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n, m, steps = (1 << 27) + 77, 4096, 14
    kw = dict(seed=11, integrator=integ, timestep=dt, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=6, record_flags=True)
    big = BatchedRocket6DOF(n, cfg, **kw)
    dev = big.device
    # windows: the first tile, the envs whose observation rows straddle byte offsets 2^31 and 2^32, the ragged last tile
    starts = [0, ((1 << 31) // 52 - m // 2) // 64 * 64, ((1 << 32) // 52 - m // 2) // 64 * 64, n - 77 - (m - 128)]
    small = [BatchedRocket6DOF(min(m, n - s), cfg, env_offset=s, **kw) for s in starts]
    big.reset()
    for e in small:
        e.reset()
    g = torch.Generator(device=dev).manual_seed(9)
    acts = [(torch.rand((n, 3), generator=g, device=dev) * 2 - 1) for _ in range(2)]
    for k in range(steps):
        a = acts[k & 1]
        big.step_async(a)
        for s, e in zip(starts, small):
            e.step_async(a[s: s + e.num_envs].contiguous())
        if k in (5, 6, steps - 1):          # the TimeLimit step, the step behind the resets, the last one
            torch.cuda.synchronize()
            for s, e in zip(starts, small):
                w = e.num_envs
                assert torch.equal(big.state[:, s: s + w], e.state[:, :w]), (k, s)
                for name in ("obs", "reward", "done", "flags", "final_obs", "ep_step", "ep_return", "fin_return", "fin_length"):
                    assert torch.equal(getattr(big, name)[s: s + w], getattr(e, name)), (name, k, s)
            if k == 5:      # (nearly) every env truncated and re-sampled in this step; the few that ended earlier are mid-episode
                assert int(big.done.sum().item()) > 0.999 * n
    # r6_step_k at this size: K = 3 steps in one launch (last step's outputs in the env's buffers), the windows again
    ak = torch.stack([acts[0], acts[1], acts[0]])
    big.step_k(ak, last_only=True)
    for s, e in zip(starts, small):
        e.step_k(ak[:, s: s + e.num_envs].contiguous(), last_only=True)
    torch.cuda.synchronize()
    for s, e in zip(starts, small):
        w = e.num_envs
        assert torch.equal(big.state[:, s: s + w], e.state[:, :w]), ("step_k", s)
        for name in ("obs", "reward", "done", "flags", "ep_step", "ep_return"):
            assert torch.equal(getattr(big, name)[s: s + w], getattr(e, name)), ("step_k", name, s)
    st = big.episode_stats()
    assert 1.999 * n <= st["episodes"] <= 2.001 * n and st["truncated"] > 1.99 * n
    del ak
    del big, small, acts
    torch.cuda.empty_cache()
