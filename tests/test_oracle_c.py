"""Pins the oracle's C restatement (oracle/oracle_c.c) against the golden vectors and against the
numpy oracle.  CPU-only."""
import json

import numpy as np
import pytest

from conftest import assert_close_scaled, load_golden, reward_rel_err, term_rel_err
from oracle import oracle_c as oc
from oracle import oracle_np as onp


# natural magnitude of the 7 reward terms (see tests/test_gpu_parity.py): the O(1..40) terms are float32
# quantities in the reference itself (float32 norms), so they cannot agree below ~1e-6 absolute
TERM_SCALE = np.array([1e-2, 1e-2, 1e-2, 1.0, 1.0, 1.0, 1.0])


@pytest.fixture(scope="module")
def cfg():
    return json.loads(str(load_golden("g3_env_rollouts.npz")["config"]))


@pytest.fixture(scope="module")
def P(cfg):
    return onp.OracleParams.from_config(cfg)


def test_g2_rhs(g2):
    scale = np.array([1, 1, 1, 1, 1, 1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 1.0])
    for i in range(0, len(g2["y"]), 3):
        got = oc.rhs(g2["y"][i], g2["u"][i], float(g2["m0"][i]))
        # glibc cosf/sinf vs numpy's float32 trig may differ by 1 ulp -> 1e-7 on the thrust direction
        assert_close_scaled(got, g2["dy"][i], scale, 3e-7, f"C rhs {i}")


@pytest.mark.parametrize("fixture", ["g3_env_rollouts.npz", "g4_ground_events.npz"])
def test_env_step_vs_reference(cfg, P, fixture):
    G = load_golden(fixture)
    n = len(G["reward"])
    env = oc.BatchOracle(n, cfg)
    env.set_state(G["pre"], m0=G["m0"])
    obs, rew, done = env.step(G["action"])
    np.testing.assert_array_equal(done, G["done"])
    np.testing.assert_array_equal((env.flags & 2) >> 1, G["bounds"])
    np.testing.assert_array_equal(env.flags & 1, (G["status"] == 1).astype(np.uint8))
    assert_close_scaled(env.state, G["post"], P.normalizer[None, :], 2e-6, "C post-state vs reference")
    assert_close_scaled(obs, G["obs"], 1.0, 2e-6, "C obs vs reference")
    assert_close_scaled(env.terms, G["terms"], TERM_SCALE, 2e-5, "C reward terms vs reference")
    assert_close_scaled(rew, G["reward"], 1e-1, 2e-5, "C reward vs reference")
    # the literal 1e-5: per term, and for the total against the largest of its own summands (a cancelled sum -- golden G4 row
    # 1179: +10.013 final_velocity against -10 attitude_constraint, reward -0.218 -- shows 1.3e-5 on |reward| alone)
    assert term_rel_err(env.terms, G["terms"], cfg["reward_coeff"], TERM_SCALE) < 1e-5
    assert reward_rel_err(rew, G["reward"], G["terms"]) < 1e-5


def test_g5_reward(cfg, P, g5):
    n = len(g5["state"])
    env = oc.BatchOracle(n, cfg, dt=1e-9)
    env.set_state(g5["state"].astype(np.float32).astype(np.float64))
    env.step(g5["action"])
    np.testing.assert_array_equal((env.flags & 2) >> 1, g5["bounds"])
    assert_close_scaled(env.terms, g5["terms"], TERM_SCALE, 5e-6, "C G5 reward terms")


def test_matches_numpy_oracle_incl_euler_and_velocity(cfg):
    G = load_golden("g4_ground_events.npz")
    idx = np.arange(0, len(G["reward"]), 11)
    for integ, shaping in (("euler", "acceleration"), ("rk4", "velocity")):
        c2 = json.loads(json.dumps(cfg))
        c2["reward_shaping_type"] = shaping
        c2["landing_params"]["waypoint"] = 40
        Pn = onp.OracleParams.from_config(c2)
        env = oc.BatchOracle(len(idx), c2, integrator=integ)
        v0 = np.linalg.norm(G["pre"][idx][:, 3:6], axis=1) + 5
        env.set_state(G["pre"][idx], m0=G["m0"][idx], v0=v0)
        env.step(G["action"][idx])
        for k, i in enumerate(idx):
            o = onp.env_step(G["pre"][i], float(G["m0"][i]), G["action"][i], Pn, integ, v0=float(v0[k]))
            assert bool(env.done[k]) == o["done"]
            assert_close_scaled(env.state[k], o["post"], Pn.normalizer, 2e-8, f"{integ} post {i}")   # glibc vs numpy float32 trig: 1 ulp
            assert_close_scaled(env.terms[k], o["terms"], TERM_SCALE, 5e-6, f"{integ}/{shaping} terms {i}")


def test_reset_stream_bit_exact_with_numpy_oracle(cfg, P):
    env = oc.BatchOracle(64, cfg, seed=99, env_offset=5_000_000_000)
    env.reset(step_index=(1 << 60) + 3)
    for i in range(0, 64, 7):
        want = onp.reset_sample(99, 5_000_000_000 + i, (1 << 60) + 3, P)
        plain = [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]
        np.testing.assert_array_equal(env.state[i, plain].astype(np.float32), want[plain])
        np.testing.assert_allclose(env.state[i, 6:10], want[6:10], atol=1.2e-7)


def test_reset_stream_against_the_references_own_draws(cfg):
    """Rocket6DOF.reset (rocket_env.py:180-199): the build's counter-based reset stream, as a sample, beside golden G12 -- 50 000
    initial conditions drawn by the reference's own reset()."""
    from conftest import check_reset_sample_against_reference_draws

    n = 1 << 17
    env = oc.BatchOracle(n, cfg, seed=7)
    env.reset(step_index=(1 << 60) + 1)
    w = check_reset_sample_against_reference_draws(env.state.astype(np.float32), "C oracle reset stream")
    print(f"[G12] worst KS distance / critical value: {w:.2f}")


def test_auto_reset_and_counters(cfg):
    n = 256
    env = oc.BatchOracle(n, cfg, auto_reset=True, track=True, clip_reward=(-1, 100), max_episode_steps=30)
    env.reset(0)
    rng = np.random.RandomState(0)
    total_done = 0
    for k in range(30):
        env.step(rng.uniform(-1, 1, (n, 3)).astype(np.float32))
        total_done += int(env.done.sum())
        assert np.all(env.reward <= 100) and np.all(env.reward >= -1)
    assert env.done.all() and ((env.flags & 4) != 0).sum() > n // 2     # TimeLimit at step 30
    assert np.all(env.ep_step == 0) and np.all(env.ep_return == 0)      # everything was just reset


def test_quartic_t_go_against_np_roots_on_wide_ranges(cfg, P):
    """rocket_env.py:528-546 picks the first positive real entry of np.roots.  The Newton solvers of the C
    oracle (and of the HIP kernel) must find that root also where the quartic has a positive minimum right of
    it (a complex pair with positive real part): from the nearly flat bottom of that minimum a Newton step
    jumps far to the left, which is the signal to restart on the other branch (a solver without that rule
    fails on 0.05 % of the second family below; a GPU-vs-C fuzz run found the first cases).  Families: wide
    uniform states, and fast approaches (v nearly anti-parallel to r, |v| = 2-25 % of |r| per second)."""
    rng = np.random.default_rng(5)
    n1, n2 = 6_000, 60_000
    n = n1 + n2
    s = np.zeros((n, 14))
    s[:, 6] = 1.0
    s[:, 13] = rng.uniform(25e3, 45e3, n)
    s[:n1, 0] = rng.uniform(0.0, 2600.0, n1)
    s[:n1, 1:3] = rng.uniform(-2600.0, 2600.0, (n1, 2))
    s[:n1, 3:6] = rng.uniform(-300.0, 300.0, (n1, 3))
    d = rng.normal(size=(n2, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rmag = rng.uniform(500.0, 4000.0, n2)
    s[n1:, 0:3] = d * rmag[:, None]
    s[n1:, 3:6] = -d * (rmag * rng.uniform(0.02, 0.25, n2))[:, None] + rng.normal(size=(n2, 3)) * 20.0
    s = s.astype(np.float32).astype(np.float64)
    a = np.zeros((n, 3), np.float32)
    env = oc.BatchOracle(n, cfg, dt=1e-9)
    env.set_state(s)
    env.step(a)
    u = onp.denormalize_action(a[0])
    worst, n_pair = 0.0, 0
    for i in range(n):
        s32 = s[i].astype(np.float32)
        want = onp.reward_terms(s32, u, s[i][6:10], P)[0]
        worst = max(worst, abs(env.terms[i, 0] - want) / max(abs(want), 1e-2))
        if i % 50 == 0:
            r, v = s32[0:3], s32[3:6]
            roots = np.roots([9.81 ** 2, 0.0, -4.0 * float(v @ v), -24.0 * float(r @ v), -36.0 * float(r @ r)])
            n_pair += int(any(z.imag > 0 and z.real > 0 for z in roots))
    assert n_pair > 100, "the sample must contain quartics with a complex pair right of the root"
    assert worst < 5e-6, f"tracking term differs from the np.roots restatement: {worst:.3e}"


@pytest.mark.parametrize("tag", ["wp50", "wp400"])
def test_g10_velocity_shaping_vs_reference(tag):
    """The C restatement with reward_shaping_type == 'velocity' against the reference's own run (golden G10)."""
    G = load_golden("g10_velocity_shaping.npz")
    vcfg = json.loads(str(G[f"config_{tag}"]))
    Pv = onp.OracleParams.from_config(vcfg)
    g = {k[len(tag) + 1:]: G[k] for k in G.files if k.startswith(tag + "_")}
    n = len(g["reward"])
    env = oc.BatchOracle(n, vcfg)
    env.set_state(g["pre"], m0=g["m0"], v0=g["v0"])
    obs, rew, done = env.step(g["action"])
    np.testing.assert_array_equal(done, g["done"])
    np.testing.assert_array_equal((env.flags & 2) >> 1, g["bounds"])
    np.testing.assert_array_equal(env.flags & 1, (g["status"] == 1).astype(np.uint8))
    assert_close_scaled(env.state, g["post"], Pv.normalizer[None, :], 2e-6, "C post-state vs reference (G10)")
    assert_close_scaled(env.terms, g["terms"], TERM_SCALE, 2e-5, "C reward terms vs reference (G10)")
    assert_close_scaled(rew, g["reward"], 1e-1, 2e-5, "C reward vs reference (G10)")


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g11_other_configs_vs_reference(tag):
    """The C restatement built from env_configs other than config.yaml against the reference's own runs (golden G11)."""
    from rl_rocket_6dof_amd.params import auto_substeps

    G = load_golden("g11_other_configs.npz")
    vcfg = json.loads(str(G[f"config_{tag}"]))
    Pv = onp.OracleParams.from_config(vcfg)
    g = {k[len(tag) + 1:]: G[k] for k in G.files if k.startswith(tag + "_")}
    n = len(g["reward"])
    env = oc.BatchOracle(n, vcfg, n_substeps=auto_substeps(vcfg["timestep"]))
    env.set_state(g["pre"], m0=g["m0"])
    obs, rew, done = env.step(g["action"])
    np.testing.assert_array_equal(done, g["done"])
    np.testing.assert_array_equal((env.flags & 2) >> 1, g["bounds"])
    np.testing.assert_array_equal(env.flags & 1, (g["status"] == 1).astype(np.uint8))
    assert_close_scaled(env.state, g["post"], Pv.normalizer[None, :], 2e-6, "C post-state vs reference (G11)")
    assert_close_scaled(obs, g["obs"], 1.0, 2e-6, "C obs vs reference (G11)")
    assert_close_scaled(env.terms, g["terms"], TERM_SCALE, 2e-5, "C reward terms vs reference (G11)")
    assert_close_scaled(rew, g["reward"], 1e-1, 2e-5, "C reward vs reference (G11)")
    assert term_rel_err(env.terms, g["terms"], vcfg["reward_coeff"], TERM_SCALE) < 1e-5
    assert reward_rel_err(rew, g["reward"], g["terms"]) < 1e-5
