"""GPU parity tests proper: the HIP path (through the C ABI / BatchedRocket6DOF) against
  (1) the golden vectors recorded from the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle (oracle/oracle_np.py) on seeded inputs.

Tolerance (BASELINE.json north_star): per-step outputs within 1e-5 relative fp32, done masks
bit-exact.  "Relative" is taken against max(|reference|, natural scale) where the natural scale of a
state component is the reference's own observation normaliser (so the test is a 1e-5 bound on the
normalised observation error); reward-term scales are given at TERM_SCALE; the total reward's is 1.
"""
import json
import os

import numpy as np
import pytest

from conftest import PARITY_LOG, assert_close_scaled, load_golden, pure_rel_err, reward_rel_err, term_rel_err

pytestmark = pytest.mark.gpu

RTOL = 1e-5     # the north_star bound; the assertions below sit at ~5x the worst error MEASURED on MI355X (round 2,
                # gpurun_out/parity_errors.json, printed by every test session) wherever that is tighter
# (post-state, observation, reward terms, reward) by kind of fixture; measured worst in the comments
TOL_STEPS = (1.3e-6, 1.3e-6, 5e-6, 1e-6)      # G3 flights without events: 2.6e-7, 2.3e-7, 8.4e-7, 1.2e-7
TOL_EVENTS = (5e-6, 5e-6, 1e-5, 1e-5)         # G4 / G10 with ground events: 1.0e-6, 1.0e-6, 2.9e-6, 3.2e-6
# natural magnitude of the 7 reward terms with config.yaml coefficients: tracking ~0.1, thrust
# penalty <= 0.01, eta = 0.1 per step; attitude (gamma = -10), goal (kappa = 10), final position /
# velocity (up to max_v_f * w_v_f = 37.5) are O(1..10) quantities whose fp32 resolution is ~1e-6.
TERM_SCALE = np.array([1e-1, 1e-1, 1e-1, 1.0, 1.0, 1.0, 1.0])
REWARD_SCALE = 1.0


@pytest.fixture(scope="module")
def torch_cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test running without a GPU")
    return torch


@pytest.fixture(scope="module")
def cfg():
    return json.loads(str(load_golden("g3_env_rollouts.npz")["config"]))


@pytest.fixture(scope="module")
def P(cfg):
    from oracle import oracle_np as onp

    return onp.OracleParams.from_config(cfg)


def _teacher_forced(torch, cfg, G, **kw):
    """Every recorded reference step becomes one env of a single launch."""
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n = len(G["reward"])
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True, **kw)
    env.set_state(G["pre"].astype(np.float32), m0=G["m0"].astype(np.float32))
    obs, rew, done, infos = env.step(torch.as_tensor(G["action"], device=env.device))
    torch.cuda.synchronize()
    return env, obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), infos


def _check_against_golden(env, obs, rew, done, G, P, what, tol=TOL_EVENTS):
    flags = env.flags.cpu().numpy()
    np.testing.assert_array_equal(done.astype(np.uint8), G["done"], err_msg=f"{what}: done mask")
    np.testing.assert_array_equal((flags & 2) >> 1, G["bounds"], err_msg=f"{what}: bounds flag")
    np.testing.assert_array_equal(flags & 1, (G["status"] == 1).astype(np.uint8), err_msg=f"{what}: ground-event status")
    post = env.get_state().cpu().numpy()
    e_state = assert_close_scaled(post, G["post"], P.normalizer[None, :], tol[0], f"{what}: post-state")
    e_obs = assert_close_scaled(obs, G["obs"], 1.0, tol[1], f"{what}: observation")
    # "1e-5 relative" read literally, on every component that is not tiny against its natural scale
    pr_state, pr_obs = pure_rel_err(post, G["post"], P.normalizer[None, :]), pure_rel_err(obs, G["obs"], 1.0)
    assert pr_state < RTOL and pr_obs < RTOL, f"{what}: pure relative error state {pr_state:.2e} obs {pr_obs:.2e}"
    terms = env.terms[:7, : env.num_envs].t().cpu().numpy()
    e_terms = assert_close_scaled(terms, G["terms"], TERM_SCALE, tol[2], f"{what}: reward terms")
    e_rew = assert_close_scaled(rew, G["reward"], REWARD_SCALE, tol[3], f"{what}: reward")
    # the literal bound for the reward side as well (round 3): every reward TERM that is not tiny against its natural scale is
    # within 1e-5 pure relative of the reference's, and so is the total against the largest of its own summands -- where
    # |reward| / |reward_ref| - 1 alone exceeds 1e-5 (golden G4 row 1179: 1.3e-5 even for the float64 oracle) the row is a
    # cancelled sum (final_velocity +10.013 against attitude_constraint -10 ...: reward -0.218) whose terms agree to 3e-7
    pr_terms = term_rel_err(terms, G["terms"], env.reward_coefficients, TERM_SCALE)
    pr_rew = reward_rel_err(rew, G["reward"], G["terms"][:, :7])
    rec = PARITY_LOG.setdefault(f"{what}: reward, PURE relative vs max(|reward|, largest |term|) @rtol {RTOL:g}",
                                {"scaled": 0.0, "pure_rel_unasserted": 0.0, "rtol": RTOL, "n": 0})      # here the asserted quantity IS the pure relative one
    rec["scaled"], rec["n"] = max(rec["scaled"], pr_rew), rec["n"] + len(rew)
    assert pr_terms < RTOL and pr_rew < RTOL, f"{what}: pure relative error reward terms {pr_terms:.2e} reward (vs its summands) {pr_rew:.2e}"
    print(f"[{what}] n={len(rew)} worst scaled err: state {e_state:.2e} obs {e_obs:.2e} terms {e_terms:.2e} reward {e_rew:.2e}; "
          f"pure relative: state {pr_state:.2e} obs {pr_obs:.2e} terms {pr_terms:.2e} reward vs summands {pr_rew:.2e}")
    return e_state, e_obs, e_terms, e_rew


def test_g2_rhs_known_answers(torch_cuda):
    """Simulator6DOF.RHS (simulator.py:106-143) on 512 random (y,u), un-normalised quaternions."""
    import ctypes as C

    from rl_rocket_6dof_amd import _lib
    from rl_rocket_6dof_amd.params import R6Params

    torch = torch_cuda
    g2 = load_golden("g2_rhs.npz")
    n = len(g2["y"])
    ld = (n + 63) // 64 * 64
    dev = "cuda:0"
    y = torch.zeros((14, ld), device=dev)
    y[:, :n] = torch.as_tensor(g2["y"].T.astype(np.float32))
    ep = torch.zeros((2, ld), device=dev)
    ep[0, :n] = torch.as_tensor(g2["m0"].astype(np.float32))
    u = torch.as_tensor(g2["u"], device=dev).contiguous()
    dy = torch.zeros((14, ld), device=dev)
    lib = _lib.load()
    p = R6Params()
    rc = lib.r6_rhs(C.byref(p), y.data_ptr(), ep.data_ptr(), u.data_ptr(), dy.data_ptr(), n, ld, None)
    _lib.check(rc)
    torch.cuda.synchronize()
    got = dy[:, :n].t().cpu().numpy()
    # derivative scales: velocities ~1e2 m/s, accelerations ~1e1, qdot ~1, wdot ~1, mdot ~1e2
    scale = np.array([100, 100, 100, 10, 10, 10, 1, 1, 1, 1, 1, 1, 1, 100.0])
    err = assert_close_scaled(got, g2["dy"], scale[None, :], 3.5e-6, "RHS vs golden G2")     # measured 6.5e-7
    print(f"[G2] worst scaled err {err:.2e}")


def test_g1_simulator_smoke_case(torch_cuda):
    """test_6DOF_simulator.py:3-7 through the Simulator6DOF drop-in (dt = 0.5 -> 5 RK4 sub-steps)."""
    from rl_rocket_6dof_amd import Simulator6DOF

    g1 = load_golden("g1_sim_step.npz")
    sim = Simulator6DOF(list(g1["ic"]))          # default dt 0.5 like the reference
    state, status = sim.step(list(g1["u"]))
    assert status == int(g1["status"]) == 0
    scale = np.array([100, 100, 100, 5, 5, 5, 1, 1, 1, 1, 1e-2, 1e-2, 1e-2, 5e4])
    assert_close_scaled(state, g1["state"], scale, 1.1e-6, "G1 Simulator6DOF.step")     # measured 2.1e-7
    assert sim.t == 0.5 and len(sim.states) == 2 and len(sim.times) == 2


@pytest.mark.parametrize("fixture", ["g3_env_rollouts.npz", "g4_ground_events.npz"])
def test_env_step_teacher_forced_vs_reference(torch_cuda, cfg, P, fixture):
    """Rocket6DOF.step (rocket_env.py:201-231), every recorded reference step restarted from the
    reference's own pre-state: post-state, obs, 7 reward terms, reward within 1e-5; done, bounds and
    ground-event masks bit-exact."""
    G = load_golden(fixture)
    env, obs, rew, done, _ = _teacher_forced(torch_cuda, cfg, G)
    _check_against_golden(env, obs, rew, done, G, P, fixture, tol=TOL_STEPS if fixture.startswith("g3") else TOL_EVENTS)


def test_g5_reward_known_answers(torch_cuda, cfg, P):
    """_compute_reward / _reward_goal / _check_attitude_limits / _check_bounds_violation
    (rocket_env.py:317-402, :591-593) on crafted post-states.  The kernel only evaluates the reward
    after a step, so the step is made degenerate (dt = 1e-9 s: the state does not move in fp32)."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    g5 = load_golden("g5_reward_cases.npz")
    S = g5["state"].astype(np.float32)
    n = len(S)
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True, timestep=1e-9, n_substeps=1)
    env.set_state(S)
    obs, rew, done, _ = env.step(torch.as_tensor(g5["action"], device=env.device))
    torch.cuda.synchronize()
    flags = env.flags.cpu().numpy()
    np.testing.assert_array_equal((flags & 2) >> 1, g5["bounds"], err_msg="bounds flag")
    terms = env.terms[:7, :n].t().cpu().numpy()
    # discrete terms (attitude, goal) must agree exactly in which cases fire
    np.testing.assert_array_equal(terms[:, 3] != 0, g5["terms"][:, 3] != 0, err_msg="attitude_constraint cases")
    np.testing.assert_array_equal(terms[:, 4] != 0, g5["terms"][:, 4] != 0, err_msg="goal_conditions cases")
    np.testing.assert_array_equal(terms[:, 6] != 0, g5["terms"][:, 6] != 0, err_msg="final_velocity cases")
    e = assert_close_scaled(terms, g5["terms"], TERM_SCALE, RTOL, "G5 reward terms")
    want = g5["reward_sum"] + np.where(g5["bounds"] == 1, -50.0, 0.0)
    assert_close_scaled(rew.cpu().numpy(), want, REWARD_SCALE, RTOL, "G5 reward")
    print(f"[G5] n={n} worst scaled term err {e:.2e}")


def test_g3_free_running_episodes(torch_cuda, cfg, P):
    """Whole reference episodes replayed open-loop in fp32 from the recorded x0 with the recorded
    actions: the episode must end on the same step for the same reason.  Trajectory drift (fp32
    accumulation over 100-250 steps vs the reference's float64 free run) is reported and bounded at
    1e-4 of each component's range (the observation normaliser; for z, whose normaliser is the
    degenerate 1 m, the +-180 m bound)."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g3_env_rollouts.npz")
    eps = np.unique(G["episode"])
    n = len(eps)
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=False)
    env.set_state(G["x0"].astype(np.float32))
    lens = np.array([(G["episode"] == e).sum() for e in eps])
    starts = np.array([np.nonzero(G["episode"] == e)[0][0] for e in eps])
    scale = P.normalizer.copy()
    scale[2] = 180.0
    worst = 0.0
    for k in range(lens.max()):
        act = np.zeros((n, 3), np.float32)
        alive = k < lens
        act[alive] = G["action"][starts[alive] + k]
        obs, rew, done, _ = env.step(torch.as_tensor(act, device=env.device))
        st, done = env.get_state().cpu().numpy(), done.cpu().numpy()
        for e in np.nonzero(alive)[0]:
            i = starts[e] + k
            assert bool(done[e]) == bool(G["done"][i]), f"episode {e} step {k}: done {done[e]} vs {G['done'][i]}"
            worst = max(worst, assert_close_scaled(st[e], G["post"][i], scale, 2e-5, f"episode {e} step {k} state"))     # free-running; measured 3.4e-6
    print(f"[G3 free-running] {n} episodes, {int(lens.sum())} steps, worst scaled state drift {worst:.2e}")


@pytest.mark.parametrize("name,integ,h", [("euler_0.04", "euler", 0.04), ("euler_0.1", "euler", 0.1),
                                           ("rk4_0.04", "rk4", 0.04), ("rk4_0.1", "rk4", 0.1)])
def test_g6_fixed_step_maps(torch_cuda, cfg, P, name, integ, h):
    """Reference RHS + textbook Euler / RK4 single steps (the parity oracle of BASELINE configs 2-3)."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    g6 = load_golden("g6_fixed_step.npz")
    n = len(g6["pre"])
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, integrator=integ, timestep=h, n_substeps=1)
    env.set_state(g6["pre"].astype(np.float32), m0=g6["m0"].astype(np.float32))
    env.step(torch.as_tensor(g6["action"], device=env.device))
    got = env.get_state().cpu().numpy().astype(np.float64)
    want = g6[f"single_{name}"].copy()
    want[:, 6:10] /= np.linalg.norm(want[:, 6:10], axis=1, keepdims=True)   # the kernel renormalises (simulator.py:97)
    # the fixture applies the bare map (no event handling); the kernel localises ground crossings, so
    # the handful of cases that cross x = 0 inside this step are checked for the event instead
    cross = (g6["pre"][:, 0] >= 0) & (want[:, 0] <= 0)
    assert 0 < cross.sum() < 20
    fl = env.flags.cpu().numpy()
    np.testing.assert_array_equal((fl & 1).astype(bool), cross)
    assert np.abs(got[cross, 0]).max() < 1e-4
    got, want = got[~cross], want[~cross]
    e = assert_close_scaled(got, want, P.normalizer[None, :], 1e-6, f"G6 {name}")     # measured 1.9e-7
    print(f"[G6 {name}] worst scaled err {e:.2e}")


def _fp32_drift_floor(ys, acts, h, integ, P, trials=5):
    """What ANY float32 rollout of this trajectory drifts by, from the float64 oracle itself: the same fixed-step map in
    float64, with the carried state rounded to float32 after every step (trial 0) or perturbed by a uniform random relative
    error of at most half a float32 ulp per component and step (the other trials: rounding errors do not repeat between
    implementations).  Returns the largest of the trials' worst scaled deviations from the golden float64 trajectory.  The
    deviation is not a random walk: these flights amplify a perturbation by two orders of magnitude around a few steps (attitude
    transients: step 391 of euler_0.04 #0, 196 of rk4_0.04 #1, 113 of rk4_0.1 #1) and let it decay again, the same steps for
    every trial and for the GPU."""
    from oracle import oracle_np as onp

    worst = 0.0
    m0 = float(ys[0][13])
    for trial in range(trials):
        rng = np.random.RandomState(100 + trial)
        y = ys[0].astype(np.float32).astype(np.float64)
        for k in range(len(acts)):
            y, _ = onp.sim_step_fixed(y, onp.denormalize_action(acts[k]), m0, h, integ)
            y = y.astype(np.float32).astype(np.float64) if trial == 0 else y * (1.0 + rng.uniform(-1, 1, 14) * 2.0 ** -24)
            worst = max(worst, float((np.abs(y - ys[k + 1]) / np.maximum(np.abs(ys[k + 1]), P.normalizer)).max()))
    return worst


@pytest.mark.parametrize("name,integ,h", [("euler_0.04", "euler", 0.04), ("rk4_0.04", "rk4", 0.04), ("rk4_0.1", "rk4", 0.1)])
def test_g6_free_running_rollouts(torch_cuda, cfg, P, name, integ, h):
    """Free-running (not teacher-forced) fixed-step rollouts of 170-520 steps against golden G6's float64 trajectories.  The
    expectation is derived, not a flat number (round-3 review: 1e-4 sat at 1.4x the measured 7.1e-5): a float32 rollout cannot
    stay closer to the float64 one than the trajectory's own sensitivity to half-ulp perturbations of the state allows
    (_fp32_drift_floor, computed here from the oracle: 1e-6 ... 8e-5 by flight); the kernel's arithmetic adds a few ulps per
    step on top of the storage rounding, so it must stay within 4x that floor (measured on MI355X: 0.2 ... 2.3x)."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    g6 = load_golden("g6_fixed_step.npz")
    ys = [g6[f"roll_{name}_{r}_states"] for r in range(2)]
    acts = [g6[f"roll_{name}_{r}_actions"] for r in range(2)]
    floors = [_fp32_drift_floor(ys[r], acts[r], h, integ, P) for r in range(2)]
    env = BatchedRocket6DOF(2, cfg, auto_reset=False, integrator=integ, timestep=h, n_substeps=1)
    env.set_state(np.stack([y[0] for y in ys]).astype(np.float32))
    nmax = max(len(a) for a in acts)
    worst = [0.0, 0.0]
    curve = [[], []]
    for k in range(nmax):
        a = np.zeros((2, 3), np.float32)
        for r in range(2):
            if k < len(acts[r]):
                a[r] = acts[r][k]
        env.step(torch.as_tensor(a, device=env.device))
        st = env.get_state().cpu().numpy()
        for r in range(2):
            if k < len(acts[r]):
                e = assert_close_scaled(st[r], ys[r][k + 1], P.normalizer, 4.0 * floors[r], f"{name} rollout {r} step {k}")
                curve[r].append(e)
                worst[r] = max(worst[r], e)
    # Teeth (round-4 review, Weak 1): 4 x floor alone would let a regression that doubles the drift pass on the flights that
    # matter (rk4_0.04 #0: measured 5.2e-5 against 4 x 9.6e-5).  The whole-flight bound is therefore
    #     min(4 x floor, max(2 x the drift measured on MI355X in round 4, 1 x floor))
    # -- twice the recorded value (tests/golden/g6_measured_drift.json), except where that value lies below the flight's own
    # float32 floor (a drift below the floor is luck of the rounding, not a property to hold a kernel to).
    rec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_measured_drift.json")))[name]
    for r in range(2):
        measured = rec["worst_scaled_drift_measured_on_mi355x"][r]
        bound = min(4.0 * floors[r], max(2.0 * measured, floors[r]))
        print(f"[G6 rollout {name} #{r}] worst scaled drift over {len(acts[r])} steps {worst[r]:.2e} = {worst[r] / floors[r]:.2f} x the "
              f"float32 floor of this flight ({floors[r]:.2e}); recorded in round 4: {measured:.2e}; asserted <= {bound:.2e} "
              f"(per step <= 4 x floor)")
        assert worst[r] <= bound, (name, r, worst[r], bound)
    try:      # the drift as a function of the step count (gpurun_out/g6_drift_curves.json)
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "g6_drift_curves.json")
        data = json.load(open(path)) if os.path.exists(path) else {}
        data[name] = {"gpu": curve, "fp32_floor": floors}
        os.makedirs(os.path.dirname(path), exist_ok=True)
        json.dump(data, open(path, "w"))
    except OSError:
        pass


def test_velocity_shaping_vs_oracle(torch_cuda, cfg):
    """reward_shaping_type == 'velocity' (rocket_env.py:345-350, :646-674) against the CPU oracle."""
    torch = torch_cuda
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    vcfg = json.loads(json.dumps(cfg))
    vcfg["reward_shaping_type"] = "velocity"
    vcfg["landing_params"]["waypoint"] = 50
    G = load_golden("g3_env_rollouts.npz")
    idx = np.arange(0, len(G["reward"]), 7)
    G4 = load_golden("g4_ground_events.npz")
    pre = np.concatenate([G["pre"][idx], G4["pre"][::5]])
    m0 = np.concatenate([G["m0"][idx], G4["m0"][::5]])
    act = np.concatenate([G["action"][idx], G4["action"][::5]])
    v0 = np.linalg.norm(pre[:, 3:6], axis=1) * 1.1
    n = len(pre)
    env = BatchedRocket6DOF(n, vcfg, auto_reset=False, record_terms=True)
    env.set_state(pre.astype(np.float32), m0=m0.astype(np.float32), v0=v0.astype(np.float32))
    obs, rew, done, _ = env.step(torch.as_tensor(act, device=env.device))
    terms = env.terms[:7, :n].t().cpu().numpy()
    Pv = onp.OracleParams.from_config(vcfg)
    for i in range(n):
        o = onp.env_step(pre[i].astype(np.float32).astype(np.float64), float(np.float32(m0[i])), act[i], Pv, "rk4",
                         v0=float(np.float32(v0[i])))
        assert bool(done[i]) == o["done"]
        assert_close_scaled(terms[i], o["terms"], TERM_SCALE, 2e-6, f"velocity shaping terms {i}")     # measured 3.7e-7


@pytest.mark.parametrize("tag", ["wp50", "wp400"])
@pytest.mark.parametrize("integ", ["rk4", "rk45"])
def test_g10_velocity_shaping_vs_reference(torch_cuda, tag, integ):
    """reward_shaping_type == 'velocity' (rocket_env.py:345-350, :646-674: _compute_vtarg) -- the HIP path against
    the REFERENCE's own run (golden G10: two waypoints, whole flights + ground events, v0 = |IC velocity|)."""
    torch = torch_cuda
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g10_velocity_shaping.npz")
    vcfg = json.loads(str(G[f"config_{tag}"]))
    Pv = onp.OracleParams.from_config(vcfg)
    g = {k[len(tag) + 1:]: G[k] for k in G.files if k.startswith(tag + "_")}
    n = len(g["reward"])
    env = BatchedRocket6DOF(n, vcfg, auto_reset=False, record_terms=True, integrator=integ)
    assert env.term_names[0] == "vtarg_tracking"
    env.set_state(g["pre"].astype(np.float32), m0=g["m0"].astype(np.float32), v0=g["v0"].astype(np.float32))
    obs, rew, done, infos = env.step(torch.as_tensor(g["action"], device=env.device))
    torch.cuda.synchronize()
    e = _check_against_golden(env, obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), g, Pv, f"G10 {tag} {integ}")
    assert e[0] < 5e-6 and e[2] < 5e-6     # measured 9.1e-7 / 8.8e-7
    assert "vtarg_tracking" in infos[0]["rewards_dict"]


@pytest.mark.parametrize("tag", ["a", "b"])
@pytest.mark.parametrize("integ", ["rk4", "rk45"])
def test_g11_other_configs_vs_reference(torch_cuda, tag, integ):
    """env_configs other than config.yaml -- timestep 0.05 / 0.2 (two RK4 sub-steps by the kernel's rule), moved IC /
    ICRange, every reward coefficient incl. a non-zero w_r_f, tighter attitude limits, other landing radius / velocity /
    attitude limits -- through `make_params` and the kernel, against the REFERENCE's own runs (golden G11)."""
    torch = torch_cuda
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g11_other_configs.npz")
    vcfg = json.loads(str(G[f"config_{tag}"]))
    Pv = onp.OracleParams.from_config(vcfg)
    g = {k[len(tag) + 1:]: G[k] for k in G.files if k.startswith(tag + "_")}
    n = len(g["reward"])
    env = BatchedRocket6DOF(n, vcfg, auto_reset=False, record_terms=True, integrator=integ)
    assert abs(env.timestep - vcfg["timestep"]) < 1e-7 and (integ == "rk45" or env.params.n_substeps == (1 if tag == "a" else 2))
    np.testing.assert_allclose(env.state_normalizer, g["normalizer"], rtol=1e-6)
    env.set_state(g["pre"].astype(np.float32), m0=g["m0"].astype(np.float32))
    obs, rew, done, _ = env.step(torch.as_tensor(g["action"], device=env.device))
    torch.cuda.synchronize()
    e = _check_against_golden(env, obs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), g, Pv, f"G11 {tag} {integ}")
    assert int((g["terms"][:, 3] != 0).sum()) > 50, "the changed attitude limit is exercised"


def test_reset_matches_oracle_stream(torch_cuda, cfg, P):
    """Rocket6DOF.reset (rocket_env.py:180-199): the Philox initial-condition stream is bit-identical
    to the oracle's for the 10 plain components; the quaternion (normalised in fp32) to 1 ulp."""
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF
    from rl_rocket_6dof_amd.batched_env import _EXPLICIT_RESET_BASE

    n = 777
    env = BatchedRocket6DOF(n, cfg, seed=1234, env_offset=10_000_000_000)
    obs = env.reset().cpu().numpy()
    st = env.get_state().cpu().numpy()
    plain = [0, 1, 2, 3, 4, 5, 10, 11, 12, 13]
    for i in range(0, n, 37):
        want = onp.reset_sample(1234, 10_000_000_000 + i, _EXPLICIT_RESET_BASE, P)
        np.testing.assert_array_equal(st[i, plain], want[plain])
        np.testing.assert_allclose(st[i, 6:10], want[6:10], rtol=0, atol=1.2e-7)
        np.testing.assert_allclose(obs[i], want / P.normalizer, rtol=3e-7, atol=1e-9)
    np.testing.assert_allclose(env.epconst[0, :n].cpu().numpy(), st[:, 13])
    np.testing.assert_allclose(env.epconst[1, :n].cpu().numpy(), np.linalg.norm(st[:, 3:6], axis=1), rtol=1e-6)
    st2 = env.reset(), env.get_state().cpu().numpy()
    assert not np.array_equal(st, st2[1]), "second reset must draw new initial conditions"


@pytest.mark.parametrize("n", [1, 63, 64, 255, 256, 257, 1000])
def test_ragged_sizes_match_full_batch(torch_cuda, cfg, n):
    """Ragged / tiny batches (partial workgroups take the scalar edge path) give the same answers as
    the same envs inside a large batch."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g3_env_rollouts.npz")
    big = BatchedRocket6DOF(1024, cfg, auto_reset=False, record_terms=True)
    pre = np.resize(G["pre"], (1024, 14)).astype(np.float32)
    act = np.resize(G["action"], (1024, 3)).astype(np.float32)
    m0 = np.resize(G["m0"], 1024).astype(np.float32)
    big.set_state(pre, m0=m0)
    bo, br, bd, _ = big.step(torch.as_tensor(act, device=big.device))
    small = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True)
    small.set_state(pre[:n], m0=m0[:n])
    so, sr, sd, _ = small.step(torch.as_tensor(act[:n], device=small.device))
    assert torch.equal(so, bo[:n]) and torch.equal(sr, br[:n]) and torch.equal(sd, bd[:n])
    assert torch.equal(small.get_state(), big.get_state()[:n])


def test_wrapped_variant_timelimit_clip_removemass(torch_cuda, cfg):
    """Fused make_env() stack (main_6DOF.py:44-53): RemoveMassFromObs, ClipReward(-1,100),
    TimeLimit, Monitor episode stats, auto-reset with terminal_observation."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n, T = 512, 40
    env = BatchedRocket6DOF(n, cfg, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=T, seed=7)
    raw = BatchedRocket6DOF(n, cfg, auto_reset=False, seed=7, record_terms=True)
    obs0 = env.reset()
    raw.reset()
    assert obs0.shape == (n, 13)
    assert torch.equal(env.get_state(), raw.get_state())
    g = torch.Generator(device="cpu").manual_seed(0)
    ret = torch.zeros(n)
    for k in range(T):
        a = (torch.rand((n, 3), generator=g) * 2 - 1).to(env.device)
        pre_state = env.get_state().clone()
        m0_pre = env.epconst[0, :n].clone()
        obs, rew, done, infos = env.step(a)
        raw.set_state(pre_state, m0=m0_pre)
        ro, rr, rd, _ = raw.step(a)
        # reward is the clipped raw reward; obs is the first 13 entries (where no reset happened)
        assert torch.equal(rew, rr.clamp(-1, 100))
        nd = ~done
        assert torch.equal(obs[nd], ro[nd][:, :13])
        ret += rew.cpu()
        if k < T - 1:
            assert torch.equal(done, rd), "before the limit only env terminations end episodes"
        else:
            assert bool(done.all()), "TimeLimit ends every remaining episode at max_episode_steps"
            fl = env.flags.cpu().numpy()
            trunc = (fl & 4) != 0
            np.testing.assert_array_equal(trunc, ~rd.cpu().numpy())
        d = done.cpu().numpy()
        for i in np.nonzero(d)[0][:5]:
            info = infos[int(i)]
            np.testing.assert_array_equal(info["terminal_observation"], ro[i, :13].cpu().numpy())
            assert info["episode"]["l"] >= 1 and "TimeLimit.truncated" in info
    stats = env.episode_stats()
    assert stats["episodes"] >= n and stats["length_sum"] == T * n
    assert abs(stats["return_sum"] - float(ret.sum())) < 1e-3 * max(1.0, abs(float(ret.sum())))
    assert stats["truncated"] + stats["ground"] + stats["bounds"] >= stats["episodes"]


def test_wrapped_variant_vs_c_oracle(torch_cuda, cfg):
    """The fused make_env() stack (main_6DOF.py:33-53, wrappers.py:119-126: RemoveMassFromObs, ClipReward(-1,100),
    TimeLimit, Monitor, auto-reset) against the ORACLE's independent C restatement of the same stack, teacher-forced
    step by step: reward, done, TimeLimit.truncated, running and final episode return / length, the 13-wide
    observation (fresh one for the envs that were reset), the terminal observation."""
    torch = torch_cuda
    from oracle import oracle_c as oc
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    n, T, seed = 768, 48, 7
    P = onp.OracleParams.from_config(cfg)
    env = BatchedRocket6DOF(n, cfg, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=T, seed=seed)
    ref = oc.BatchOracle(n, cfg, seed=seed, track=True, auto_reset=True, clip_reward=(-1, 100), max_episode_steps=T)
    env.reset()
    # low-altitude, fast states for a part of the batch so that ground events / bounds / goal terms occur well before T
    G4 = load_golden("g4_ground_events.npz")
    st = env.get_state().cpu().numpy()
    k4 = min(n // 2, len(G4["pre"]))
    st[:k4] = G4["pre"][:k4].astype(np.float32)
    env.set_state(st)
    g = torch.Generator(device="cpu").manual_seed(0)
    n_done = n_trunc = n_clip = 0
    worst_obs = worst_rew = 0.0
    for k in range(T + 8):
        a = (torch.rand((n, 3), generator=g) * 2 - 1)
        # teacher forcing: the oracle starts every step from the HIP path's pre-step state and counters
        ref.state[:] = env.get_state().cpu().numpy().astype(np.float64)
        ref.epconst[:, 0] = env.epconst[0, :n].cpu().numpy()
        ref.epconst[:, 1] = env.epconst[1, :n].cpu().numpy()
        ref.ep_step[:] = env.ep_step.cpu().numpy().astype(np.uint32)
        ref.ep_return[:] = env.ep_return.cpu().numpy().astype(np.float64)
        ref.step_count = env._step_count
        obs, rew, done, infos = env.step(a.to(env.device))
        ro, rr, rd = ref.step(a.numpy())
        d = done.cpu().numpy()
        np.testing.assert_array_equal(d.astype(np.uint8), rd, err_msg=f"step {k}: done")
        fl = env.flags.cpu().numpy()
        np.testing.assert_array_equal(fl, ref.flags, err_msg=f"step {k}: ground / bounds / truncated flags")
        worst_rew = max(worst_rew, assert_close_scaled(rew.cpu().numpy(), rr, REWARD_SCALE, 6e-6, f"step {k}: clipped reward"))     # measured 1.3e-6
        assert float(rew.min()) >= -1.0 and float(rew.max()) <= 100.0
        n_clip += int(((rr == -1.0) | (rr == 100.0)).sum())
        worst_obs = max(worst_obs, assert_close_scaled(obs.cpu().numpy(), ro[:, :13], 1.0, 3e-6, f"step {k}: obs (13-wide, reset envs included)"))     # measured 6.0e-7
        # running counters of the live envs, Monitor's episode record and the terminal observation of the finished ones
        np.testing.assert_array_equal(env.ep_step.cpu().numpy(), ref.ep_step.astype(np.int32))
        assert_close_scaled(env.ep_return.cpu().numpy(), ref.ep_return, 1.0, 1e-6, f"step {k}: running return")     # measured 1.3e-7
        if d.any():
            idx = np.nonzero(d)[0]
            np.testing.assert_array_equal(env.fin_length.cpu().numpy()[idx], ref.fin_length[idx].astype(np.int32))
            assert_close_scaled(env.fin_return.cpu().numpy()[idx], ref.fin_return[idx], 1.0, 6e-6, f"step {k}: episode return")     # measured 1.2e-6
            assert_close_scaled(env.final_obs.cpu().numpy()[idx], ref.final_obs[idx, :13], 1.0, 1e-6, f"step {k}: terminal obs")     # measured 1.8e-7
            i = int(idx[0])
            info = infos[i]
            assert info["episode"]["l"] == int(ref.fin_length[i]) and info["TimeLimit.truncated"] == bool(ref.flags[i] & 4)
            np.testing.assert_allclose(info["terminal_observation"], ref.final_obs[i, :13], rtol=0, atol=2e-5)
        n_done += int(d.sum())
        n_trunc += int(((fl & 4) != 0).sum())
    print(f"[wrapped vs C oracle] {T + 8} steps x {n} envs: episodes {n_done}, truncated {n_trunc}, clipped rewards {n_clip}, "
          f"worst obs err {worst_obs:.2e}, reward err {worst_rew:.2e}")
    assert n_done > n and n_trunc > n // 4 and n_clip > 0


def test_single_env_surface(torch_cuda, cfg):
    """gym-0.21 surface of Rocket6DOF (what SB3's check_env exercises, test_6DOF_sb_integration.py:24)."""
    from rl_rocket_6dof_amd import Rocket6DOF

    env = Rocket6DOF(**cfg)
    obs = env.reset()
    assert obs.dtype == np.float32 and obs.shape == (14,)
    assert env.observation_space.shape == (14,) and env.action_space.shape == (3,)
    G = load_golden("g3_env_rollouts.npz")
    ep0 = G["episode"] == 0
    obs = env.reset_to(G["x0"][0])
    total = 0.0
    for k, i in enumerate(np.nonzero(ep0)[0]):
        obs, rew, done, info = env.step(G["action"][i])
        assert isinstance(rew, float) and isinstance(done, bool) and isinstance(info, dict)
        assert obs.dtype == np.float32 and obs.shape == (14,)
        for key in ("rewards_dict", "is_done", "state_history", "action_history", "timesteps", "bounds_violation",
                    "atarg_tracking", "final_velocity"):
            assert key in info
        assert done == bool(G["done"][i])
        assert_close_scaled(obs, G["obs"][i], 1.0, 1e-4, f"single-env obs step {k}")
        total += rew
    assert done and len(env.SIM.states) == ep0.sum() + 1 and env.SIM.t == round(0.1 * ep0.sum(), 3)
    env.close()


@pytest.mark.parametrize("name,kw", [("annealing", dict(reward_annealing=True)),
                                     ("vertical", dict(vertical_attitude_reward=(1e-3, -0.5))),
                                     ("both", dict(reward_annealing=True, vertical_attitude_reward=(1e-3, -0.5)))])
def test_g7_reward_wrappers_vs_reference(torch_cuda, cfg, name, kw):
    """RewardAnnealing / VerticalAttitudeReward (my_environment/wrappers/wrappers.py:39-61, :128-155) as
    fused kernel modes, against the reference wrappers run around the reference env."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    g7 = load_golden("g7_wrappers.npz")
    pre, m0, act = g7[f"{name}_pre"], g7[f"{name}_m0"], g7[f"{name}_action"]
    n = len(pre)
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True, **kw)
    env.set_state(pre.astype(np.float32), m0=m0.astype(np.float32))
    obs, rew, done, infos = env.step(torch.as_tensor(act, device=env.device))
    np.testing.assert_array_equal(done.cpu().numpy(), g7[f"{name}_done"])
    assert_close_scaled(rew.cpu().numpy(), g7[f"{name}_reward"], REWARD_SCALE, 1e-6, f"G7 {name} reward")     # measured 1.8e-7
    terms = env.terms[:, :n].t().cpu().numpy()
    np.testing.assert_array_equal(terms[:, 7] != 0, g7[f"{name}_vatt"] != 0)
    assert_close_scaled(terms[:, 7], g7[f"{name}_vatt"], 1.0, 1.3e-6, f"G7 {name} vertical attitude term")     # measured 2.4e-7
    if "reward_annealing" in kw:
        assert_close_scaled(terms[:, 1], g7[f"{name}_tpen"], 1e-2, 1e-6, f"G7 {name} thrust penalty")     # measured 9.3e-8
        assert infos[0]["rewards_dict"]["eta"] == 0.0 and "vertical_attitude_reward" in infos[0]


def test_terminal_errors_montecarlo(torch_cuda, cfg):
    """montecarlo_script.py:33-51 statistics from the terminal observations of finished envs."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g4_ground_events.npz")
    last = np.nonzero(G["done"] == 1)[0]
    n = len(last)
    env = BatchedRocket6DOF(n, cfg, auto_reset=True, seed=1)
    env.set_state(G["pre"][last].astype(np.float32), m0=G["m0"][last].astype(np.float32))
    env.step(torch.as_tensor(G["action"][last], device=env.device))
    te = env.terminal_errors()
    assert len(te["env_index"]) == n
    post = G["post"][last]
    assert_close_scaled(te["final_position_error"].cpu().numpy(), np.linalg.norm(post[:, 0:3], axis=1), 1.0, 1.3e-6, "position error")     # measured < 2.5e-7
    assert_close_scaled(te["final_velocity_error"].cpu().numpy(), np.linalg.norm(post[:, 3:6], axis=1), 1.0, 1.3e-6, "velocity error")     # measured < 2.5e-7
    assert_close_scaled(te["attitude_error"].cpu().numpy(), 0.5 * np.rad2deg(np.arccos(post[:, 6])), 1.0, 1e-4, "attitude error")
    assert_close_scaled(te["angular_velocity_error"].cpu().numpy(), np.linalg.norm(post[:, 10:13], axis=1), 0.1, 1e-6, "omega error")     # measured 1.3e-7
    assert_close_scaled(te["used mass"].cpu().numpy(), post[:, 13], 1.0, 1e-6, "terminal mass")     # measured 9.3e-8


@pytest.mark.parametrize("fixture", ["g3_env_rollouts.npz", "g4_ground_events.npz"])
def test_rk45_adaptive_twin_vs_reference(torch_cuda, cfg, P, fixture):
    """integrator="rk45": the in-kernel adaptive Dormand-Prince with scipy's step control and the
    dense-output ground event (the algorithm of simulator.py:83-88 itself, in fp32).  Same 1e-5 bound,
    same bit-exact masks; additionally the per-step RHS-evaluation count is compared with scipy's nfev."""
    G = load_golden(fixture)
    env, obs, rew, done, _ = _teacher_forced(torch_cuda, cfg, G, integrator="rk45", record_nfev=True)
    _check_against_golden(env, obs, rew, done, G, P, fixture + " rk45", tol=TOL_STEPS if fixture.startswith("g3") else TOL_EVENTS)
    nfev = env.nfev.cpu().numpy().astype(int)
    same = float((nfev == G["nfev"]).mean())
    print(f"[{fixture} rk45] nfev identical to scipy on {100 * same:.1f}% of the steps "
          f"(mean {nfev.mean():.2f} vs {G['nfev'].mean():.2f})")
    assert same > 0.97 and abs(nfev.mean() - G["nfev"].mean()) < 0.2


@pytest.mark.parametrize("n,grid,K,kw", [(150_037, 48, 24, dict()),                       # 12 tiles per wave: hand-over, hosting, list recycling
                                         (9_001, 8, 24, dict(remove_mass=True, clip_reward=(-1, 100), max_episode_steps=7, reward_shaping="velocity")),
                                         (40_003, 16, 10, dict(timestep=0.5)),          # the reference's simulator-test step: 4-6 RK steps per env step, guests under way for several tiles
                                         (1 << 20, 0, 6, dict())])                      # the bench's own launch shape: 4 tiles per wave
def test_rk45_straggler_pass_is_bit_identical(torch_cuda, cfg, n, grid, K, kw):
    """integrator="rk45": a wave hands its last few unfinished lanes (the 0.8 % of env steps that need a third or fourth RK step,
    simulator.py:83-88 through scipy's step control) over as GUESTS -- they ride in idle lanes of the wave's own later attempts -- instead
    of following them (r6_kernels.hip step_kernel, r6_tune("rk45_defer")).  Whoever integrates a lane does the same arithmetic: every output of every
    step -- observation, reward, done, flag byte, nfev, the 7 reward terms, terminal observation / return / length, the state, the
    episode counters and statistics -- must equal the run without the hand-over, bit for bit; ragged last tile, in-kernel resets,
    the fused wrappers and velocity shaping included.  Also: lanes that need a third RK step exist (nfev >= 20), and a threshold of
    64 lanes (the list is full at once: what does not fit goes on in the wave, the list is recycled after every tile) changes
    nothing either."""
    torch = torch_cuda
    import json

    from rl_rocket_6dof_amd import BatchedRocket6DOF, _lib

    c2 = json.loads(json.dumps(cfg))
    if kw.pop("reward_shaping", None) == "velocity":
        c2["reward_shaping_type"] = "velocity"
        c2["landing_params"]["waypoint"] = 50
    # (a wave hands lanes over only when it has a LATER tile to host them in: the grid is pinned so that every wave walks several
    #  tiles -- with the default grid a batch below 524 288 envs is one tile per wave and the hand-over never happens)
    gen = torch.Generator(device="cuda:0").manual_seed(7)
    acts = [(torch.rand((n, 3), generator=gen, device="cuda:0") * 2 - 1) for _ in range(K)]
    G4 = load_golden("g4_ground_events.npz")
    runs = []
    try:
        _lib.tune("grid_blocks", grid)
        for defer in (0, 6, 64) if n < (1 << 20) else (0, 6):
            _lib.tune("rk45_defer", defer)
            env = BatchedRocket6DOF(n, c2, integrator="rk45", seed=3, record_flags=True, record_terms=True, record_nfev=True,
                                    track_episodes=True, **kw)
            env.reset()
            st = env.get_state().cpu().numpy()
            k4 = min(n // 3, len(G4["pre"]))            # a third of the batch a few metres above the pad: ground events, goal rewards, resets
            st[:k4] = G4["pre"][:k4].astype(np.float32)
            env.set_state(st)
            snaps = []
            for k in range(K):
                env.step_async(acts[k])
                snaps.append([t.clone() for t in (env.obs, env.reward, env.done, env.flags, env.nfev, env.terms, env.final_obs, env.fin_return,
                                                  env.fin_length, env.state, env.ep_step, env.ep_return)])
            torch.cuda.synchronize()
            snaps.append([env.stats.sum(0).clone()])
            runs.append(snaps)
            del env
    finally:
        _lib.tune("rk45_defer", 6)
        _lib.tune("grid_blocks", 0)
    nf = torch.stack([s[4] for s in runs[0][:K]])
    assert int((nf >= 20).sum()) > K, "no env step needed a third RK step: the hand-over was not exercised"
    assert int(torch.stack([s[2] for s in runs[0][:K]]).sum()) > 200      # episode ends (ground events of the G4 states, bounds, time limit): resets ran
    for other in runs[1:]:
        for k, (a, b) in enumerate(zip(runs[0], other)):
            for i, (x, y) in enumerate(zip(a, b)):
                assert torch.equal(x, y), (k, i)


def test_rk45_simulator_smoke_case_single_step(torch_cuda):
    """test_6DOF_simulator.py:3-7 with the adaptive integrator: dt = 0.5 in ONE env step, like the reference."""
    from rl_rocket_6dof_amd import Simulator6DOF

    g1 = load_golden("g1_sim_step.npz")
    sim = Simulator6DOF(list(g1["ic"]), integrator="rk45")
    state, status = sim.step(list(g1["u"]))
    assert status == 0
    scale = np.array([100, 100, 100, 5, 5, 5, 1, 1, 1, 1, 1e-2, 1e-2, 1e-2, 5e4])
    assert_close_scaled(state, g1["state"], scale, 1e-6, "G1 Simulator6DOF.step rk45")     # measured 1.0e-7


def test_edge_cases_nan_unclipped_actions_substeps(torch_cuda, cfg, P):
    """Edge cases the reference's semantics define: a non-finite position counts as a bounds violation
    (gym Box.contains is False for NaN, rocket_env.py:591-593); actions are NOT clipped by the env
    (rocket_env.py:509-521), so |a| > 1 must follow the same formulas (large gimbal angles take the
    full-range sincos path); a 0.5 s env step is integrated as 5 RK4 sub-steps."""
    torch = torch_cuda
    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    G = load_golden("g3_env_rollouts.npz")
    n = 256
    pre = G["pre"][:n].astype(np.float32)
    m0 = G["m0"][:n].astype(np.float32)
    rng = np.random.RandomState(5)
    act = rng.uniform(-3.5, 3.5, (n, 3)).astype(np.float32)       # far outside the Box
    act[:, 2] = rng.uniform(-1.0, 1.6, n)
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True)
    env.set_state(pre, m0=m0)
    obs, rew, done, _ = env.step(torch.as_tensor(act, device=env.device))
    post = env.get_state().cpu().numpy()
    for i in range(0, n, 3):
        o = onp.env_step(pre[i].astype(np.float64), float(m0[i]), act[i], P, "rk4")
        assert bool(done[i]) == o["done"]
        assert_close_scaled(post[i], o["post"], P.normalizer, 1e-6, f"unclipped action env {i}")     # measured 1.3e-7
        assert_close_scaled(float(rew[i]), o["reward"], REWARD_SCALE, 1e-6, f"unclipped action reward {i}")     # measured 7.1e-8

    bad = pre.copy()
    bad[0, 0] = np.nan
    bad[1, 1] = np.inf
    env.set_state(bad, m0=m0)
    obs, rew, done, _ = env.step(torch.zeros((n, 3), device=env.device))
    fl = env.flags.cpu().numpy()
    assert bool(done[0]) and bool(done[1]) and (fl[0] & 2) and (fl[1] & 2)

    c5 = dict(cfg, timestep=0.5)
    P5 = onp.OracleParams.from_config(c5)
    env5 = BatchedRocket6DOF(n, c5, auto_reset=False)
    assert env5.params.n_substeps == 5
    env5.set_state(pre, m0=m0)
    a5 = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    obs, rew, done, _ = env5.step(torch.as_tensor(a5, device=env5.device))
    post = env5.get_state().cpu().numpy()
    for i in range(0, n, 5):
        o = onp.env_step(pre[i].astype(np.float64), float(m0[i]), a5[i], P5, "rk4", n_substeps=5)
        oa = onp.env_step(pre[i].astype(np.float64), float(m0[i]), a5[i], P5, "adaptive")     # the reference's own integrator
        assert bool(done[i]) == o["done"] == oa["done"]
        assert_close_scaled(post[i], o["post"], P5.normalizer, RTOL, f"dt=0.5 env {i} vs fixed-step oracle")
        assert_close_scaled(post[i], oa["post"], P5.normalizer, RTOL, f"dt=0.5 env {i} vs adaptive (reference) oracle")


def test_quartic_near_double_root_family_vs_c_oracle(torch_cuda, cfg):
    """t_go (rocket_env.py:528-546) where the quartic has a positive minimum right of its only positive root
    (fast approach: v nearly anti-parallel to r): the kernel's Newton must restart on the other branch when a
    step from the flat bottom of that minimum jumps across the inflection point.  500 000 states; the C oracle
    is pinned against np.roots on the same family in tests/test_oracle_c.py."""
    torch = torch_cuda
    from oracle import oracle_c as oc
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    rng = np.random.default_rng(17)
    n = 500_000
    s = np.zeros((n, 14))
    s[:, 6] = 1.0
    s[:, 13] = rng.uniform(25e3, 45e3, n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rmag = rng.uniform(500.0, 4000.0, n)
    s[:, 0:3] = d * rmag[:, None]
    s[:, 3:6] = -d * (rmag * rng.uniform(0.02, 0.25, n))[:, None] + rng.normal(size=(n, 3)) * 20.0
    s = s.astype(np.float32)
    env = BatchedRocket6DOF(n, cfg, auto_reset=False, record_terms=True, timestep=1e-9, n_substeps=1)
    env.set_state(s)
    env.step(torch.zeros((n, 3), device=env.device))
    ref = oc.BatchOracle(n, cfg, dt=1e-9)
    ref.set_state(s.astype(np.float64))
    ref.step(np.zeros((n, 3), np.float32))
    got = env.terms[0, :n].cpu().numpy().astype(np.float64)
    err = np.abs(got - ref.terms[:, 0]) / np.maximum(np.abs(ref.terms[:, 0]), TERM_SCALE[0])   # scale as in G3-G5
    bad = int((err > 1e-5).sum())
    print(f"[quartic family] worst scaled error {err.max():.2e}, {bad} of {n} beyond 1e-5")
    # What remains are quartics whose LARGEST root is itself nearly double (three positive roots, two of them
    # within 1 %): there t_go moves by 1e-4 with one ulp of the reference's own float32 r.v product, and the
    # three implementations (np.roots / C / HIP) scatter by that much among themselves -- a few per million.
    assert bad <= 10 and err.max() < 2e-4, f"tracking term: {bad} states beyond 1e-5, worst {err.max():.3e} at {int(err.argmax())}"


def test_rk45_fuzz_vs_scipy_oracle(torch_cuda, cfg, P):
    """integrator="rk45" on 1500 states far outside the golden distribution (fast tumbling, any attitude,
    ground contact within the step, dt = 0.1 and 0.5) against the numpy/scipy restatement of simulator.py:83-88
    (solve_ivp RK45 + terminal event).  fp32 step control follows scipy's accept/reject decisions except on
    borderline error norms, so: status equal, state within 1e-4 of the normaliser scale everywhere and within
    1e-5 on at least 99 % of the states; nfev equal on at least 90 %."""
    torch = torch_cuda
    import json

    from oracle import oracle_np as onp
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    for dt in (0.1, 0.5):
        c2 = json.loads(json.dumps(cfg))
        c2["timestep"] = dt
        P2 = onp.OracleParams.from_config(c2)
        rng = np.random.default_rng(int(dt * 10))
        n = 750
        s = np.zeros((n, 14))
        s[:, 0] = rng.uniform(5.0, 2500.0, n)
        s[: n // 3, 0] = rng.uniform(0.1, 20.0, n // 3)
        s[:, 1:3] = rng.uniform(-2000.0, 2000.0, (n, 2))
        s[:, 3:6] = rng.uniform(-200.0, 200.0, (n, 3))
        s[: n // 3, 3] = rng.uniform(-60.0, 5.0, n // 3)
        q = rng.normal(size=(n, 4))
        s[:, 6:10] = q / np.linalg.norm(q, axis=1, keepdims=True)
        s[:, 10:13] = rng.uniform(-0.8, 0.8, (n, 3))
        s[:, 13] = rng.uniform(25e3, 45e3, n)
        s = s.astype(np.float32)
        m0 = (s[:, 13] + rng.uniform(0.0, 2e3, n).astype(np.float32)).astype(np.float32)
        a = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
        env = BatchedRocket6DOF(n, c2, integrator="rk45", auto_reset=False, record_flags=True, record_nfev=True, timestep=dt)
        env.set_state(s, m0=torch.as_tensor(m0))
        env.step(torch.as_tensor(a))
        post = env.get_state().cpu().numpy().astype(np.float64)
        flags = env.flags.cpu().numpy()
        nfev = env.nfev.cpu().numpy().astype(int)
        errs, same_nfev, bad_status = [], 0, 0
        for i in range(n):
            y, status, nf = onp.sim_step_adaptive(s[i].astype(np.float64), onp.denormalize_action(a[i]), float(m0[i]), dt)
            bad_status += int((flags[i] & 1) != (status == 1))
            errs.append(float(np.max(np.abs(post[i] - y) / np.maximum(np.abs(y), P2.normalizer))))
            same_nfev += int(nf == nfev[i])
        errs = np.array(errs)
        print(f"[rk45 fuzz dt={dt}] worst {errs.max():.2e}, {100 * (errs < 1e-5).mean():.1f}% within 1e-5, status mismatches "
              f"{bad_status}, nfev equal on {100 * same_nfev / n:.1f}%, events {int((flags & 1).sum())}")
        assert bad_status == 0
        assert errs.max() < 1e-4 and (errs < 1e-5).mean() >= 0.99
        assert same_nfev >= 0.9 * n


@pytest.mark.gpu
@pytest.mark.parametrize("n,kw", [(1, dict(auto_reset=False, record_terms=True)),
                                  (200, dict(auto_reset=True, record_terms=True, max_episode_steps=40, remove_mass=True, clip_reward=(-1, 100))),
                                  (4096, dict(auto_reset=True, record_terms=False, record_nfev=True, integrator="rk45"))])
def test_host_mirror_changes_nothing(torch_cuda, cfg, n, kw):
    """``host_mirror=True`` (every per-env buffer a view of one device allocation + one pinned image; the single-env drop-in
    steps through it): the same bytes in every buffer as the separately allocated env, step by step, and ``infos`` built
    from the pinned image equal to the ones read from the device."""
    torch = torch_cuda
    from rl_rocket_6dof_amd import BatchedRocket6DOF
    from rl_rocket_6dof_amd.batched_env import LazyInfos

    a = BatchedRocket6DOF(n, cfg, seed=5, **kw)
    b = BatchedRocket6DOF(n, cfg, seed=5, host_mirror=True, **kw)
    a.reset()
    b.reset()
    rng = np.random.RandomState(3)
    for k in range(60):
        act = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        _, _, _, infos = a.step(torch.as_tensor(act, device=a.device))
        h = b.step_host(act)
        for name in b._MIRRORED:
            ta = getattr(a, name)
            if name == "_act" or ta is None:
                continue
            assert np.array_equal(ta.cpu().numpy(), h[name], equal_nan=True), (name, k)
        hi = LazyInfos.from_host(b, h)
        for i in (0, n // 2, n - 1):
            x, y = infos[i], hi[i]
            x.get("episode", {}).pop("t", None)
            y.get("episode", {}).pop("t", None)
            assert x.keys() == y.keys()
            for key in x:
                if isinstance(x[key], dict):
                    assert x[key] == y[key], (key, k)
                else:
                    assert np.array_equal(np.asarray(x[key]), np.asarray(y[key]), equal_nan=True), (key, k)
    with pytest.raises(RuntimeError):
        a.fetch_host()
    # SB3's way of stepping (numpy in, numpy out) takes the mirror by itself: the same numbers as the per-tensor read-back
    c = BatchedRocket6DOF(n, cfg, seed=5, numpy_io=True, **kw)
    d = BatchedRocket6DOF(n, cfg, seed=5, numpy_io=True, host_mirror=False, **kw)
    assert c._arena is not None and d._arena is None
    assert np.array_equal(c.reset(), d.reset())
    for k in range(40):
        act = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        oc, od = c.step(act), d.step(act)
        for x, y in zip(oc[:3], od[:3]):
            assert isinstance(x, np.ndarray) and x.dtype == y.dtype and np.array_equal(x, y, equal_nan=True), k
        for i in np.nonzero(oc[2])[0][:4]:
            x, y = oc[3][int(i)], od[3][int(i)]
            x.get("episode", {}).pop("t", None)
            y.get("episode", {}).pop("t", None)
            assert x.keys() == y.keys() and all(np.array_equal(np.asarray(x[q]), np.asarray(y[q]), equal_nan=True) for q in x if not isinstance(x[q], dict))
