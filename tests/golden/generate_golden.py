#!/usr/bin/env python3
"""Generate the golden vectors G1-G6 by RUNNING THE REFERENCE.

TEST INFRASTRUCTURE ONLY -- runs only in the build container (needs /root/reference).
Outputs are small ``.npz`` fixtures committed next to this script; they are data (inputs and the
reference's outputs), never reference source.  Version caveat recorded in every file's ``meta``:
the reference pins scipy 1.7.3 / numpy 1.21.6, the container has scipy 1.15.3 / numpy 2.2.6.

    python tests/golden/generate_golden.py          # rewrites tests/golden/g*.npz

What each file pins (reference file:line):
  g1_sim_step      test_6DOF_simulator.py:3-7  -> Simulator6DOF.step (simulator.py:69-104)
  g2_rhs           Simulator6DOF.RHS (simulator.py:106-143) on random (y,u), incl. un-normalised q
  g3_env_rollouts  Rocket6DOF.reset/step (rocket_env.py:180-231) with config.yaml, random actions
  g4_ground_events same, crafted low-altitude initial conditions (terminal event simulator.py:77-90)
  g5_reward_cases  _compute_reward/_reward_goal/_check_attitude_limits/_check_bounds_violation
                   (rocket_env.py:317-402, :591-593) on crafted post-states
  g6_fixed_step    reference RHS driven by textbook Euler / RK4 (the reference has no fixed-step
                   integrator; this is the parity oracle for BASELINE configs 2-3)
  g7_wrappers      RewardAnnealing / VerticalAttitudeReward (my_environment/wrappers/wrappers.py:39-61,
                   :128-155) around the reference env   (python generate_golden.py --only g7)
  g8_closed_loop   whole landing episodes of the reference env (RemoveMassFromObs view, wrappers.py:119-126)
                   flown by the reference's own trained policies (best_model_*.zip, the checkpoints
                   montecarlo_script.py / main_6DOF.py load with PPO.load), deterministic actions clipped to
                   the Box like SB3's predict.  The checkpoint tensors are read with
                   torch.load(weights_only=True) -- no pickled code is executed -- and stored in the fixture
                   (they are the reference's data files, 10 k floats each)   (--only g8)
  g10_velocity_shaping   the reference env with reward_shaping_type='velocity' (rocket_env.py:345-350, :646-674:
                   _compute_vtarg, two waypoints on either side of the flight, v0 = |IC velocity| recorded):
                   random-action episodes from the config's IC distribution plus low-altitude ones that end
                   in ground events   (--only g10)
  g11_other_configs   the reference env built from env_configs OTHER than config.yaml (timestep 0.05 / 0.2, moved IC and
                   ICRange, every reward coefficient changed incl. a non-zero w_r_f, tighter trajectory / landing limits, a
                   different landing radius and velocity): pins the host-side digestion of env_config (rocket_env.py:22-178)
                   and the parameters' way through the kernel beyond the one shipped configuration   (--only g11)
  g12_reset_draws   50 000 initial conditions drawn by the reference's own Rocket6DOF.reset() (rocket_env.py:180-199: init_space.sample(),
                   quaternion normalised, new Simulator6DOF) after seed(2022): the simulator state of each fresh episode, float32 --
                   a SAMPLE of the reset distribution (gym's RNG stream itself is not reproducible without gym 0.21: Box.sample
                   here is the loader's gym-0.21 restatement, np_random.uniform(low, high))   (--only g12)
  g9_shipped_policy_in_reference   the policy trained on the HIP env (examples/policies/) flown in the REFERENCE
                   env: 16 episodes, terminal states, landing flags   (--only g9)
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _ref_loader import load_reference  # noqa: E402

warnings.simplefilter("ignore")

TERMS = [
    "atarg_tracking",
    "thrust_penalty",
    "eta",
    "attitude_constraint",
    "goal_conditions",
    "final_position",
    "final_velocity",
]


def _meta(extra=None):
    import scipy

    m = {
        "generator": "tests/golden/generate_golden.py",
        "reference": "Tuxliri/RL_Rocket_6DOF @ /root/reference (2024-12-20 snapshot)",
        "numpy": np.__version__,
        "scipy": scipy.__version__,
        "reference_pins": "numpy==1.21.6 scipy==1.7.3 gym==0.21.0 (not installed here)",
        "reward_terms": TERMS,
    }
    if extra:
        m.update(extra)
    return np.array(json.dumps(m))


class _IvpSpy:
    """Wraps scipy.solve_ivp inside the reference simulator module to record nfev / status."""

    def __init__(self, sim_module):
        self.mod = sim_module
        self.orig = sim_module.solve_ivp
        self.last = None
        sim_module.solve_ivp = self

    def __call__(self, *a, **k):
        sol = self.orig(*a, **k)
        self.last = sol
        return sol


def _quat_from_axis_angle(axis, ang):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def _run_episode(env, spy, actions, ic=None, max_steps=None):
    """Drive reference env; returns list of per-step records."""
    if ic is not None:
        orig_sample = env.init_space.sample
        env.init_space.sample = lambda: np.array(ic, dtype=np.float32)
    env.reset()
    if ic is not None:
        env.init_space.sample = orig_sample
    m0 = float(env.SIM.m)
    x0 = np.array(env.SIM.state, dtype=np.float64)
    recs = []
    n = len(actions) if max_steps is None else min(max_steps, len(actions))
    for k in range(n):
        pre = np.array(env.SIM.state, dtype=np.float64)
        a = np.asarray(actions[k], dtype=np.float32)
        obs, rew, done, info = env.step(a)
        rd = info["rewards_dict"]
        recs.append(
            dict(
                pre=pre,
                m0=m0,
                action=a,
                post=np.array(env.SIM.state, dtype=np.float64),
                obs=np.asarray(obs, dtype=np.float32),
                reward=float(rew),
                terms=np.array([float(rd[t] if t in rd else rd["vtarg_tracking"]) for t in TERMS]),   # slot 0: a|v tracking
                done=bool(done),
                bounds=bool(info["bounds_violation"]),
                status=int(spy.last.status),
                nfev=int(spy.last.nfev),
                atarg=np.array(env.atarg_history[-1] if env.atarg_history else env.vtarg_history[-1], dtype=np.float64),
                t=float(env.SIM.t),
                step=k,
            )
        )
        if done:
            break
    return x0, recs


def _pack(recs_by_ep):
    out = {}
    keys = ["pre", "m0", "action", "post", "obs", "reward", "terms", "done", "bounds", "status", "nfev", "atarg", "t", "step"]
    flat = [r for ep in recs_by_ep for r in ep]
    for k in keys:
        out[k] = np.array([r[k] for r in flat])
    out["episode"] = np.array([i for i, ep in enumerate(recs_by_ep) for _ in ep], dtype=np.int32)
    out["done"] = out["done"].astype(np.uint8)
    out["bounds"] = out["bounds"].astype(np.uint8)
    return out


def gen_g7():
    from _ref_loader import load_reference_wrappers

    Sim, Env, cfg = load_reference()
    W = load_reference_wrappers()
    sim_module = sys.modules["my_environment.utils.simulator"]
    spy = _IvpSpy(sim_module)
    rng = np.random.RandomState(7)
    out = {}
    for name in ("annealing", "vertical", "both"):
        base = Env(**cfg)
        if name == "annealing":
            env = W.RewardAnnealing(base)
        elif name == "vertical":
            env = W.VerticalAttitudeReward(base)
        else:
            env = W.VerticalAttitudeReward(W.RewardAnnealing(base))
        rec = dict(pre=[], m0=[], action=[], reward=[], done=[], terms=[], vatt=[], tpen=[])
        for k in range(60):
            icv = np.zeros(14)
            icv[0] = rng.uniform(0.3, 8.0)
            icv[1:3] = rng.uniform(-25, 25, 2)
            icv[3] = rng.uniform(-12, -1)
            icv[4:6] = rng.uniform(-3, 3, 2)
            tilt = rng.uniform(0, 0.6)
            icv[6:10] = _quat_from_axis_angle([0.0, rng.normal(), rng.normal() + 1e-9], tilt)
            icv[10:13] = rng.uniform(-0.2, 0.2, 3)
            icv[13] = rng.uniform(30e3, 41e3)
            orig = base.init_space.sample
            base.init_space.sample = lambda: np.array(icv, dtype=np.float32)
            env.reset()
            base.init_space.sample = orig
            for j in range(40):
                pre = np.array(base.SIM.state, dtype=np.float64)
                a = rng.uniform(-1, 1, 3).astype(np.float32)
                a[2] = rng.uniform(-0.8, 0.2)
                obs, rew, done, info = env.step(a)
                rd = info["rewards_dict"]
                rec["pre"].append(pre)
                rec["m0"].append(float(base.SIM.m))
                rec["action"].append(a)
                rec["reward"].append(float(rew))
                rec["done"].append(bool(done))
                rec["terms"].append([float(rd.get(t, 0.0)) for t in TERMS])
                rec["vatt"].append(float(rd.get("vertical_attitude_reward", 0.0)))
                rec["tpen"].append(float(rd.get("thrust_penalty", 0.0)))
                if done:
                    break
        for k_, v in rec.items():
            out[f"{name}_{k_}"] = np.array(v)
        print("G7", name, "steps", len(rec["reward"]), "vatt!=0", int((np.array(rec["vatt"]) != 0).sum()))
    np.savez_compressed(os.path.join(HERE, "g7_wrappers.npz"), config=np.array(json.dumps(cfg)), meta=_meta(), **out)


def _sb3_forward(sd, obs13):
    """Deterministic SB3 ActorCriticPolicy forward (mean action) from the checkpoint tensors, float32 torch."""
    import torch

    x = torch.as_tensor(obs13, dtype=torch.float32)
    pre = "mlp_extractor.shared_net." if any(k.startswith("mlp_extractor.shared_net.") for k in sd) else "mlp_extractor.policy_net."
    h = torch.tanh(torch.nn.functional.linear(x, sd[pre + "0.weight"], sd[pre + "0.bias"]))
    h = torch.tanh(torch.nn.functional.linear(h, sd[pre + "2.weight"], sd[pre + "2.bias"]))
    return torch.nn.functional.linear(h, sd["action_net.weight"], sd["action_net.bias"]).numpy()


def gen_g8():
    import io
    import zipfile

    import torch

    from _ref_loader import REF_ROOT

    Sim, Env, cfg = load_reference()
    out = {}
    for name in ("best_model_2bo71j9m", "best_model_2i3ipkag"):
        with zipfile.ZipFile(os.path.join(REF_ROOT, name + ".zip")) as z:
            sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)   # tensors only
        tag = name.split("_")[-1]
        for k, v in sd.items():
            out[f"{tag}_w_{k}"] = v.numpy().astype(np.float32)
        env = Env(**cfg)
        env.seed(1234)
        ics, n_steps, terminal, returns, landed, traj_obs, traj_act = [], [], [], [], [], [], []
        for ep in range(12):
            obs = env.reset()
            ics.append(np.array(env.SIM.state, dtype=np.float64))
            ret, k, o_hist, a_hist = 0.0, 0, [], []
            for k in range(1500):
                a = np.clip(_sb3_forward(sd, obs[0:13]), -1.0, 1.0).astype(np.float32)     # SB3 predict clips to the Box
                if k % 25 == 0:
                    o_hist.append(np.asarray(obs[0:13], dtype=np.float32))
                    a_hist.append(a)
                obs, rew, done, info = env.step(a)
                ret += float(rew)
                if done:
                    break
            n_steps.append(k + 1)
            terminal.append(np.array(env.SIM.state, dtype=np.float64))
            returns.append(ret)
            landed.append(float(info["rewards_dict"]["goal_conditions"]) != 0.0)
            pad = 20 - len(o_hist)
            traj_obs.append(np.array(o_hist + [np.zeros(13, np.float32)] * pad)[:20])
            traj_act.append(np.array(a_hist + [np.zeros(3, np.float32)] * pad)[:20])
        out[f"{tag}_ic"] = np.array(ics)
        out[f"{tag}_n_steps"] = np.array(n_steps)
        out[f"{tag}_terminal"] = np.array(terminal)
        out[f"{tag}_return"] = np.array(returns)
        out[f"{tag}_landed"] = np.array(landed)
        out[f"{tag}_obs_every25"] = np.array(traj_obs)
        out[f"{tag}_act_every25"] = np.array(traj_act)
        t = np.array(terminal)
        print("G8", name, "steps", n_steps, "landed", int(np.sum(landed)), "of", len(landed),
              "| |r_T| mean %.2f m, |v_T| mean %.2f m/s, return mean %.1f" % (np.linalg.norm(t[:, 0:3], axis=1).mean(),
                                                                            np.linalg.norm(t[:, 3:6], axis=1).mean(), np.mean(returns)))
    np.savez_compressed(os.path.join(HERE, "g8_closed_loop.npz"), config=np.array(json.dumps(cfg)), meta=_meta(), **out)


def gen_g9():
    """The policy trained on the HIP env (examples/policies/landing_policy_shared_128_64.pt) flown in the REFERENCE env."""
    import torch

    root = os.path.dirname(os.path.dirname(HERE))
    sd = torch.load(os.path.join(root, "examples", "policies", "landing_policy_shared_128_64.pt"), map_location="cpu", weights_only=True)

    def forward(obs13):
        x = torch.as_tensor(obs13, dtype=torch.float32)
        h = torch.tanh(torch.nn.functional.linear(x, sd["pi.0.weight"], sd["pi.0.bias"]))
        h = torch.tanh(torch.nn.functional.linear(h, sd["pi.2.weight"], sd["pi.2.bias"]))
        return torch.nn.functional.linear(h, sd["pi.4.weight"], sd["pi.4.bias"]).numpy()

    Sim, Env, cfg = load_reference()
    env = Env(**cfg)
    env.seed(4321)
    ics, n_steps, terminal, returns, landed = [], [], [], [], []
    for ep in range(16):
        obs = env.reset()
        ics.append(np.array(env.SIM.state, dtype=np.float64))
        ret = 0.0
        for k in range(1500):
            a = np.clip(forward(obs[0:13]), -1.0, 1.0).astype(np.float32)
            obs, rew, done, info = env.step(a)
            ret += float(rew)
            if done:
                break
        n_steps.append(k + 1)
        terminal.append(np.array(env.SIM.state, dtype=np.float64))
        returns.append(ret)
        landed.append(float(info["rewards_dict"]["goal_conditions"]) != 0.0)
    t = np.array(terminal)
    print("G9 steps", n_steps, "landed", int(np.sum(landed)), "of", len(landed),
          "| |r_T| mean %.2f m, |v_T| mean %.2f m/s, return mean %.1f" % (np.linalg.norm(t[:, 0:3], axis=1).mean(),
                                                                        np.linalg.norm(t[:, 3:6], axis=1).mean(), np.mean(returns)))
    np.savez_compressed(os.path.join(HERE, "g9_shipped_policy_in_reference.npz"), config=np.array(json.dumps(cfg)), meta=_meta(),
                        ic=np.array(ics), n_steps=np.array(n_steps), terminal=t, ret=np.array(returns), landed=np.array(landed))


def gen_g10():
    """Rocket6DOF with reward_shaping_type == 'velocity' (rocket_env.py:345-350, :646-674)."""
    import copy

    Sim, Env, cfg = load_reference()
    sim_module = sys.modules["my_environment.utils.simulator"]
    spy = _IvpSpy(sim_module)
    out = {}
    cfgs = {}
    for tag, waypoint in (("wp50", 50.0), ("wp400", 400.0)):
        vcfg = copy.deepcopy(cfg)
        vcfg["reward_shaping_type"] = "velocity"
        vcfg["landing_params"]["waypoint"] = waypoint
        cfgs[tag] = vcfg
        env = Env(**vcfg)
        rng = np.random.RandomState(1000 + int(waypoint))
        eps, x0s, v0s = [], [], []
        for k in range(4):       # whole flights from the config's IC distribution (both branches of the waypoint test)
            acts = rng.uniform(-1, 1, (1000, 3)).astype(np.float32)
            if k >= 2:
                acts = (0.3 * acts + np.array([0.0, 0.0, 0.1 * k])).astype(np.float32)
            x0, recs = _run_episode(env, spy, acts, max_steps=150 if k < 2 else None)
            v0 = float(np.linalg.norm(np.asarray(env.SIM.states[0])[3:6]))     # rocket_env.py:649-651
            for r in recs:
                r["v0"] = v0
            eps.append(recs)
            x0s.append(x0)
        for k in range(24):      # low-altitude descents: ground events, goal / final-velocity terms, x around the waypoint
            icv = np.zeros(14)
            icv[0] = rng.uniform(0.5, 1.6 * waypoint if k % 2 else 30.0)
            icv[1:3] = rng.uniform(-60, 60, 2)
            icv[3] = rng.uniform(-45, -2)
            icv[4:6] = rng.uniform(-10, 10, 2)
            icv[6:10] = _quat_from_axis_angle([0.0, rng.normal(), rng.normal() + 1e-9], rng.uniform(0, 0.5))
            icv[10:13] = rng.uniform(-0.3, 0.3, 3)
            icv[13] = rng.uniform(30e3, 41.5e3)
            acts = rng.uniform(-1, 1, (300, 3)).astype(np.float32)
            x0, recs = _run_episode(env, spy, acts, ic=icv)
            v0 = float(np.linalg.norm(np.asarray(env.SIM.states[0])[3:6]))
            for r in recs:
                r["v0"] = v0
            eps.append(recs)
            x0s.append(x0)
        g = _pack(eps)
        g["v0"] = np.array([r["v0"] for ep in eps for r in ep])
        g["x0"] = np.array(x0s)
        for k_, v in g.items():
            out[f"{tag}_{k_}"] = v
        print("G10", tag, "steps", len(g["reward"]), "ground events", int((g["status"] == 1).sum()), "done", int(g["done"].sum()),
              "x>waypoint", int((g["post"][:, 0] > waypoint).sum()))
    np.savez_compressed(os.path.join(HERE, "g10_velocity_shaping.npz"), config_wp50=np.array(json.dumps(cfgs["wp50"])),
                        config_wp400=np.array(json.dumps(cfgs["wp400"])),
                        meta=_meta({"reward_terms": ["vtarg_tracking"] + TERMS[1:]}), **out)


def gen_g11():
    """Rocket6DOF built from env_configs other than config.yaml."""
    import copy

    Sim, Env, cfg = load_reference()
    sim_module = sys.modules["my_environment.utils.simulator"]
    spy = _IvpSpy(sim_module)
    variants = {}
    a = copy.deepcopy(cfg)
    a["timestep"] = 0.05
    a["IC"] = [900.0, 400.0, -60.0, -60.0, -40.0, 8.0, 0.9239, 0.0, 0.3827, 0.0, 0.0, 0.0, 0.0, 36.0e3]
    a["ICRange"] = [100.0, 100.0, 40.0, 20.0, 20.0, 6.0, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 4.0e3]
    a["reward_coeff"].update(alfa=-0.03, beta=-3.0e-8, eta=0.05, gamma=-4.0, kappa=25.0, w_r_f=0.2, w_v_f=1.5, max_r_f=60.0, max_v_f=30.0)
    a["trajectory_limits"]["attitude_limit"] = [60, 70, 360]
    a["landing_params"].update(landing_radius=20, maximum_velocity=6, landing_attitude_limit=[6, 12, 360])
    variants["a"] = a
    b = copy.deepcopy(cfg)
    b["timestep"] = 0.2
    b["IC"] = [1500.0, -300.0, 120.0, -70.0, 30.0, -12.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 39.0e3]
    b["ICRange"] = [200.0, 300.0, 100.0, 30.0, 30.0, 10.0, 0.2, 0.2, 0.2, 0.2, 0.05, 0.05, 0.05, 2.0e3]
    b["reward_coeff"].update(alfa=-0.002, beta=-5.0e-9, eta=0.2, gamma=-20.0, kappa=5.0, w_r_f=0.0, w_v_f=0.3, max_r_f=150.0, max_v_f=80.0)
    b["trajectory_limits"]["attitude_limit"] = [45, 45, 360]
    b["landing_params"].update(landing_radius=50, maximum_velocity=15, landing_attitude_limit=[20, 20, 360])
    variants["b"] = b
    out = {}
    for tag, vcfg in variants.items():
        env = Env(**vcfg)
        rng = np.random.RandomState(1100 + ord(tag))
        eps, x0s = [], []
        for k in range(3):       # flights from this config's own IC distribution
            acts = (0.6 * rng.uniform(-1, 1, (400, 3)) + np.array([0.0, 0.0, 0.2 * k])).astype(np.float32)
            x0, recs = _run_episode(env, spy, acts, max_steps=120)
            eps.append(recs)
            x0s.append(x0)
        for k in range(30):      # low-altitude descents: events, goal conditions with the changed landing limits
            icv = np.zeros(14)
            icv[0] = rng.uniform(0.4, 40.0)
            icv[1:3] = rng.uniform(-45, 45, 2)
            icv[3] = rng.uniform(-25, -1.5)
            icv[4:6] = rng.uniform(-6, 6, 2)
            icv[6:10] = _quat_from_axis_angle([0.0, rng.normal(), rng.normal() + 1e-9], rng.uniform(0, 0.45))
            icv[10:13] = rng.uniform(-0.3, 0.3, 3)
            icv[13] = rng.uniform(30e3, 41e3)
            acts = rng.uniform(-1, 1, (200, 3)).astype(np.float32)
            acts[:, 2] = rng.uniform(-0.7, 0.3, 200)
            x0, recs = _run_episode(env, spy, acts, ic=icv)
            eps.append(recs)
            x0s.append(x0)
        g = _pack(eps)
        g["x0"] = np.array(x0s)
        g["normalizer"] = np.asarray(env.state_normalizer, float)
        g["bounds_low"] = np.asarray(env.position_bounds_space.low, float)
        g["bounds_high"] = np.asarray(env.position_bounds_space.high, float)
        for k_, v in g.items():
            out[f"{tag}_{k_}"] = v
        out[f"config_{tag}"] = np.array(json.dumps(vcfg))
        print("G11", tag, "dt", vcfg["timestep"], "steps", len(g["reward"]), "ground events", int((g["status"] == 1).sum()), "done", int(g["done"].sum()),
              "goal", int((g["terms"][:, 4] != 0).sum()), "attitude penalty", int((g["terms"][:, 3] != 0).sum()), "bounds", int(g["bounds"].sum()))
    np.savez_compressed(os.path.join(HERE, "g11_other_configs.npz"), meta=_meta(), **out)


def gen_g12(n=50000):
    Sim, Env, cfg = load_reference()
    env = Env(**cfg)
    env.seed(2022)
    out = np.zeros((n, 14), np.float32)
    obs = np.zeros((n, 14), np.float32)
    for i in range(n):
        obs[i] = env.reset()                                   # rocket_env.py:180-199
        out[i] = np.asarray(env.SIM.state, np.float64)          # the fresh simulator's state
    np.savez_compressed(os.path.join(HERE, "g12_reset_draws.npz"), state=out, obs_first=obs[:256], config=np.array(json.dumps(cfg)),
                        meta=_meta({"seed": 2022, "n": n}))
    print("G12", out.shape, "mean", out.mean(0).round(3), "min", out.min(0).round(3), "max", out.max(0).round(3))


def main():
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g12":
        gen_g12()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g11":
        gen_g11()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g10":
        gen_g10()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g9":
        gen_g9()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g7":
        gen_g7()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "g8":
        gen_g8()
        return
    Sim, Env, cfg = load_reference()
    sim_module = sys.modules["my_environment.utils.simulator"]
    spy = _IvpSpy(sim_module)
    cfg_json = json.dumps(cfg)

    # ------------------------------------------------------------------ G1
    ic = np.array([100, 100, 100, 0, 3, 4, 1, 1, 0, 0, 0, 0, 0, 50e3], dtype=np.float64)
    s = Sim(ic.copy(), 0.5)  # default dt of Simulator6DOF (simulator.py:9)
    out, status = s.step([0, 0, 1])
    np.savez_compressed(
        os.path.join(HERE, "g1_sim_step.npz"),
        ic=ic, u=np.array([0.0, 0.0, 1.0]), dt=0.5, state=np.asarray(out, float), status=status,
        nfev=spy.last.nfev, meta=_meta(),
    )

    # ------------------------------------------------------------------ G2
    rng = np.random.RandomState(20240101)
    n = 512
    y = np.zeros((n, 14))
    y[:, 0] = rng.uniform(-30, 2200, n)
    y[:, 1:3] = rng.uniform(-2000, 2000, (n, 2))
    y[:, 3:6] = rng.uniform(-300, 300, (n, 3))
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q *= rng.uniform(0.8, 1.2, (n, 1))  # deliberately un-normalised
    y[:, 6:10] = q
    y[:, 10:13] = rng.uniform(-1, 1, (n, 3))
    y[:, 13] = rng.uniform(30e3, 42e3, n)
    y[:8, 3:6] = 0.0  # zero-velocity rows (aero force vanishes)
    m0 = rng.uniform(40e3, 42e3, n)
    u = np.zeros((n, 3), dtype=np.float32)
    u[:, 0:2] = rng.uniform(-0.35, 0.35, (n, 2))
    u[:, 2] = rng.uniform(0, 981e3, n)
    dy = np.zeros((n, 14))
    for i in range(n):
        ici = y[i].copy()
        ici[13] = m0[i]
        sim = Sim(ici, 0.1)
        dy[i] = sim.RHS(0.0, y[i].copy(), u[i])  # u as float32 array, like the env passes it
    np.savez_compressed(os.path.join(HERE, "g2_rhs.npz"), y=y, u=u, m0=m0, dy=dy, meta=_meta())

    # ------------------------------------------------------------------ G3
    env = Env(**cfg)
    eps = []
    x0s = []
    for k in range(10):
        acts = np.random.RandomState(k).uniform(-1, 1, (1000, 3)).astype(np.float32)
        if k >= 7:  # smoother, biased actions: longer flights, different attitude history
            acts = (0.25 * acts + np.array([0.0, 0.0, 0.15 * (k - 7)])).astype(np.float32)
        x0, recs = _run_episode(env, spy, acts)
        eps.append(recs)
        x0s.append(x0)
    g3 = _pack(eps)
    np.savez_compressed(
        os.path.join(HERE, "g3_env_rollouts.npz"), x0=np.array(x0s), config=np.array(cfg_json),
        normalizer=np.asarray(env.state_normalizer, float),
        bounds_low=np.asarray(env.position_bounds_space.low, float),
        bounds_high=np.asarray(env.position_bounds_space.high, float),
        meta=_meta({"dt": cfg["timestep"]}), **g3,
    )

    # ------------------------------------------------------------------ G4
    rng = np.random.RandomState(4)
    eps = []
    x0s = []
    for k in range(96):
        icv = np.zeros(14)
        soft = k % 3 == 0  # soft, near-pad touch-downs: exercise goal / final_velocity terms
        icv[0] = rng.uniform(0.3, 6.0) if soft else rng.uniform(0.5, 60.0)
        icv[1:3] = rng.uniform(-20, 20, 2) if soft else rng.uniform(-150, 150, 2)
        icv[3] = rng.uniform(-9, -1) if soft else rng.uniform(-90, -2)
        icv[4:6] = rng.uniform(-2, 2, 2) if soft else rng.uniform(-40, 40, 2)
        tilt = rng.uniform(0, 0.15) if soft else rng.uniform(0, 1.4)
        axis = np.array([0.0, *rng.normal(size=2)])
        qq = _quat_from_axis_angle(axis + 1e-9, tilt)
        # compose with a roll about body x
        roll = rng.uniform(-np.pi, np.pi)
        qr = np.array([np.cos(roll / 2), np.sin(roll / 2), 0, 0])
        w0, v0 = qq[0], qq[1:]
        w1, v1 = qr[0], qr[1:]
        icv[6] = w0 * w1 - v0 @ v1
        icv[7:10] = w0 * v1 + w1 * v0 + np.cross(v0, v1)
        icv[10:13] = rng.uniform(-0.1, 0.1, 3) if soft else rng.uniform(-0.6, 0.6, 3)
        icv[13] = rng.uniform(30e3, 41.5e3)
        acts = rng.uniform(-1, 1, (400, 3)).astype(np.float32)
        if soft:
            acts[:, 2] = rng.uniform(-0.6, 0.1, 400)
        x0, recs = _run_episode(env, spy, acts, ic=icv)
        eps.append(recs)
        x0s.append(x0)
    g4 = _pack(eps)
    np.savez_compressed(
        os.path.join(HERE, "g4_ground_events.npz"), x0=np.array(x0s), config=np.array(cfg_json),
        meta=_meta({"dt": cfg["timestep"]}), **g4,
    )

    # ------------------------------------------------------------------ G5
    from scipy.spatial.transform import Rotation as R

    rng = np.random.RandomState(5)
    cases = []

    def add_case(st, u_norm):
        cases.append((np.asarray(st, float), np.asarray(u_norm, np.float32)))

    def rand_state():
        st = np.zeros(14)
        st[0] = rng.uniform(-40, 2300)
        st[1] = rng.uniform(-2300, 2300)
        st[2] = rng.uniform(-200, 200)
        st[3:6] = rng.uniform(-250, 250, 3)
        qv = rng.normal(size=4)
        st[6:10] = qv / np.linalg.norm(qv)
        st[10:13] = rng.uniform(-1.5, 1.5, 3)
        st[13] = rng.uniform(25e3, 42e3)
        return st

    for _ in range(600):
        add_case(rand_state(), rng.uniform(-1, 1, 3))
    # multi-positive-root quartics: (near) head-on approach with |r| g / |v|^2 < 1/6
    for _ in range(200):
        st = rand_state()
        w = rng.uniform(30, 250)
        sdist = rng.uniform(0.02, 0.16) * w * w / 9.81
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        d[0] = abs(d[0])
        pert = rng.normal(size=3) * 10 ** rng.uniform(-5, -1.3)
        vdir = -d + pert
        vdir /= np.linalg.norm(vdir)
        st[0:3] = d * sdist
        st[3:6] = vdir * w
        add_case(st, rng.uniform(-1, 1, 3))
    # attitude just inside / outside 85 deg (pitch and yaw), and around the 10 deg landing limit
    for ang_deg in [84.0, 84.9, 84.999, 85.001, 85.1, 86.0, 89.9, 90.0, 95.0, 9.9, 10.1, 170.0, 179.9]:
        for ax in ([0, 1, 0], [0, 0, 1], [0, 1, 1], [1, 0, 0]):
            for sgn in (1, -1):
                st = rand_state()
                st[6:10] = _quat_from_axis_angle(ax, sgn * np.deg2rad(ang_deg))
                add_case(st, rng.uniform(-1, 1, 3))
    # landing thresholds: |r| ~ 30 / 100, |v| ~ 10 / 50, x ~ 1e-3, |omega_i| ~ 0.2
    for rr in [29.0, 29.99, 30.01, 31, 99.0, 99.99, 100.01, 101]:
        for vv in [9.0, 9.99, 10.01, 11, 49.0, 49.99, 50.01, 51]:
            for xx in [-1e-2, 0.0, 5e-16, 9e-4, 1.1e-3, 0.5]:
                st = rand_state()
                yz = rng.normal(size=2)
                yz = yz / np.linalg.norm(yz) * np.sqrt(max(rr * rr - xx * xx, 0))
                st[0] = xx
                st[1:3] = yz
                vd = rng.normal(size=3)
                st[3:6] = vd / np.linalg.norm(vd) * vv
                st[10:13] = rng.choice([0.05, 0.19, 0.21, 0.5], 3) * rng.choice([-1, 1], 3)
                tilt = rng.choice([0.02, 0.15, 0.2, 1.0])
                st[6:10] = _quat_from_axis_angle([0, rng.normal(), rng.normal() + 1e-9], tilt)
                add_case(st, rng.uniform(-1, 1, 3))
    # bounds faces (float32 compares, inclusive)
    lo = np.asarray(env.position_bounds_space.low, float)
    hi = np.asarray(env.position_bounds_space.high, float)
    for ax in range(3):
        for face in (lo, hi):
            for eps_ in [-1.0, -1e-3, 0.0, 1e-3, 1.0]:
                st = rand_state()
                st[0] = rng.uniform(100, 2000)
                st[1] = rng.uniform(-2000, 2000)
                st[2] = rng.uniform(-150, 150)
                st[ax] = face[ax] + eps_
                add_case(st, rng.uniform(-1, 1, 3))
    # mass-starved / saturation of a_targ (|a| vs T_max/m)
    for _ in range(60):
        st = rand_state()
        st[0:3] = rng.uniform(1, 60, 3)
        st[3:6] = rng.uniform(-3, 3, 3)
        add_case(st, rng.uniform(-1, 1, 3))

    env.reset()
    n = len(cases)
    S = np.zeros((n, 14))
    A = np.zeros((n, 3), np.float32)
    REW = np.zeros(n)
    TRM = np.zeros((n, 7))
    BND = np.zeros(n, np.uint8)
    ATG = np.zeros((n, 3))
    for i, (st, un) in enumerate(cases):
        u_phys = env._denormalize_action(un)
        # place the simulator in the crafted post-step situation (simulator.py:177-186 reads
        # states[-1] / actions[-1]); rocket_env.py:206-215 order of operations
        env.SIM.states.append(st.copy())
        env.SIM.actions.append(u_phys)
        env.SIM.state = st.copy()
        env.state = st.copy()
        st32 = st.astype(np.float32)
        env.rotation_obj = R.from_quat(np.roll(st32[6:10], -1))
        rew, rd = env._compute_reward(st32, u_phys)
        S[i] = st
        A[i] = un
        REW[i] = rew
        TRM[i] = [float(rd[t]) for t in TERMS]
        BND[i] = env._check_bounds_violation(st32)
        ATG[i] = env.get_atarg()
    np.savez_compressed(
        os.path.join(HERE, "g5_reward_cases.npz"), state=S, action=A, reward_sum=REW, terms=TRM,
        bounds=BND, atarg=ATG, config=np.array(cfg_json), meta=_meta(),
    )

    # ------------------------------------------------------------------ G6
    def rhs_of(sim):
        return lambda yy, uu: np.asarray(sim.RHS(0.0, yy, uu), float)

    def euler(f, y0, uu, h):
        return y0 + h * f(y0, uu)

    def rk4(f, y0, uu, h):
        k1 = f(y0, uu)
        k2 = f(y0 + 0.5 * h * k1, uu)
        k3 = f(y0 + 0.5 * h * k2, uu)
        k4 = f(y0 + h * k3, uu)
        return y0 + h / 6.0 * (k1 + 2 * k2 + 2 * k3 + k4)

    # single steps from states visited by the reference env (G3 pre-states), f32 physical actions
    idx = np.random.RandomState(6).choice(len(g3["pre"]), 256, replace=False)
    pre = g3["pre"][idx]
    m0s = g3["m0"][idx]
    an = g3["action"][idx]
    single = {k: np.zeros((len(idx), 14)) for k in ("euler_0.04", "rk4_0.04", "rk4_0.1", "euler_0.1")}
    for i in range(len(idx)):
        ici = pre[i].copy()
        ici[13] = m0s[i]
        sim = Sim(ici, 0.1)
        f = rhs_of(sim)
        uu = env._denormalize_action(an[i])
        single["euler_0.04"][i] = euler(f, pre[i], uu, 0.04)
        single["euler_0.1"][i] = euler(f, pre[i], uu, 0.1)
        single["rk4_0.04"][i] = rk4(f, pre[i], uu, 0.04)
        single["rk4_0.1"][i] = rk4(f, pre[i], uu, 0.1)

    # free-running rollouts (quaternion renormalised after each step like simulator.py:97;
    # no event handling: truncated before the first x <= 0)
    roll = {}
    for name, stepper, h in (("euler_0.04", euler, 0.04), ("rk4_0.04", rk4, 0.04), ("rk4_0.1", rk4, 0.1)):
        for r_i in range(2):
            env.seed(100 + r_i)
            env.reset()
            y0 = np.array(env.SIM.state, float)
            sim = Sim(y0.copy(), h)
            f = rhs_of(sim)
            acts = np.random.RandomState(200 + r_i).uniform(-1, 1, (1000, 3)).astype(np.float32)
            ys = [y0]
            for k in range(1000):
                yn = stepper(f, ys[-1], env._denormalize_action(acts[k]), h)
                yn[6:10] /= np.linalg.norm(yn[6:10])
                if yn[0] <= 0 or not np.all(np.isfinite(yn)):
                    break
                ys.append(yn)
            ys = np.array(ys)
            roll[f"roll_{name}_{r_i}_states"] = ys
            roll[f"roll_{name}_{r_i}_actions"] = acts[: len(ys) - 1]
    np.savez_compressed(
        os.path.join(HERE, "g6_fixed_step.npz"), pre=pre, m0=m0s, action=an,
        **{f"single_{k}": v for k, v in single.items()}, **roll, config=np.array(cfg_json), meta=_meta(),
    )

    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print(f"{fn:28s} {os.path.getsize(os.path.join(HERE, fn)) / 1024:8.1f} KiB")
    d = np.load(os.path.join(HERE, "g3_env_rollouts.npz"))
    print("G3 steps", len(d["reward"]), "dones", int(d["done"].sum()), "status1", int((d["status"] == 1).sum()),
          "bounds", int(d["bounds"].sum()), "mean nfev", d["nfev"].mean())
    d = np.load(os.path.join(HERE, "g4_ground_events.npz"))
    print("G4 steps", len(d["reward"]), "dones", int(d["done"].sum()), "status1", int((d["status"] == 1).sum()),
          "goal>0", int((d["terms"][:, 4] > 0).sum()), "final_v>0", int((d["terms"][:, 6] > 0).sum()))
    d = np.load(os.path.join(HERE, "g5_reward_cases.npz"))
    print("G5 cases", len(d["reward_sum"]), "att", int((d["terms"][:, 3] != 0).sum()), "goal", int((d["terms"][:, 4] != 0).sum()),
          "bounds", int(d["bounds"].sum()))


if __name__ == "__main__":
    main()
