"""r6_step_k -- K env steps per launch -- against K single r6_step launches: BIT FOR BIT (state, observations, rewards,
done masks, flags, terminal observations, episode records, the in-kernel reset stream), at n in {1, 63, 65 536, 2^20},
for the raw env, the fused make_env() stack, velocity shaping, Euler / RK4 / RK45 and sub-stepped RK4.
The reference loop this replaces: main_6DOF.py:136-139 / montecarlo_script.py:33-51 (one env.step per timestep)."""
import json

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg():
    return json.loads(str(load_golden("g3_env_rollouts.npz")["config"]))


def _pair(n, cfg, seed=11, **kw):
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    a = BatchedRocket6DOF(n, cfg, seed=seed, record_terms=n <= 4096, record_nfev=n <= 4096, **kw)
    b = BatchedRocket6DOF(n, cfg, seed=seed, record_terms=n <= 4096, record_nfev=n <= 4096, **kw)
    return a, b


def _seed_low_altitude(env_a, env_b):
    """Part of a small batch starts a few metres above the pad (golden G4's recorded pre-states), so that ground events,
    bounds violations and goal rewards -- and with them in-kernel resets -- fall inside a K of a few steps."""
    G4 = load_golden("g4_ground_events.npz")
    st = env_a.get_state().cpu().numpy()
    k = min(env_a.num_envs // 2, len(G4["pre"]))
    if k:
        st[:k] = G4["pre"][len(G4["pre"]) - k:].astype(np.float32)
    env_a.set_state(st)
    env_b.set_state(st)


def _compare(torch, a, b, acts, what):
    K = acts.shape[0]
    res = b.step_k(acts)
    n_done = 0
    for k in range(K):
        o, r, d, _ = a.step(acts[k])
        assert torch.equal(o, res["obs"][k]), f"{what}: obs of step {k}"
        assert torch.equal(r, res["reward"][k]), f"{what}: reward of step {k}"
        assert torch.equal(d, res["done"][k]), f"{what}: done of step {k}"
        if a.flags is not None:
            assert torch.equal(a.flags, res["flags"][k]), f"{what}: flags of step {k}"
        if a.terms is not None:
            assert torch.equal(a.terms[:, : a.num_envs], res["terms"][k][:, : a.num_envs]), f"{what}: reward terms of step {k}"
        if a.nfev is not None:
            assert torch.equal(a.nfev, res["nfev"][k]), f"{what}: nfev of step {k}"
        if bool(d.any()):
            idx = d.nonzero(as_tuple=True)[0]
            n_done += len(idx)
            assert torch.equal(a.final_obs[idx], res["final_obs"][k][idx]), f"{what}: terminal observation, step {k}"
            if a.fin_return is not None:
                assert torch.equal(a.fin_return[idx], res["fin_return"][k][idx]) and torch.equal(a.fin_length[idx], res["fin_length"][k][idx])
    assert torch.equal(a.state, b.state), f"{what}: state after {K} steps"
    assert torch.equal(a.epconst, b.epconst), f"{what}: m0 / v0 after {K} steps (in-kernel resets)"
    if a.ep_step is not None:
        assert torch.equal(a.ep_step, b.ep_step) and torch.equal(a.ep_return, b.ep_return), f"{what}: episode counters"
    assert a._step_count == b._step_count
    return n_done


CASES = [
    ("n1_rk4", 1, 9, dict()),
    ("n63_rk4", 63, 12, dict()),
    ("n63_euler_dt004", 63, 12, dict(integrator="euler", timestep=0.04)),
    ("n200_rk4_substeps3", 200, 6, dict(n_substeps=3)),
    ("n130_rk45", 130, 5, dict(integrator="rk45")),
    ("n257_wrapped", 257, 40, dict(remove_mass=True, clip_reward=(-1, 100), max_episode_steps=16)),
]


@pytest.mark.parametrize("name,n,K,kw", CASES, ids=[c[0] for c in CASES])
def test_k_fused_steps_equal_k_launches_small(cfg, name, n, K, kw):
    import torch

    a, b = _pair(n, cfg, **kw)
    a.reset()
    b.reset()
    if n >= 60:
        _seed_low_altitude(a, b)
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = (torch.rand((K, n, 3), generator=g) * 2 - 1).to(a.device)
    n_done = _compare(torch, a, b, acts, name)
    if n >= 60:
        assert n_done > 0, "the case must exercise in-kernel resets"
    sa, sb = a.episode_stats(), b.episode_stats()
    for key in sa:
        assert abs(sa[key] - sb[key]) <= 1e-6 * max(1.0, abs(sa[key])), (key, sa, sb)
    # a second call continues where the first one stopped (global step counter, reset stream)
    acts2 = (torch.rand((3, n, 3), generator=g) * 2 - 1).to(a.device)
    _compare(torch, a, b, acts2, name + " (second call)")


def test_velocity_shaping_k_steps(cfg):
    import torch

    vcfg = json.loads(json.dumps(cfg))
    vcfg["reward_shaping_type"] = "velocity"
    vcfg["landing_params"]["waypoint"] = 50
    a, b = _pair(300, vcfg)
    a.reset()
    b.reset()
    _seed_low_altitude(a, b)
    acts = (torch.rand((10, 300, 3), generator=torch.Generator().manual_seed(1)) * 2 - 1).to(a.device)
    assert _compare(torch, a, b, acts, "velocity shaping") > 0


@pytest.mark.parametrize("n,K,kw", [(65536, 16, dict(integrator="euler", timestep=0.04)), (1 << 20, 6, dict())],
                         ids=["configs1_65536_euler_dt004", "configs2_2p20_rk4"])
def test_k_fused_steps_equal_k_launches_full_size(cfg, n, K, kw):
    """BASELINE configs[1] (65 536 envs, Euler dt 0.04) and configs[2] (2^20 envs, RK4): burn-in with
    last_only launches on one env and single steps on the other (must leave identical states), then K recorded steps."""
    import torch

    a, b = _pair(n, cfg, **kw)
    a.reset()
    b.reset()
    g = torch.Generator(device=a.device).manual_seed(3)
    burn = torch.rand((25, n, 3), generator=g, device=a.device) * 2 - 1
    for rep in range(12):     # 300 steps: past the first landings, so the K window holds resets
        b.step_k(burn, last_only=True)
        for k in range(burn.shape[0]):
            a.step_async(burn[k])
    assert torch.equal(a.state, b.state) and torch.equal(a.obs, b.obs) and torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done)
    acts = torch.rand((K, n, 3), generator=g, device=a.device) * 2 - 1
    n_done = _compare(torch, a, b, acts, f"n={n}")
    assert n_done > n // 1000
    print(f"[step_k n={n} K={K}] bit-identical to {K} launches; {n_done} episodes ended inside the window")


def test_last_only_and_bad_arguments(cfg):
    import ctypes as C

    import torch

    from rl_rocket_6dof_amd.params import R6StepK

    a, b = _pair(100, cfg)
    a.reset()
    b.reset()
    acts = (torch.rand((7, 100, 3), generator=torch.Generator().manual_seed(2)) * 2 - 1).to(a.device)
    for k in range(7):
        o, r, d, _ = a.step(acts[k])
    out = b.step_k(acts, last_only=True)
    assert out["obs"] is b.obs and torch.equal(b.obs, o) and torch.equal(b.reward, r) and torch.equal(out["done"], d)
    assert torch.equal(a.state, b.state)
    sk = R6StepK(k=0, action_stride=300)
    assert b.lib.r6_step_k(C.byref(b.params), C.byref(b._buf), C.byref(sk), 0, 1, 0, None) == -10001
    sk = R6StepK(k=4, action_stride=10)       # stride smaller than one step of actions
    assert b.lib.r6_step_k(C.byref(b.params), C.byref(b._buf), C.byref(sk), 0, 1, 0, None) == -10001
    with pytest.raises(ValueError):
        b.step_k(torch.zeros((0, 100, 3), device=b.device))


def test_a_failed_launch_internal_wait_is_reported_by_every_later_call(cfg):
    """step_k_pair_kernel's waits between the waves of a workgroup are bounded; one that ran out raises the library's device error
    word (mapped host memory).  From then on EVERY stepping entry point -- r6_step, r6_step_k, r6_rollout -- must fail with
    R6_E_DEVICE, r6_device_status / r6_last_error must say so without any further call, and only r6_clear_device_error clears it.
    The kernel side is provoked by scripts/stepk_abort_check.py (an 8-poll build); here the word is raised through the hook of the
    -DR6_TEST_HOOKS build (csrc/variants/libr6dof_testhooks.so: the release library does not export the hook)."""
    import ctypes as C
    import os

    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, _lib, build

    release = _lib.load()
    assert not hasattr(release, "r6_internal_test_pair_abort"), "the release library must not export the test hook"
    if not os.path.exists(build.TEST_HOOKS_LIB):
        pytest.skip("libr6dof_testhooks.so not built (python -m rl_rocket_6dof_amd.build --test-hooks)")
    env = BatchedRocket6DOF(4096, cfg, integrator="euler", timestep=0.04, seed=5, lib_path=build.TEST_HOOKS_LIB)
    lib = env.lib
    env.reset()                               # the first eager call creates the word
    acts = (torch.rand((4, 4096, 3), generator=torch.Generator().manual_seed(3)) * 2 - 1).to(env.device)
    env.step_k(acts)
    torch.cuda.synchronize()
    assert _lib.device_status(lib) == 0
    hook = lib.r6_internal_test_pair_abort
    hook.argtypes, hook.restype = [C.c_int], C.c_int
    assert hook(1) == 1
    try:
        assert _lib.device_status(lib) == -10003
        assert b"bounded wait" in lib.r6_last_error()          # reported without another stepping call (e.g. after the LAST launch of a run)
        with pytest.raises(_lib.R6Error, match="-10003"):
            env.step_k(acts)
        with pytest.raises(_lib.R6Error, match="-10003"):
            env.step(acts[0])
        env.reset()                                             # re-initialising the envs is how a caller recovers: allowed
        assert _lib.device_status(lib) == -10003                # ... and does not clear the word by itself
    finally:
        assert _lib.clear_device_error(lib) == -10003
    assert _lib.device_status(lib) == 0 and _lib.clear_device_error(lib) == 0
    env.step_k(acts)                          # and the library works again
    env.step(acts[0])
    torch.cuda.synchronize()


def test_one_buffer_for_all_steps_keeps_the_last_step(cfg):
    """include/r6dof.h: stride 0 = 'all steps write the same buffer, as K r6_step calls would' -- the buffer must end up holding step
    K - 1.  The three-waves-per-tile form stores the rewards of steps k and k + 1 from two different waves; with one buffer for
    all steps the launcher must therefore take the in-order kernel (round-3 advice).  Checked through the C ABI at a size where
    the pair form would otherwise be chosen, against K single steps."""
    import ctypes as C

    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF
    from rl_rocket_6dof_amd.params import R6Buffers, R6StepK

    n, K = 4096, 9
    a = BatchedRocket6DOF(n, cfg, integrator="euler", timestep=0.04, seed=21, record_terms=True)
    b = BatchedRocket6DOF(n, cfg, integrator="euler", timestep=0.04, seed=21, record_terms=True)
    a.reset()
    b.reset()
    _seed_low_altitude(a, b)
    acts = (torch.rand((K, n, 3), generator=torch.Generator().manual_seed(5)) * 2 - 1).to(a.device)
    for rep in range(20):                     # a race would not show every time
        for k in range(K):
            o, r, d, _ = a.step(acts[k])
        buf = R6Buffers.from_buffer_copy(b._buf)
        buf.action = acts.data_ptr()
        sk = R6StepK(k=K, flags=0, action_stride=n * 3)          # every output stride 0: the env's own single-step buffers
        rc = b.lib.r6_step_k(C.byref(b.params), C.byref(buf), C.byref(sk), b._seed, b._step_count + 1, b.env_offset, b._stream())
        assert rc == 0, b.lib.r6_last_error()
        b._step_count += K
        torch.cuda.synchronize()
        assert torch.equal(b.reward, r), f"rep {rep}: the reward buffer does not hold the last step"
        assert torch.equal(b.terms[:, :n], a.terms[:, :n]) and torch.equal(b.obs, o) and torch.equal(b.done, a.done)
        assert torch.equal(a.state, b.state)


def test_waves_of_a_tile_share_a_simd():
    """step_k_pair_kernel runs waves w, w + 4, w + 8, w + 12 of a 1024-thread workgroup as the four waves of ONE tile (dynamics, two
    reward waves, the output wave) and counts on the hardware placing them on the same SIMD (the dynamics wave's idle issue slots
    are what the others fill).  Results never depend
    on it (the hand-off is LDS counters), the speed does: this test watches the assumption (HW_ID bits 5:4 = simd_id on gfx9),
    with the -DR6_TEST_HOOKS build."""
    import ctypes as C
    import os

    import torch

    from rl_rocket_6dof_amd import _lib, build

    if not os.path.exists(build.TEST_HOOKS_LIB):
        pytest.skip("libr6dof_testhooks.so not built")
    lib = _lib.load(build.TEST_HOOKS_LIB)
    blocks, waves = 256, 16
    out = torch.zeros((blocks, waves, 2), dtype=torch.int32, device="cuda:0")
    fn = lib.r6_internal_test_wave_placement
    fn.argtypes, fn.restype = [C.c_void_p, C.c_int, C.c_void_p], C.c_int
    assert fn(out.data_ptr(), blocks, None) == 0
    torch.cuda.synchronize()
    hw = out[:, :, 0].cpu().numpy().astype(np.uint32)
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 15
    same = 0
    for b in range(blocks):
        assert len(set(cu[b].tolist())) == 1, "a workgroup lives on one CU"
        same += int(all(simd[b, w] == simd[b, w + 4] == simd[b, w + 8] == simd[b, w + 12] for w in range(4)))
        assert sorted(simd[b, :4].tolist()) == [0, 1, 2, 3], f"block {b}: the first four waves do not cover the four SIMDs: {simd[b].tolist()}"
    print(f"[pair placement] {same} of {blocks} workgroups have waves w, w + 4, w + 8, w + 12 on one SIMD")
    assert same == blocks, f"only {same} of {blocks} workgroups place waves w, w + 4, w + 8, w + 12 on one SIMD: the pair kernel's speed assumption no longer holds"
