"""CPU-only checks of the drop-in boundary: the C-ABI library loads (dlopen needs no GPU) and exports
every symbol include/r6dof.h declares; the ctypes mirrors match the header; the host-side digestion of
the reference's env_config reproduces the reference's derived quantities.  No compute calls."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "r6dof.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(r6_[a-z_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from rl_rocket_6dof_amd import _lib

    names = _declared_functions()
    assert set(names) == set(_lib.SYMBOLS), (names, _lib.SYMBOLS)
    assert os.path.exists(_lib.lib_path()), "run `python -m rl_rocket_6dof_amd.build` first"
    lib = C.CDLL(_lib.lib_path())
    for n in names:
        assert hasattr(lib, n), f"libr6dof.so does not export {n}"
    # ... and nothing named r6_* beyond them: the cross-unit helpers (r6_internal_*) are hidden, the test hooks live in
    # csrc/variants/libr6dof_testhooks.so only (round-4 review, Weak 9)
    import shutil
    import subprocess

    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("r6_")})
    assert exported == names, f"exports not declared in include/r6dof.h: {sorted(set(exported) - set(names))}"


def test_ctypes_mirrors_match_header_and_library():
    from rl_rocket_6dof_amd import _lib
    from rl_rocket_6dof_amd.params import NUM_STATS, NUM_TERMS, STATE_DIM, MAX_BLOCKS, R6Buffers, R6Params, R6StepK

    lib = _lib.load()
    assert lib.r6_abi_version() == _lib.ABI_VERSION
    assert lib.r6_params_size() == C.sizeof(R6Params)
    assert lib.r6_buffers_size() == C.sizeof(R6Buffers)
    assert lib.r6_stepk_size() == C.sizeof(R6StepK)
    src = open(HEADER).read()
    consts = dict(re.findall(r"#define\s+(R6_[A-Z_]+)\s+(\d+)", src))
    assert int(consts["R6_STATE_DIM"]) == STATE_DIM and int(consts["R6_NUM_TERMS"]) == NUM_TERMS
    assert int(consts["R6_NUM_STATS"]) == NUM_STATS and int(consts["R6_MAX_BLOCKS"]) == MAX_BLOCKS
    # field order of the structs = order in the header
    body = re.search(r"typedef struct R6Params \{(.*?)\} R6Params;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hdr_fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for name in decl.split(None, 1)[1].split(","):
            hdr_fields.append(re.sub(r"\[.*\]", "", name).strip())
    assert hdr_fields == [f[0] for f in R6Params._fields_]
    body = re.search(r"typedef struct R6StepK \{(.*?)\} R6StepK;", src, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    assert [d.split()[-1] for d in body.split(";") if d.strip()] == [f[0] for f in R6StepK._fields_]


def test_bad_arguments_are_rejected_without_a_gpu():
    from rl_rocket_6dof_amd import _lib
    from rl_rocket_6dof_amd.params import R6Buffers, default_env_config, make_params

    lib = _lib.load()
    p, _ = make_params(default_env_config())
    b = R6Buffers()
    b.n, b.ld = 8, 4                                   # ld < n
    assert lib.r6_step(C.byref(p), C.byref(b), 0, 1, 0, None) == -10001
    assert b"ld" in lib.r6_last_error()
    b.n, b.ld = 0, 0                                   # empty shard: nothing to do, no launch
    assert lib.r6_step(C.byref(p), C.byref(b), 0, 1, 0, None) == 0
    b.n, b.ld = 8, 64                                  # required pointers missing
    assert lib.r6_step(C.byref(p), C.byref(b), 0, 1, 0, None) == -10001
    p.obs_dim = 12
    assert lib.r6_step(C.byref(p), C.byref(b), 0, 1, 0, None) == -10001
    p.obs_dim = 14
    from rl_rocket_6dof_amd.params import R6StepK

    assert lib.r6_step_k(C.byref(p), C.byref(b), None, 0, 1, 0, None) == -10001          # no R6StepK
    assert lib.r6_step_k(C.byref(p), C.byref(b), C.byref(R6StepK(k=0)), 0, 1, 0, None) == -10001
    assert b"k must be" in lib.r6_last_error()
    # the PPO-side entry points validate before they touch the device as well
    from rl_rocket_6dof_amd.params import R6PolicyGrad, R6PolicyNet

    assert lib.r6_gae(None, None, None, None, 4, 8, 0.99, 0.95, None, 4, None, None, None, None) == -10001
    assert lib.r6_permutation(None, 16, 1, 1, None, None) == -10001
    assert lib.r6_permutation(C.c_void_p(64), 0, 1, 1, None, None) == 0                  # empty: nothing to do, no launch
    assert lib.r6_minibatch_adv_norm(C.c_void_p(64), None, 16, 0, C.c_void_p(64), C.c_void_p(64), None) == -10001   # n_batches = 0
    assert lib.r6_adam_clip_step(None, None, None, None, 8, 0.5, 1e-3, None, 0.9, 0.999, 1e-5, None, 0, 0, 0.0, None) == -10001
    net, grad = R6PolicyNet(), R6PolicyGrad()
    net.obs_dim, net.hidden1, net.hidden2, net.shared = 13, 256, 64, 1                    # unsupported topology
    fake = C.c_void_p(64)
    assert lib.r6_ppo_minibatch_grad(C.byref(net), fake, fake, fake, fake, fake, None, 32, 32, None, 0.2, 0.5, 0.0, C.byref(grad), fake, fake,
                                     1 << 20, None) == -10001
    assert b"supported nets" in lib.r6_last_error()
    # a rollout buffer of 2^31 rows or more cannot be indexed (32-bit row numbers in the kernel): rejected, not mis-gathered
    assert lib.r6_ppo_minibatch_grad(C.byref(net), fake, fake, fake, fake, fake, fake, 32, 1 << 31, None, 0.2, 0.5, 0.0, C.byref(grad), fake,
                                     fake, 1 << 20, None) == -10001
    assert b"n_rows" in lib.r6_last_error()
    assert lib.r6_policy_act(C.byref(net), None, 32, None, None, None, None, 0, 1, None, 0, 0, None, 0, None) == -10001


def test_derived_quantities_match_reference():
    from rl_rocket_6dof_amd.params import derive_spec, make_params

    g3 = load_golden("g3_env_rollouts.npz")
    cfg = json.loads(str(g3["config"]))
    spec = derive_spec(cfg)
    np.testing.assert_allclose(spec.state_normalizer, g3["normalizer"], rtol=1e-12)      # rocket_env.py:106-126
    np.testing.assert_array_equal(spec.bounds_low.astype(np.float64), g3["bounds_low"])   # rocket_env.py:129-134
    np.testing.assert_array_equal(spec.bounds_high.astype(np.float64), g3["bounds_high"])
    p, _ = make_params(cfg, remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500)
    assert p.obs_dim == 13 and p.clip_reward == 1 and p.max_episode_steps == 1500
    assert p.n_substeps == 1 and abs(p.dt - 0.1) < 1e-7
    np.testing.assert_allclose(list(p.inv_norm), 1.0 / g3["normalizer"], rtol=1e-6)
    # attitude limits [85, 85, 360] deg and landing [10, 10, 360] deg in the kernel's trig-free form
    assert abs(p.traj_c0 - np.cos(np.deg2rad(85))) < 1e-7 and abs(p.traj_s1 - np.sin(np.deg2rad(85))) < 1e-7
    assert p.traj_c2 == -2.0 and p.land_c2 == -2.0
    assert abs(p.land_c0 - np.cos(np.deg2rad(10))) < 1e-7
    lo = np.float32(cfg["IC"]) - np.float32(cfg["ICRange"]) / np.float32(2)
    np.testing.assert_array_equal(np.float32(list(p.ic_lo)), lo)


def test_reference_default_kwargs_raise_like_the_reference():
    # the reference's own class defaults lack landing_params["waypoint"] -> KeyError (rocket_env.py:168)
    from rl_rocket_6dof_amd.params import default_env_config, derive_spec

    cfg = default_env_config()
    del cfg["landing_params"]["waypoint"]
    with pytest.raises(KeyError):
        derive_spec(cfg)


def test_auto_substeps_rule():
    from rl_rocket_6dof_amd.params import auto_substeps

    assert auto_substeps(0.04) == 1 and auto_substeps(0.1) == 1 and auto_substeps(0.5) == 5 and auto_substeps(0.25) == 3


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from rl_rocket_6dof_amd import BatchedRocket6DOF

    with pytest.raises(RuntimeError, match="no CPU implementation"):
        BatchedRocket6DOF(4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rl_rocket_6dof_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "oracle_c" not in txt and "liboracle" not in txt, f


def test_builtin_defaults_equal_reference_config_and_yaml_loader(tmp_path):
    """The built-in defaults are the reference's config.yaml values (as recorded in the golden fixture), and a
    YAML file with the reference's layout loads through load_config (main_6DOF.py:18-27)."""
    import yaml

    from rl_rocket_6dof_amd import default_env_config, load_config

    ref = json.loads(str(load_golden("g3_env_rollouts.npz")["config"]))

    def norm(d):
        if isinstance(d, dict):
            return {k: norm(v) for k, v in sorted(d.items())}
        if isinstance(d, list):
            return [float(x) for x in d]
        return float(d) if isinstance(d, (int, float)) else d

    assert norm(default_env_config()) == norm(ref)
    f = tmp_path / "cfg.yaml"
    f.write_text(yaml.safe_dump({"env_config": ref, "sb3_config": {"max_time": 150, "total_timesteps": 1e6}}))
    sb3, env = load_config(str(f))
    assert norm(env) == norm(ref) and sb3["max_time"] == 150


def test_env_group_bounds_are_tile_aligned_and_cover_the_batch():
    """BatchedRocket6DOF(groups=G) (round 5): contiguous env groups cut at multiples of 64 -- tile-aligned rows, 16-byte aligned
    action (12 B per env) and observation (52 / 56 B per env) blocks -- covering [0, n) without gaps; a split that would leave a group
    with less than two tiles is refused.  (The launches themselves: tests/test_gpu_fullsize.py.)"""
    from rl_rocket_6dof_amd.batched_env import group_bounds

    for n in (128, 129, 4096, 200_003, 1 << 20, (1 << 20) + 77, 3 * (1 << 19) + 5):
        for g in (1, 2, 3, 4, 7):
            if g > 1 and n < 128 * g:
                with pytest.raises(ValueError):
                    group_bounds(n, g)
                continue
            b = group_bounds(n, g)
            assert b[0] == 0 and b[-1] == n and len(b) == g + 1
            assert all(x % 64 == 0 for x in b[:-1]) and all(b[i] < b[i + 1] for i in range(g))
            assert all((x * 12) % 16 == 0 and (x * 52) % 16 == 0 and (x * 56) % 16 == 0 for x in b[:-1])
            assert max(b[i + 1] - b[i] for i in range(g)) - min(b[i + 1] - b[i] for i in range(g)) <= 128
    with pytest.raises(ValueError):
        group_bounds(1000, 0)
