import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def env_config():
    """config.yaml env_config of the reference, as recorded inside the golden fixture."""
    d = load_golden("g3_env_rollouts.npz")
    return json.loads(str(d["config"]))


@pytest.fixture(scope="session")
def g1():
    return load_golden("g1_sim_step.npz")


@pytest.fixture(scope="session")
def g2():
    return load_golden("g2_rhs.npz")


@pytest.fixture(scope="session")
def g3():
    return load_golden("g3_env_rollouts.npz")


@pytest.fixture(scope="session")
def g4():
    return load_golden("g4_ground_events.npz")


@pytest.fixture(scope="session")
def g5():
    return load_golden("g5_reward_cases.npz")


@pytest.fixture(scope="session")
def g6():
    return load_golden("g6_fixed_step.npz")


# Mixed tolerance used across parity tests: |a-b| <= RTOL * max(|b|, scale), with `scale` the
# natural magnitude of the component (the reference's own state normaliser for states).
def assert_close_scaled(a, b, scale, rtol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), b.shape)
    err = np.abs(a - b) / np.maximum(np.abs(b), scale)
    bad = ~(err <= rtol)
    if bad.any():
        i = np.unravel_index(np.nanargmax(np.where(np.isnan(err), np.inf, err)), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.size} beyond rtol={rtol:g}; worst idx={i} got={a[i]!r} want={b[i]!r} "
            f"scaled_err={err[i]:.3e}")
    return float(err.max()) if err.size else 0.0
