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
#
# Every call also records, per `what` (trailing indices stripped), the worst scaled error and the worst PURE relative
# error |a-b| / |b| over the entries whose reference is not tiny (|b| >= 0.05 * scale); the session writes them to
# gpurun_out/parity_errors.json and prints them, so that the asserted tolerances can be kept at a small multiple of
# what is measured (a 10x numerical regression must not pass silently).
PARITY_LOG = {}


def pure_rel_err(a, b, scale, floor=0.05):
    """Worst |a-b| / |b| over the entries whose reference is not tiny (|b| >= floor * scale)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), b.shape)
    m = np.abs(b) >= floor * scale
    return float((np.abs(a - b)[m] / np.abs(b[m])).max()) if m.any() else 0.0


def _log_key(what):
    import re

    what = re.sub(r"^step \d+: ", "", str(what))
    return re.sub(r"[\s:]*(step|env|case|episode|rollout)?\s*\d+(\s+(step|env)\s*\d+)*\s*$", "", str(what)).strip()


def assert_close_scaled(a, b, scale, rtol, what=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.broadcast_to(np.asarray(scale, dtype=np.float64), b.shape)
    diff = np.abs(a - b)
    err = diff / np.maximum(np.abs(b), scale)
    sizable = np.abs(b) >= 0.05 * scale
    pure = float((diff[sizable] / np.abs(b[sizable])).max()) if sizable.any() else 0.0
    worst = float(np.nanmax(err)) if err.size else 0.0
    rec = PARITY_LOG.setdefault(_log_key(what), {"scaled": 0.0, "pure_rel": 0.0, "rtol": rtol, "n": 0})
    rec["scaled"], rec["pure_rel"], rec["n"] = max(rec["scaled"], worst), max(rec["pure_rel"], pure), rec["n"] + int(err.size)
    rec["rtol"] = min(rec["rtol"], rtol)
    bad = ~(err <= rtol)
    if bad.any():
        i = np.unravel_index(np.nanargmax(np.where(np.isnan(err), np.inf, err)), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.size} beyond rtol={rtol:g}; worst idx={i} got={a[i]!r} want={b[i]!r} "
            f"scaled_err={err[i]:.3e}")
    return float(err.max()) if err.size else 0.0


def pytest_sessionfinish(session, exitstatus):
    if not PARITY_LOG:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "parity_errors.json"), "w") as f:
            json.dump(PARITY_LOG, f, indent=1, sort_keys=True)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter):
    if not PARITY_LOG:
        return
    terminalreporter.section("parity: worst errors measured (scaled | pure relative where |ref| >= 5 % of scale | asserted rtol)")
    for k in sorted(PARITY_LOG):
        r = PARITY_LOG[k]
        terminalreporter.write_line(f"{k[:70]:70s} {r['scaled']:.2e} | {r['pure_rel']:.2e} | {r['rtol']:.0e}  (n={r['n']})")
