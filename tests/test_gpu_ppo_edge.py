"""GPU: ppo.PPO across the shapes that pick different code paths -- env counts whose rollout rows are / are not 16-byte
aligned (direct env outputs vs copies), a batch larger than the rollout, epochs of 1 / 8 / 34 / 37 minibatch steps (graph
chunks 1, 8, 17, 1), both fused topologies and an unsupported one (torch paths), with and without a time limit."""
import math

import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(n=1000, T=5, bs=777, arch=(128, 64), shared=True, tl=50, fused=True, direct=True),
    dict(n=1001, T=5, bs=777, arch=(128, 64), shared=True, tl=50, fused=True, direct=False),
    dict(n=4096, T=8, bs=1 << 20, arch=(64, 64), shared=False, tl=0, fused=True, direct=True),
    dict(n=4096, T=8, bs=4096, arch=(64, 64), shared=False, tl=6, fused=True, direct=True),
    dict(n=2048, T=34, bs=2048, arch=(128, 64), shared=True, tl=0, fused=True, direct=True),
    dict(n=512, T=37, bs=512, arch=(128, 64), shared=True, tl=0, fused=True, direct=True),
    dict(n=4096, T=8, bs=4096, arch=(32, 32), shared=False, tl=6, fused=False, direct=None),
]


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"n{c['n']}-T{c['T']}-bs{c['bs']}-{c['arch'][0]}x{c['arch'][1]}-{'shared' if c['shared'] else 'sep'}-tl{c['tl']}")
def test_ppo_runs_on_every_path(c):
    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    kw = dict(record_flags=True, max_episode_steps=c["tl"]) if c["tl"] else {}
    env = BatchedRocket6DOF(c["n"], default_env_config(), remove_mass=True, clip_reward=(-1, 100), seed=3, **kw)
    algo = PPO(env, PPOConfig(n_steps=c["T"], batch_size=c["bs"], n_epochs=3, net_arch=c["arch"], shared_trunk=c["shared"], seed=1))
    assert (algo.fused_update is not None) == c["fused"] and (algo.fused is not None) == c["fused"]
    assert getattr(algo, "_direct", None) == c["direct"]
    before = [p.detach().clone() for p in algo.policy.parameters()]
    hist = algo.learn(6 * c["T"] * c["n"])
    assert len(hist) == 6
    assert all(math.isfinite(v) for h in hist for v in h.values())
    assert all(bool(torch.isfinite(p).all()) for p in algo.policy.parameters())
    assert any(not torch.equal(a, b) for a, b in zip(before, algo.policy.parameters()))      # it trained
    if c["fused"] and c["direct"]:
        assert algo._rollout_graph is not None                                                # and the rollouts were graph replays
