"""r6_step's launch shapes (include/r6dof.h r6_tune) never change a result: the XCD share, the grid size, the tile order, the
wave-priority ladder -- against the default launch, BIT FOR BIT, on the benchmark size (2^20 envs, where the persistent grid
engages), for the raw env and the fused make_env() stack, incl. the episode statistics (exact fixed-point sums: independent of
which workgroup saw an episode).  (Round 3's tail pool, fork/join split and staggered first loads were measured --
profiles/r3_experiments.json -- and removed in round 4.)  The loop being served: Rocket6DOF.step per timestep,
rocket_env.py:201-231."""
import pytest

pytestmark = pytest.mark.gpu

SHAPES = [
    {},                                                        # the default shape
    {"even_xcd_share": 0.5},
    {"even_xcd_share": 0.6},
    {"grid_blocks": 1024},
    {"grid_blocks": 768, "sweep": 2},
    {"sweep": 1},
    {"sweep_flags": 4},                                        # no priority ladder
    {"streaming_stores": 1},                                   # non-temporal output stores (the launcher's choice from 1.2 M envs on)
]
DEFAULTS = {"even_xcd_share": -1.0, "grid_blocks": 0, "sweep": -1, "sweep_flags": 0, "streaming_stores": -1, "balanced_tiles_per_wave": 0}


def _run(torch, shape, wrapped, steps, n, graph=False):
    from rl_rocket_6dof_amd import BatchedRocket6DOF, _lib, default_env_config

    kw = dict(seed=7, record_flags=True, track_episodes=wrapped)
    if wrapped:
        kw.update(remove_mass=True, clip_reward=(-1, 100), max_episode_steps=40)
    env = BatchedRocket6DOF(n, default_env_config(), **kw)
    for k, v in {**DEFAULTS, **shape}.items():
        _lib.tune(k, v, env.lib)
    try:
        gen = torch.Generator(device=env.device).manual_seed(3)
        acts = [torch.rand((n, 3), generator=gen, device=env.device) * 2 - 1 for _ in range(2)]
        env.reset()
        if graph:
            side = torch.cuda.Stream(device=env.device)
            with torch.cuda.stream(side):
                env.step_async(acts[0])
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                env.step_async(acts[1])                        # recorded with step index 2
            g.replay()
            torch.cuda.synchronize()
        else:
            for j in range(steps):
                env.step_async(acts[j % 2])
            torch.cuda.synchronize()
        out = [env.state.clone(), env.epconst.clone(), env.obs.clone(), env.reward.clone(), env.done.clone(), env.flags.clone(),
               env.final_obs.clone()]
        if wrapped:
            out += [env.ep_step.clone(), env.ep_return.clone(), env.stats.sum(0).clone()]
        return out
    finally:
        for k, v in DEFAULTS.items():
            _lib.tune(k, v, env.lib)


@pytest.mark.parametrize("wrapped", [False, True])
def test_every_launch_shape_is_bit_identical(wrapped):
    import torch

    n = 1 << 20
    ref = _run(torch, SHAPES[0], wrapped, 60, n)
    if wrapped:
        assert float(ref[-1][0]) > n / 2, "the time limit of 40 steps must have ended every episode at least once"
    for shape in SHAPES[1:]:
        got = _run(torch, shape, wrapped, 60, n)
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), f"shape {shape}: output {i} differs from the plain launch"


def test_default_launch_records_into_a_graph():
    """A graph of the step replays to the eager result."""
    import torch

    n = 1 << 20
    ref = _run(torch, {}, False, 2, n)
    got = _run(torch, {"even_xcd_share": 0.5}, False, 2, n, graph=True)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"output {i}"


def test_balanced_grid_shapes_are_bit_identical():
    """From 49 152 tiles (3.1 M envs) on the grid is dispatcher-balanced (a few tiles per wave, the tile range mirrored on odd
    steps): tiles per wave and the store form never change a result."""
    import torch

    n = 49152 * 64
    ref = _run(torch, {}, False, 40, n)
    for shape in ({"balanced_tiles_per_wave": 1}, {"balanced_tiles_per_wave": 4}, {"balanced_tiles_per_wave": 3, "streaming_stores": 1}):
        got = _run(torch, shape, False, 40, n)
        for i, (a, b) in enumerate(zip(ref, got)):
            assert torch.equal(a, b), f"shape {shape}: output {i} differs from the default launch"


def test_small_shards_ignore_the_knobs():
    """Below the persistent-grid size nothing engages: a 1000-env shard under every knob at once equals the default."""
    import torch

    ref = _run(torch, {}, True, 50, 1000)
    got = _run(torch, {"grid_blocks": 3, "even_xcd_share": 0.6, "sweep": 2, "sweep_flags": 4}, True, 50, 1000)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), f"output {i}"
