"""GPU: the on-device PPO loop drives the fused make_env() stack end to end."""
import math

import pytest

pytestmark = pytest.mark.gpu


def test_ppo_improves_return_on_the_real_env():
    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    env = BatchedRocket6DOF(8192, default_env_config(), remove_mass=True, clip_reward=(-1, 100),
                            max_episode_steps=1500, seed=42)
    algo = PPO(env, PPOConfig(n_steps=128, batch_size=1 << 16, net_arch=(128, 64), seed=42))   # main_6DOF.py:113
    hist = algo.learn(40_000_000)
    for h in hist:
        assert all(math.isfinite(v) for v in h.values())
    early = min(h["ep_rew_mean"] for h in hist[1:6])
    late = hist[-1]["ep_rew_mean"]
    print(f"[ppo] {len(hist)} iterations, {hist[-1]['timesteps']} steps: ep_rew_mean {early:.2f} -> {late:.2f}, "
          f"ep_len_mean {hist[-1]['ep_len_mean']:.0f}")
    assert late > early + 10.0 and late > -20.0
    assert env.obs.shape == (8192, 13) and bool(torch.isfinite(env.obs).all())


@pytest.mark.gpu
def test_tall_linear_gradients_match_plain_linear():
    """The split-K weight gradient of ppo.TallLinear against torch.nn.Linear on the same data."""
    import torch

    from rl_rocket_6dof_amd.ppo import TallLinear

    torch.manual_seed(0)
    for rows, fin, fout in ((1 << 17, 64, 64), (1 << 16, 14, 128), (98304, 64, 3)):
        ref = torch.nn.Linear(fin, fout).cuda()
        tall = TallLinear(fin, fout).cuda()
        tall.load_state_dict(ref.state_dict())
        x = torch.randn(rows, fin, device="cuda")
        x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        w = torch.randn(rows, fout, device="cuda") / rows
        (ref(x1) * w).sum().backward()
        (tall(x2) * w).sum().backward()
        for a, b in ((x1.grad, x2.grad), (ref.weight.grad, tall.weight.grad), (ref.bias.grad, tall.bias.grad)):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))


def test_graphed_update_follows_the_eager_update():
    """The HIP-graph replay of the minibatch step must read every rollout's fresh advantages / returns (a
    captured pointer to a stale tensor trains on old data): three iterations, graph on vs off, same seeds ->
    the same parameters up to fp32 reduction order."""
    import torch

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    out = []
    for graph in (False, True):
        env = BatchedRocket6DOF(4096, default_env_config(), remove_mass=True, clip_reward=(-1, 100),
                                max_episode_steps=1500, seed=3)
        algo = PPO(env, PPOConfig(n_steps=16, batch_size=1 << 13, n_epochs=2, net_arch=(64, 64), seed=3, graph_update=graph))
        hist = algo.learn(3 * 16 * 4096)
        assert (algo._graph is not None) == graph
        out.append((torch.cat([p.detach().reshape(-1) for p in algo.policy.parameters()]), hist[-1]))
    pe, pg = out[0][0], out[1][0]
    assert float((pe - pg).abs().max()) < 2e-3, float((pe - pg).abs().max())
    for k in ("pg_loss", "v_loss", "approx_kl"):
        assert abs(out[0][1][k] - out[1][1][k]) <= 2e-2 * max(1.0, abs(out[0][1][k])), (k, out[0][1][k], out[1][1][k])


def test_ppo_with_the_reference_topology_runs_fused_and_graphed():
    """PPOConfig(shared_trunk=True, net_arch=(128, 64)) = SB3 1.6's reading of the reference's net_arch=[128, 64]:
    rollouts go through the fused forward (one trunk), the update through the graphed minibatch step."""
    import math

    from rl_rocket_6dof_amd import BatchedRocket6DOF, default_env_config
    from rl_rocket_6dof_amd.ppo import PPO, PPOConfig

    env = BatchedRocket6DOF(4096, default_env_config(), remove_mass=True, clip_reward=(-1, 100), max_episode_steps=1500, seed=11)
    algo = PPO(env, PPOConfig(n_steps=32, batch_size=1 << 14, n_epochs=3, net_arch=(128, 64), shared_trunk=True, seed=11))
    assert algo.fused is not None and algo.fused.shared
    assert sum(p.numel() for p in algo.policy.parameters()) == 10311
    hist = algo.learn(6 * 32 * 4096)
    assert algo._graph is not None
    assert len(hist) == 6 and all(math.isfinite(v) for h in hist for v in h.values())
    assert sum(h["episodes"] for h in hist) > 1000       # episodes are ~170 steps long: the later rollouts finish some


def test_fused_ppo_loss_matches_autograd():
    """r6_ppo_loss_grad against the torch expression it replaces (SB3 PPO.train): loss, gradients w.r.t. action mean,
    value and log_std, and the logged statistics; with and without advantage normalisation, ratios well outside
    the clip range included."""
    import math

    import torch

    from rl_rocket_6dof_amd.ppo import _FusedPPOLoss

    torch.manual_seed(0)
    dev = "cuda"
    for n, norm_adv in ((70001, True), (4096, False)):
        mean = torch.randn(n, 3, device=dev, requires_grad=True)
        value = torch.randn(n, device=dev, requires_grad=True)
        log_std = torch.tensor([-0.4, 0.2, -1.0], device=dev, requires_grad=True)
        action = (mean + torch.randn(n, 3, device=dev) * log_std.exp()).detach()
        old_logp = torch.randn(n, device=dev) * 0.5 - 3.0
        adv, ret = torch.randn(n, device=dev) * 2 + 0.3, torch.randn(n, device=dev)
        clip, vf, ec = 0.2, 0.5, 0.01

        def ref(mean, value, log_std):
            var = torch.exp(2 * log_std)
            logp = (-0.5 * ((action - mean) ** 2 / var + 2 * log_std + math.log(2 * math.pi))).sum(-1)
            a = (adv - adv.mean()) / (adv.std() + 1e-8) if norm_adv else adv
            ratio = torch.exp(logp - old_logp)
            pg = -torch.min(a * ratio, a * ratio.clamp(1 - clip, 1 + clip)).mean()
            v = torch.nn.functional.mse_loss(ret, value)
            ent = -(0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum()
            return pg + ec * ent + vf * v, pg, v, -ent, ((ratio - 1).abs() > clip).float().mean(), ((ratio - 1) - (logp - old_logp)).mean()

        l_ref, pg, v, ent, cf, kl = ref(mean, value, log_std)
        g_ref = torch.autograd.grad(l_ref, (mean, value, log_std))
        norm = torch.stack([adv.mean(), 1.0 / (adv.std() + 1e-8)]) if norm_adv else None
        l_f, st = _FusedPPOLoss.apply(mean, value, log_std, action, old_logp, adv, ret, norm, clip, vf, ec)
        g_f = torch.autograd.grad(l_f, (mean, value, log_std))
        assert abs(float(l_f) - float(l_ref)) <= 2e-5 * max(1.0, abs(float(l_ref)))
        for a, b, name in zip(g_f, g_ref, ("mean", "value", "log_std")):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-6) + 1e-9, name
        for got, want in zip(st[:5].tolist(), (pg, v, ent, cf, kl)):
            assert abs(got - float(want)) <= 3e-5 * max(1.0, abs(float(want)))
        assert int(st[5]) == n


@pytest.mark.parametrize("arch,shared", [((128, 64), True), ((64, 64), False)])
@pytest.mark.parametrize("obs_dim,n_rows,n_mb,norm_adv", [(13, 5000, 3001, True), (14, 70000, 65536, False), (13, 300, 17, True)])
def test_fused_minibatch_grad_matches_autograd(obs_dim, n_rows, n_mb, norm_adv, arch, shared):
    """r6_ppo_minibatch_grad -- forward, PPO loss and the whole backward pass in one launch, for the reference's shared-trunk
    net [128, 64] and for SB3's default two MLPs (64, 64) -- against torch autograd through the same modules on the same
    minibatch (rows picked by index): every parameter gradient within 2e-5 of its tensor's largest entry, the loss
    statistics within 3e-5."""
    import math

    import torch

    from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

    torch.manual_seed(1)
    dev = "cuda"
    pol = ActorCritic(obs_dim, 3, arch, log_std_init=-0.3, shared_trunk=shared).to(dev)
    with torch.no_grad():                       # SB3's init makes the action head tiny (gain 0.01): scale it up so that its
        pol.pi[-1].weight.mul_(30.0)            # gradient path is not numerically negligible in the comparison
        pol.pi[-1].bias.uniform_(-0.2, 0.2)
        pol.log_std.copy_(torch.tensor([-0.4, 0.2, -1.0]))
    assert FusedMinibatchGrad.supports(pol)
    obs = torch.randn(n_rows, obs_dim, device=dev) * 0.7
    with torch.no_grad():
        mean_all, _ = pol._forward(obs)
    action = (mean_all + torch.randn(n_rows, 3, device=dev) * pol.log_std.exp()).detach()   # detached: the buffer's actions are data
    old_logp = pol._logp(mean_all, action).detach() + torch.randn(n_rows, device=dev) * 0.15
    adv, ret = torch.randn(n_rows, device=dev) * 2 + 0.3, torch.randn(n_rows, device=dev)
    mb = torch.randperm(n_rows, device=dev)[:n_mb]
    clip, vf, ec = 0.2, 0.5, 0.01

    # reference: autograd through the torch modules (ppo.PPO's unfused branch)
    a = adv[mb]
    if norm_adv:
        a = (a - a.mean()) / (a.std() + 1e-8)
    values, logp, entropy = pol.evaluate(obs[mb], action[mb])
    ratio = torch.exp(logp - old_logp[mb])
    pg = -torch.min(a * ratio, a * ratio.clamp(1 - clip, 1 + clip)).mean()
    v_loss = torch.nn.functional.mse_loss(ret[mb], values)
    ent = -entropy.mean()
    loss = pg + ec * ent + vf * v_loss
    params = list(pol.parameters())
    g_ref = torch.autograd.grad(loss, params)
    want_stats = (pg, v_loss, -ent, ((ratio - 1).abs() > clip).float().mean(), ((ratio - 1) - (logp - old_logp[mb])).mean())

    fused = FusedMinibatchGrad(pol)
    norm = fused.adv_norm(adv, mb).clone() if norm_adv else None
    if norm_adv:
        want_norm = torch.stack([adv[mb].mean(), 1.0 / (adv[mb].std() + 1e-8)])
        assert float(((norm - want_norm).abs() / want_norm.abs().clamp_min(1e-6)).max()) < 2e-6, (norm, want_norm)
    for p in params:
        p.grad = None
    st = fused(obs, action, old_logp, adv, ret, mb, norm, clip, vf, ec)
    torch.cuda.synchronize()
    names = [n for n, _ in pol.named_parameters()]
    for name, p, gr in zip(names, params, g_ref):
        assert p.grad is not None, name
        scale = float(gr.abs().max())
        err = float((p.grad - gr).abs().max())
        assert err <= 2e-5 * max(scale, 1e-7) + 1e-10, f"{name}: max abs err {err:.3e} vs scale {scale:.3e}"
    for got, want in zip(st[:5].tolist(), want_stats):
        assert abs(got - float(want)) <= 3e-5 * max(1.0, abs(float(want))), (st.tolist(), [float(w) for w in want_stats])
    assert int(st[5]) == n_mb
    print(f"[fused minibatch grad {arch} shared={shared} od={obs_dim} n={n_mb}] worst relative gradient error "
          f"{max(float((p.grad - g).abs().max()) / max(float(g.abs().max()), 1e-12) for p, g in zip(params, g_ref)):.2e}")


@pytest.mark.parametrize("arch,shared", [((128, 64), True), ((64, 64), False)])
@pytest.mark.parametrize("log2_ratio", [15, 26])
def test_fused_minibatch_grad_with_tiles_of_very_different_size(log2_ratio, arch, shared):
    """The f16 products of the backward pass scale delta2 by a power of two PER TILE (r6_ppo_update.hip, R6U_F16_BWD / R6U_F16_DW2), so
    tiles of any size relative to each other keep full accuracy.  Here every wave's first tile has gradients 2^15 (2^26) times smaller
    than its second one: 65 536 ordered rows, the first half scaled down -- advantages and value errors alike; every gradient within
    2e-5 of its tensor's largest entry as always.  (The measured-and-dropped R6U_DW2_DIRECT form, one factor per wave and launch,
    passes the first case and saturates in the second: this test is what it would have to be changed for.)"""
    import torch

    from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

    torch.manual_seed(3)
    dev, od, n = "cuda", 13, 65536
    pol = ActorCritic(od, 3, arch, log_std_init=-0.3, shared_trunk=shared).to(dev)
    with torch.no_grad():
        pol.pi[-1].weight.mul_(30.0)
        pol.log_std.copy_(torch.tensor([-0.4, 0.2, -1.0]))
    obs = torch.randn(n, od, device=dev) * 0.7
    with torch.no_grad():
        mean_all, v_all = pol._forward(obs)
    action = (mean_all + torch.randn(n, 3, device=dev) * pol.log_std.exp()).detach()
    old_logp = pol._logp(mean_all, action).detach() + torch.randn(n, device=dev) * 0.1
    adv, ret = torch.randn(n, device=dev) * 2 + 0.3, torch.randn(n, device=dev)
    small = 2.0 ** -log2_ratio
    adv[: n // 2] *= small
    ret[: n // 2] = (v_all.reshape(-1)[: n // 2] + (ret[: n // 2] - v_all.reshape(-1)[: n // 2]) * small).detach()
    mb = torch.arange(n, device=dev)
    clip, vf, ec = 0.2, 0.5, 0.01
    values, logp, entropy = pol.evaluate(obs, action)
    ratio = torch.exp(logp - old_logp)
    loss = -torch.min(adv * ratio, adv * ratio.clamp(1 - clip, 1 + clip)).mean() - ec * entropy.mean() + vf * torch.nn.functional.mse_loss(ret, values)
    params = list(pol.parameters())
    g_ref = torch.autograd.grad(loss, params)
    for p in params:
        p.grad = None
    FusedMinibatchGrad(pol)(obs, action, old_logp, adv, ret, mb, None, clip, vf, ec)
    torch.cuda.synchronize()
    worst = {}
    for (name, p), gr in zip(pol.named_parameters(), g_ref):
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), name
        scale, err = float(gr.abs().max()), float((p.grad - gr).abs().max())
        worst[name] = err / max(scale, 1e-12)
        assert err <= 2e-5 * max(scale, 1e-7) + 1e-10, f"{name}: max abs err {err:.3e} vs scale {scale:.3e}"
    print(f"[tiles 2^{log2_ratio} apart, {arch} shared={shared}] worst relative gradient error " + ", ".join(f"{k} {v:.1e}" for k, v in worst.items()))


def test_fused_adam_clip_step_matches_torch():
    """r6_adam_clip_step on the flat parameter / gradient / moment arrays against torch.nn.utils.clip_grad_norm_ +
    torch.optim.Adam(eps=1e-5) on the same tensors, several steps, with the clip active and inactive and a log-std floor."""
    import copy

    import torch

    from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

    torch.manual_seed(3)
    dev = "cuda"
    pol = ActorCritic(13, 3, (128, 64), log_std_init=-2.45, shared_trunk=True).to(dev)
    ref = copy.deepcopy(pol)
    fused = FusedMinibatchGrad(pol)                 # flattens pol's parameters in place
    for (na, a), (nb, b) in zip(pol.named_parameters(), ref.named_parameters()):
        assert na == nb and torch.equal(a, b)
    assert sum(p.numel() for p in pol.parameters()) == fused.flat_p.numel() and pol.log_std.data_ptr() == fused.flat_p[fused.log_std_offset:].data_ptr()
    opt = torch.optim.Adam(ref.parameters(), lr=3e-4, eps=1e-5)
    lr_dev = torch.tensor(3e-4, device=dev)
    for it, gscale in enumerate((5.0, 0.01, 1.0, 20.0)):      # total norms above and below max_grad_norm = 0.5
        g_flat = torch.randn_like(fused.flat_p) * gscale / fused.flat_p.numel() ** 0.5
        g_flat[fused.log_std_offset:] = torch.tensor([0.5, -0.2, 0.9], device=dev) * gscale     # pushes log_std down: the floor acts
        fused.flat_g.copy_(g_flat)
        off = 0
        for p in fused.params:                           # same gradients for the torch reference, mapped by parameter identity
            k = p.numel()
            rp = dict(ref.named_parameters())[[n for n, q in pol.named_parameters() if q is p][0]]
            rp.grad = g_flat[off:off + k].view_as(rp).clone()
            off += k
        torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.5)
        opt.step()
        with torch.no_grad():
            ref.log_std.clamp_(min=-2.5)
        fused.adam_step(lr_dev if it % 2 else 3e-4, 0.5, -2.5, betas=(0.9, 0.999), eps=1e-5)
        torch.cuda.synchronize()
        for (na, a), (nb, b) in zip(pol.named_parameters(), ref.named_parameters()):
            err = float((a - b).abs().max())
            assert err <= 2e-7 + 2e-6 * float(b.abs().max()) * 0 + 3e-7, (it, na, err)
    assert int(fused.adam_step_count) == 4
    assert float(pol.log_std.min()) >= -2.5


def test_batched_advantage_norm_and_running_stats():
    """r6_minibatch_adv_norm over an epoch's consecutive minibatches == one call per minibatch == torch; the gradient
    kernel's running sums stats[8:] add up its per-launch stats."""
    import torch

    from rl_rocket_6dof_amd.ppo import ActorCritic, FusedMinibatchGrad

    torch.manual_seed(2)
    rows, bs, nb, od = 50000, 4096, 7, 13
    pol = ActorCritic(od, 3, (128, 64), shared_trunk=True).cuda()
    fused = FusedMinibatchGrad(pol)
    adv = torch.randn(rows, device="cuda") * 3 + 1
    perm = torch.randperm(rows, device="cuda")[: nb * bs].contiguous()
    batched = fused.adv_norm(adv, perm, nb).clone()
    for b in range(nb):
        mb = perm[b * bs:(b + 1) * bs]
        one = fused.adv_norm(adv, mb).clone()
        assert torch.equal(one, batched[b])
        ref = torch.stack([adv[mb].mean(), 1.0 / (adv[mb].std() + 1e-8)])
        assert float((one - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    with pytest.raises(ValueError):
        fused.adv_norm(adv, perm[:-1], nb)
    obs, act = torch.randn(rows, od, device="cuda"), torch.randn(rows, 3, device="cuda")
    old, ret = torch.randn(rows, device="cuda") - 3, torch.randn(rows, device="cuda")
    fused.stats.zero_()
    tot = torch.zeros(6, device="cuda")
    for b in range(3):
        st = fused(obs, act, old, adv, ret, perm[b * bs:(b + 1) * bs], batched[b], 0.2, 0.5, 0.0)
        tot += st[:6]
    assert float((fused.stats[8:14] - tot).abs().max()) <= 1e-5 * float(tot.abs().max())


@pytest.mark.parametrize("n", [1, 1000, 12288, 12289, 40000])
def test_adam_clip_step_on_plain_arrays(n):
    """Both kernels behind r6_adam_clip_step (register-resident for n <= 12 288, strided loops above) against
    clip_grad_norm_ + Adam on one flat tensor."""
    import ctypes as C

    import torch

    from rl_rocket_6dof_amd import _lib

    lib = _lib.load()
    torch.manual_seed(n)
    p = torch.randn(n, device="cuda")
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=1e-3, eps=1e-5)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    for it in range(3):
        g = torch.randn(n, device="cuda") * (3.0 if it != 1 else 1e-3) / n ** 0.5
        ref.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref], 0.5)
        opt.step()
        rc = lib.r6_adam_clip_step(vp(p), vp(g), vp(m), vp(v), n, 0.5, 1e-3, None, 0.9, 0.999, 1e-5, vp(step), 0, 0, 0.0, None)
        _lib.check(rc, "r6_adam_clip_step", lib)
        assert float((p - ref.detach()).abs().max()) <= 5e-7, (n, it)
    assert int(step) == 3


def test_multi_rank_update_runs_from_a_graph_with_the_all_reduce_inside():
    """Round-4 review, Missing 3 / Next 5: with several ranks every minibatch step ran eagerly (`use_graph = ... and self.world == 1`).
    Now the gradient all-reduce is captured into the epoch graph.  Rehearsed here the only way one GPU allows: ONE rank on a real
    RCCL process group with R6_FORCE_PROCESS_GROUP=1 (tests/_ppo_rccl_graph_worker.py, a process of its own because of the process
    group).  Accepted outcomes, both printed: (a) the graph was captured with the collective inside and replayed -- weights bit for
    bit those of a trainer without collectives (an all-reduce over one rank is the identity); (b) the stack refused the capture --
    the error text is in `graph_collective_error`, the update fell back to eager steps, and the weights still agree.  Two real ranks
    run the same arithmetic over gloo in tests/test_ppo_host.py (CPU)."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "R6_FORCE_PROCESS_GROUP")}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_ppo_rccl_graph_worker.py"), str(port)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    c = res["collectives"]
    print(f"[ppo rccl graph] backend {res['backend']}: graph captured with the all-reduce inside: {c['graph_captured']}; "
          f"capture error: {c['graph_collective_error']}; weights identical to the collective-free trainer: {res['weights_identical']} "
          f"(max |diff| {res['max_abs_diff']:.3g})")
    assert res["backend"] == "nccl" and c["fused_update"] and res["plain"]["graph_captured"]
    assert c["graph_captured"] or c["graph_collective_error"], "neither a captured graph nor a recorded refusal"
    assert res["weights_identical"], res
    try:
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        json.dump(res, open(os.path.join(root, "gpurun_out", "ppo_rccl_graph.json"), "w"), indent=1)
    except OSError:
        pass
