"""Fused actor-critic forward for rollouts: ``r6_policy_act`` behind an SB3-shaped ``forward``.

``FusedActorCritic(policy)`` wraps a ``ppo.ActorCritic`` (the SB3 1.6 ``MlpPolicy`` the reference trains,
main_6DOF.py:107-139) and evaluates it for N envs in ONE HIP launch -- both tanh MLPs on the matrix cores in
exact fp32, hidden activations in registers, Gaussian sample + log-probability + Box clip in the epilogue --
instead of ~20 torch kernels whose (N, 64) intermediates go through HBM.  The torch module stays the owner
of the weights (the kernel reads the parameter tensors in place), so an optimiser step is visible to the
next ``forward`` without any copy.  No CPU path: needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .params import R6PolicyNet

SUPPORTED_ARCHS = ((64, 64), (128, 64))


def _linears(seq):
    return [m for m in seq if isinstance(m, torch.nn.Linear)]


class FusedActorCritic:
    def __init__(self, policy, seed: int = 0, env_offset: int = 0, lib_path: Optional[str] = None):
        self.lib = _lib.load(lib_path)
        self.policy = policy
        pi, vf = _linears(policy.pi), _linears(policy.vf)
        if len(pi) != 3 or len(vf) != 3:
            raise ValueError("the fused forward supports two hidden layers per network")
        self.obs_dim = pi[0].in_features
        arch = (pi[0].out_features, pi[1].out_features)
        self.shared = pi[0] is vf[0] and pi[1] is vf[1]        # SB3 1.6 plain-list net_arch: one trunk, two heads
        if arch not in SUPPORTED_ARCHS or (vf[0].out_features, vf[1].out_features) != arch or self.obs_dim not in (13, 14):
            raise ValueError(f"supported: obs_dim 13|14, hidden {SUPPORTED_ARCHS}; got obs_dim {self.obs_dim}, hidden {arch}")
        if pi[2].out_features != 3 or vf[2].out_features != 1:
            raise ValueError("expected a 3-dim action head and a scalar value head")
        self._params = [pi[0].weight, pi[0].bias, pi[1].weight, pi[1].bias, pi[2].weight, pi[2].bias,
                        vf[0].weight, vf[0].bias, vf[1].weight, vf[1].bias, vf[2].weight, vf[2].bias, policy.log_std]
        for t in self._params:
            if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("policy parameters must be contiguous float32 CUDA tensors")
        self.device = self._params[0].device
        self.arch = arch
        self.seed, self.env_offset = int(seed), int(env_offset)
        self._calls = 0

    def _net(self) -> R6PolicyNet:
        # rebuilt per call: optimisers update parameters in place, but .data_ptr() is what the launch needs
        net = R6PolicyNet()
        net.obs_dim, net.hidden1, net.hidden2, net.shared = self.obs_dim, self.arch[0], self.arch[1], int(self.shared)
        names = ("pi_w1", "pi_b1", "pi_w2", "pi_b2", "pi_w3", "pi_b3", "vf_w1", "vf_b1", "vf_w2", "vf_b2", "vf_w3", "vf_b3", "log_std")
        for name, t in zip(names, self._params):
            setattr(net, name, t.data_ptr())
        return net

    @torch.no_grad()
    def forward(self, obs: torch.Tensor, deterministic: bool = False, step_index: Optional[int] = None,
                out: Optional[Tuple[torch.Tensor, ...]] = None, step_base: Optional[torch.Tensor] = None,
                row_flags: Optional[torch.Tensor] = None, flag_mask: int = 0):
        """Returns ``(clipped_actions (N,3), raw_actions (N,3), values (N,), log_probs (N,))``.
        ``step_index`` keys the noise stream (defaults to an internal call counter); ``out`` reuses buffers.
        ``step_base``: int64 device scalar added to ``step_index`` inside the kernel (a launch replayed from a HIP graph
        keeps its by-value arguments; the base in device memory moves on).  ``row_flags`` (N,) uint8 + ``flag_mask``:
        only 32-row tiles with a flagged row are evaluated, the outputs of the others are left untouched."""
        if obs.device != self.device or obs.dtype != torch.float32 or not obs.is_contiguous() or obs.dim() != 2 \
                or obs.shape[1] != self.obs_dim:
            raise ValueError(f"obs must be a contiguous float32 (N, {self.obs_dim}) tensor on {self.device}")
        n = obs.shape[0]
        if out is None:
            f32 = dict(dtype=torch.float32, device=self.device)
            out = (torch.empty((n, 3), **f32), torch.empty((n, 3), **f32), torch.empty((n,), **f32), torch.empty((n,), **f32))
        act, raw, val, logp = out
        self._calls += 1
        step = self._calls if step_index is None else int(step_index)
        net = self._net()
        if step_base is not None and (step_base.dtype != torch.int64 or step_base.device != self.device or step_base.numel() != 1):
            raise ValueError("step_base must be an int64 scalar tensor on the policy's device")
        if row_flags is not None and (row_flags.dtype != torch.uint8 or row_flags.device != self.device
                                      or row_flags.numel() != n or not row_flags.is_contiguous()):
            raise ValueError("row_flags must be a contiguous uint8 (N,) tensor on the policy's device")
        with torch.cuda.device(self.device):
            rc = self.lib.r6_policy_act(C.byref(net), obs.data_ptr(), n, act.data_ptr(), raw.data_ptr(), val.data_ptr(),
                                        logp.data_ptr(), self.seed, step, None if step_base is None else step_base.data_ptr(),
                                        self.env_offset, int(bool(deterministic)),
                                        None if row_flags is None else row_flags.data_ptr(), int(flag_mask),
                                        C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        _lib.check(rc, "r6_policy_act", self.lib)
        return act, raw, val, logp

    __call__ = forward


def load_sb3_policy(path: str, device="cuda"):
    """Builds a ``ppo.ActorCritic`` from a stable-baselines3 (1.x) PPO checkpoint ``*.zip`` -- the format of
    the reference's ``best_model_*.zip`` (``PPO.load`` in montecarlo_script.py / main_6DOF.py) -- without SB3.

    Only the JSON ``data`` entry and the tensors of ``policy.pth`` are read; the tensors are loaded with
    ``torch.load(..., weights_only=True)``, so no pickled code of the checkpoint is ever executed.
    Both SB3 topologies are recognised: ``mlp_extractor.policy_net`` / ``value_net`` (separate MLPs) and
    ``mlp_extractor.shared_net`` (plain-list ``net_arch``, one trunk)."""
    import io
    import zipfile

    from .ppo import ActorCritic

    with zipfile.ZipFile(path) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu", weights_only=True)
    return policy_from_sb3_state_dict(sd, device=device)


def policy_from_sb3_state_dict(sd, device="cuda"):
    """``ActorCritic`` from the tensors of an SB3 ``ActorCriticPolicy.state_dict()`` (two tanh hidden layers)."""
    from .ppo import ActorCritic

    shared = any(k.startswith("mlp_extractor.shared_net.") for k in sd)
    pre = "mlp_extractor.shared_net." if shared else "mlp_extractor.policy_net."
    w1, w2 = sd[pre + "0.weight"], sd[pre + "2.weight"]
    pol = ActorCritic(w1.shape[1], sd["action_net.weight"].shape[0], (w1.shape[0], w2.shape[0]), shared_trunk=shared)
    lin = lambda seq: [m for m in seq if isinstance(m, torch.nn.Linear)]
    pi, vf = lin(pol.pi), lin(pol.vf)
    with torch.no_grad():
        for m, key in ((pi[0], pre + "0"), (pi[1], pre + "2"), (pi[2], "action_net"), (vf[2], "value_net")):
            m.weight.copy_(sd[key + ".weight"])
            m.bias.copy_(sd[key + ".bias"])
        if not shared:
            for m, key in ((vf[0], "mlp_extractor.value_net.0"), (vf[1], "mlp_extractor.value_net.2")):
                m.weight.copy_(sd[key + ".weight"])
                m.bias.copy_(sd[key + ".bias"])
        pol.log_std.copy_(sd["log_std"])
    return pol.to(device)


def load_policy(path: str, device="cuda"):
    """``ActorCritic`` from a state_dict written by ``examples/train_ppo.py --save`` (tensor-only load); the
    topology is read off the tensors (a shared trunk appears under both ``pi.*`` and ``vf.*`` with equal values)."""
    from .ppo import ActorCritic

    sd = torch.load(path, map_location="cpu", weights_only=True)
    shared = all(k not in sd or torch.equal(sd[k], sd["pi" + k[2:]]) for k in ("vf.0.weight", "vf.0.bias", "vf.2.weight", "vf.2.bias"))
    w1, w2 = sd["pi.0.weight"], sd["pi.2.weight"]
    pol = ActorCritic(w1.shape[1], sd["pi.4.weight"].shape[0], (w1.shape[0], w2.shape[0]), shared_trunk=shared)
    pol.load_state_dict(sd)
    return pol.to(device)
