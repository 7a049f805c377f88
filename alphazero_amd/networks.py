"""Policy / value MLPs and the fused in-tree model.

Mirrors networks/networks.py:25-48, 153-193 (flax MLP: Dense + swish on all but
the last layer, lecun_uniform kernels, zero biases), alphazero.py:208-216
(forward = normalise, policy MLP, value MLP) and alphazero.py:317-339 (the
recurrent_fn = environment step + forward).  PyTorch only owns the parameter
storage; every forward pass runs in the HIP kernels of libaz_mcts.so.
"""
import math

import torch

from . import _native as nat
from . import base
from . import envs as envs_lib


def init_mlp(layer_sizes, generator=None, device="cuda"):
    """flax lecun_uniform Dense stack: list of (kernel[in,out], bias[out])."""
    layers = []
    for fan_in, fan_out in zip(layer_sizes[:-1], layer_sizes[1:]):
        lim = math.sqrt(3.0 / fan_in)
        w = (torch.rand((fan_in, fan_out), generator=generator, device=device) * 2 - 1) * lim
        layers.append((w.float().contiguous(),
                       torch.zeros((fan_out,), dtype=torch.float32, device=device)))
    return layers


def _mlp_struct(layers):
    if not 1 <= len(layers) <= nat.MAX_LAYERS:
        raise NotImplementedError(f"networks of 1..{nat.MAX_LAYERS} dense layers are supported")
    p = nat.MlpParams()
    p.n_layers = len(layers)
    p.sizes[0] = layers[0][0].shape[0]
    keep = []
    for i, (w, b) in enumerate(layers):
        w = w.detach().to(torch.float32).contiguous()
        b = b.detach().to(torch.float32).contiguous()
        if w.dim() != 2 or b.shape != (w.shape[1],) or w.shape[0] != p.sizes[i]:
            raise ValueError(f"layer {i}: kernel {tuple(w.shape)} / bias {tuple(b.shape)} mismatch")
        p.sizes[i + 1] = w.shape[1]
        p.w[i] = w.data_ptr()
        p.b[i] = b.data_ptr()
        keep += [w, b]
    return p, keep


class FusedModel:
    """Environment + observation normaliser + policy and value MLPs, with the weights
    re-laid-out for the matrix cores (az_model_pack).  Call `update` after the optimiser
    changed the parameters."""

    def __init__(self, env_id, policy_layers, value_layers, norm_mean=None, norm_std=None,
                 gamma=0.99, hidden_bf16=False):
        """hidden_bf16: run the layers between the first and the last one of each network on the
        bfloat16 matrix instructions (az_model_pack_bf16: bfloat16 weights and inputs, float32
        accumulation).  Not the reference's arithmetic (float32): results agree to about three
        digits, and nothing computed this way is comparable bit for bit with the CPU checker."""
        self.hidden_bf16 = bool(hidden_bf16)
        self.env_id = env_id
        self.code = envs_lib.env_code(env_id)
        self.obs_dim = envs_lib.observation_size(env_id)
        self.num_actions = envs_lib.action_size(env_id)
        self.gamma = float(gamma)
        self.update(policy_layers, value_layers, norm_mean, norm_std)

    def update(self, policy_layers, value_layers, norm_mean=None, norm_std=None):
        device = policy_layers[0][0].device
        pp, keep_p = _mlp_struct(policy_layers)
        vp, keep_v = _mlp_struct(value_layers)
        if pp.sizes[0] != self.obs_dim or vp.sizes[0] != self.obs_dim:
            raise ValueError("network input width must equal the observation size")
        if pp.sizes[pp.n_layers] != self.num_actions:
            raise ValueError("policy output width must equal the number of actions")
        self.num_atoms = int(vp.sizes[vp.n_layers])
        lib = nat.lib()
        size_fn, pack_fn = ((lib.az_model_packed_floats_bf16, lib.az_model_pack_bf16) if self.hidden_bf16
                            else (lib.az_model_packed_floats, lib.az_model_pack))
        n = size_fn(pp, vp)
        if n < 0:
            nat.check(-2)
        self.packed = torch.empty((n,), dtype=torch.float32, device=device)
        nat.check(pack_fn(pp, vp, nat.ptr(self.packed), nat.stream_ptr()))
        self._keep = keep_p + keep_v  # sources stay alive until the pack kernels have run
        f = dict(dtype=torch.float32, device=device)
        self.norm_mean = (torch.zeros(self.obs_dim, **f) if norm_mean is None
                          else norm_mean.detach().to(**f).contiguous())
        self.norm_std = (torch.ones(self.obs_dim, **f) if norm_std is None
                         else norm_std.detach().to(**f).contiguous())
        m = nat.Model()
        m.env_id, m.gamma = self.code, self.gamma
        m.obs_dim, m.num_actions, m.num_atoms = self.obs_dim, self.num_actions, self.num_atoms
        m.policy_layers, m.value_layers = pp.n_layers, vp.n_layers
        for i in range(nat.MAX_LAYERS + 1):
            m.policy_sizes[i] = pp.sizes[i]
            m.value_sizes[i] = vp.sizes[i]
        m.hidden_bf16 = 1 if self.hidden_bf16 else 0
        m.packed = self.packed.data_ptr()
        m.norm_mean = self.norm_mean.data_ptr()
        m.norm_std = self.norm_std.data_ptr()
        self.native = m
        return self

    def forward(self, obs):
        """alphazero.py:208-216: (prior_logits [B,A], value [B,K])."""
        obs = obs.to(torch.float32).contiguous()
        B = obs.shape[0]
        logits = torch.empty((B, self.num_actions), dtype=torch.float32, device=obs.device)
        value = torch.empty((B, self.num_atoms), dtype=torch.float32, device=obs.device)
        nat.check(nat.lib().az_forward(self.native, B, nat.ptr(obs), nat.ptr(logits),
                                       nat.ptr(value), nat.stream_ptr()))
        return logits, value


class FusedRecurrentFn:
    """The recurrent_fn of alphazero.py:317-339 for a built-in environment.

    Passed to a policy it selects the fused persistent search kernel (environment step
    and both MLPs inside the simulation loop).  It is also an ordinary recurrent_fn:
    calling it runs one batched environment step + forward (az_recurrent_step), which
    is what the per-simulation callback path does with it."""

    def __init__(self, model):
        self.model = model

    def __call__(self, params, rng_key, action, embedding):
        m = self.model
        state = embedding["state"].clone()
        time = embedding["time"].clone()
        B = state.shape[0]
        dev = state.device
        action = action.to(torch.int32).contiguous()
        reward = torch.empty((B,), dtype=torch.float32, device=dev)
        discount = torch.empty((B,), dtype=torch.float32, device=dev)
        logits = torch.empty((B, m.num_actions), dtype=torch.float32, device=dev)
        value = torch.empty((B, m.num_atoms), dtype=torch.float32, device=dev)
        nat.check(nat.lib().az_recurrent_step(
            m.native, B, nat.ptr(state), nat.ptr(time), nat.ptr(action), nat.ptr(reward),
            nat.ptr(discount), nat.ptr(logits), nat.ptr(value), nat.stream_ptr()))
        out = base.RecurrentFnOutput(reward=reward, discount=discount, prior_logits=logits,
                                     value=value)
        return out, {"state": state, "time": time}
