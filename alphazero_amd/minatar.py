"""MinAtar Breakout and the convolutional torso the reference pairs it with.

`BreakoutEnv` mirrors `envs.Env` for gymnax's MinBreakout
(gymnax/gymnax/environments/minatar/breakout.py) under the reference's wrappers
(alphazero.py:884-889); the game logic runs in HIP kernels (include/az_minatar.h), this module
owns the per-lane tensors.  `AtariTorso` is networks/networks.py:51-130 (the convolutional
feature extractor the reference switches on for this environment, alphazero.py:246-267,
961-965) and `AtariNetworks` the torso with its two dense heads; their convolutions and dense
layers are plain library calls through PyTorch-ROCm, like the learner's networks.
`recurrent_fn()` builds the search's recurrent_fn of alphazero.py:317-339 from both; it runs on
the per-simulation kernels (the fused search kernels carry a four-float state and MLPs only).
"""
import ctypes as C
import dataclasses
import math
from typing import Any

import torch
import torch.nn.functional as F

from . import _native as nat
from . import base
from . import prng

WORDS, OBS, ACTIONS = nat.BREAKOUT_WORDS, nat.BREAKOUT_OBS, nat.BREAKOUT_ACTIONS
OBS_SHAPE = (10, 10, 4)


def _key_words(key):
    return (C.c_uint32 * 2)(*prng.as_key(key))


@dataclasses.dataclass(frozen=True)
class State:
    """Per-lane state of the wrapped environment (the fields of gymnax/wrappers/brax.py:16-24
    the wrappers read); `state` is the packed EnvState of include/az_minatar.h."""
    state: Any        # i32 [B, WORDS]
    obs: Any          # f32 [B, 10, 10, 4]
    real_obs: Any     # f32 [B, 10, 10, 4] observation before auto-reset
    reward: Any       # f32 [B]
    done: Any         # f32 [B]
    truncation: Any   # f32 [B]
    steps: Any        # f32 [B]
    rng: Any          # i32 [B, 2]

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    @property
    def embedding(self):
        return {"state": self.state}


class BreakoutEnv:
    """'Breakout-MinAtar' with the minimal action set ['n', 'l', 'r'] (breakout.py:48-60)."""
    env_id = "Breakout-MinAtar"
    observation_size = OBS_SHAPE
    action_size = ACTIONS

    def __init__(self, episode_length=1000, max_steps_in_episode=1000):
        self.episode_length = int(episode_length)
        self.max_steps_in_episode = int(max_steps_in_episode)   # EnvParams (breakout.py:24-26)

    def reset(self, rng, batch_size=None, device="cuda"):
        lib = nat.lib()
        if torch.is_tensor(rng) and rng.dim() == 2:
            keys = rng.to(device=device, dtype=torch.int32).contiguous()
            B = keys.shape[0]
        else:
            if batch_size is None:
                raise ValueError("batch_size is required when a single key is given")
            B = int(batch_size)
            keys = torch.empty((B, 2), dtype=torch.int32, device=device)
            nat.check(lib.az_split_keys(_key_words(rng), B, nat.ptr(keys), nat.stream_ptr()))
        state = torch.empty((B, WORDS), dtype=torch.int32, device=device)
        obs = torch.empty((B,) + OBS_SHAPE, dtype=torch.float32, device=device)
        rng_out = torch.empty((B, 2), dtype=torch.int32, device=device)
        nat.check(lib.az_breakout_reset(B, nat.ptr(keys), nat.ptr(state), nat.ptr(obs), nat.ptr(rng_out),
                                        nat.stream_ptr()))
        z = torch.zeros((B,), dtype=torch.float32, device=device)
        return State(state=state, obs=obs, real_obs=obs.clone(), reward=z, done=z.clone(),
                     truncation=z.clone(), steps=z.clone(), rng=rng_out)

    def step(self, st, action):
        """alphazero.py:383: one auto-resetting step of the real environment; a new State."""
        B = st.state.shape[0]
        new = {f.name: getattr(st, f.name).clone() for f in dataclasses.fields(State)}
        pre = torch.empty_like(new["state"])
        action = action.to(torch.int32).contiguous()
        nat.check(nat.lib().az_breakout_real_step(
            B, self.episode_length, self.max_steps_in_episode, nat.ptr(new["state"]), nat.ptr(pre),
            nat.ptr(new["obs"]), nat.ptr(new["real_obs"]), nat.ptr(new["reward"]), nat.ptr(new["done"]),
            nat.ptr(new["truncation"]), nat.ptr(new["steps"]), nat.ptr(new["rng"]), nat.ptr(action),
            nat.stream_ptr()))
        return State(**new)

    def model_step(self, state, action):
        """The environment as the search's model (VmapWrapper(GymnaxToBraxWrapper(env)).step,
        alphazero.py:320): (new packed state, obs, reward, done, truncation)."""
        B = state.shape[0]
        dev = state.device
        out = torch.empty_like(state)
        obs = torch.empty((B,) + OBS_SHAPE, dtype=torch.float32, device=dev)
        reward, done, trunc = (torch.empty((B,), dtype=torch.float32, device=dev) for _ in range(3))
        nat.check(nat.lib().az_breakout_step(
            B, self.max_steps_in_episode, nat.ptr(state.contiguous()), nat.ptr(action.to(torch.int32).contiguous()),
            nat.ptr(out), nat.ptr(obs), nat.ptr(reward), nat.ptr(done), nat.ptr(trunc), nat.stream_ptr()))
        return out, obs, reward, done, trunc


# ---------------------------------------------------------------- networks
def _orthogonal(shape, gain, generator, device):
    """jax.nn.initializers.orthogonal(scale) for a [.., out] kernel (rows = everything else)."""
    rows, cols = math.prod(shape[:-1]), shape[-1]
    a = torch.randn((max(rows, cols), min(rows, cols)), generator=generator, device=device)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.T
    return (gain * q).reshape(shape).contiguous()


class AtariTorso:
    """networks/networks.py:51-130 with flax's parameter layout: conv_3 3x3 (4 -> 32) and
    conv_4 2x2 (32 -> 64), VALID, stride 1, kernels [kh, kw, in, out]; ReLU after each; the
    7 x 7 x 64 map flattened in (y, x, channel) order; Dense 512 WITHOUT an activation
    (networks.py:117-120); then one Dense + activation per entry of `layer_sizes`.
    Orthogonal(sqrt 2) kernels, zero biases (networks.py:55, 101-102)."""

    def __init__(self, layer_sizes=(512,), obs_shape=OBS_SHAPE, generator=None, device="cuda",
                 activation=F.relu):
        g, gain = generator, math.sqrt(2.0)
        h, w, c = obs_shape
        self.activation = activation
        self.conv = [(_orthogonal((3, 3, c, 32), gain, g, device), torch.zeros(32, device=device)),
                     (_orthogonal((2, 2, 32, 64), gain, g, device), torch.zeros(64, device=device))]
        flat = (h - 3) * (w - 3) * 64
        sizes = (flat, 512) + tuple(layer_sizes)
        self.dense = [(_orthogonal((sizes[i], sizes[i + 1]), gain, g, device),
                       torch.zeros(sizes[i + 1], device=device)) for i in range(len(sizes) - 1)]
        self.out_features = sizes[-1]

    def leaves(self):
        return [t for w, b in self.conv + self.dense for t in (w, b)]

    def __call__(self, obs):
        x = obs.reshape((-1,) + tuple(obs.shape[-3:])).permute(0, 3, 1, 2)            # NHWC -> NCHW
        for w, b in self.conv:
            x = self.activation(F.conv2d(x, w.permute(3, 2, 0, 1), b))               # HWIO -> OIHW
        x = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1)                             # (y, x, c) order
        w, b = self.dense[0]
        x = x @ w + b
        for w, b in self.dense[1:]:
            x = self.activation(x @ w + b)
        return x.reshape(tuple(obs.shape[:-3]) + (x.shape[-1],))


class AtariNetworks:
    """alphazero.py:246-267: the shared torso and, on its features, one dense layer each for the
    policy logits and the value atoms (`hidden_layer_sizes=()`; lecun_uniform kernels)."""

    def __init__(self, num_actions=ACTIONS, num_atoms=1, dense_layer_sizes=(512,), generator=None,
                 device="cuda"):
        from . import networks
        self.torso = AtariTorso(dense_layer_sizes, generator=generator, device=device)
        f = self.torso.out_features
        self.policy = networks.init_mlp((f, num_actions), generator, device)
        self.value = networks.init_mlp((f, num_atoms), generator, device)

    def leaves(self):
        """Every parameter tensor (the L2 penalty and the optimiser cover the torso too,
        alphazero.py:787-789)."""
        return self.torso.leaves() + [t for w, b in self.policy + self.value for t in (w, b)]

    def forward(self, obs, mean=None, std=None):
        """alphazero.py:208-216 with use_feature_extractor: (prior logits [B,A], value [B,K]).
        obs [..., 10, 10, 4] or flattened [..., 400]; mean / std (flat or observation-shaped): the
        running-statistics normaliser in front of the torso (networks/networks.py:145-147)."""
        if obs.shape[-1] == OBS:
            obs = obs.reshape(tuple(obs.shape[:-1]) + OBS_SHAPE)
        if mean is not None:
            obs = (obs - mean.reshape(OBS_SHAPE)) / std.reshape(OBS_SHAPE)
        h = self.torso(obs)
        (wp, bp), (wv, bv) = self.policy[0], self.value[0]
        return h @ wp + bp, h @ wv + bv


def recurrent_fn(env, nets, gamma=0.99, normalizer=None):
    """alphazero.py:317-339 for this environment: step the model, evaluate the networks on the
    new observation, discount = gamma * (1 if truncated else 1 - done).
    normalizer (optional): a callable returning the current (mean, std) of the observation
    statistics, read at every call."""
    def fn(params, rng_key, action, embedding):
        state, obs, reward, done, trunc = env.model_step(embedding["state"], action)
        mean, std = normalizer() if normalizer is not None else (None, None)
        with torch.no_grad():
            logits, value = nets.forward(obs, mean, std)
        discount = gamma * torch.where(trunc != 0, torch.ones_like(done), 1 - done)
        return base.RecurrentFnOutput(reward=reward, discount=discount, prior_logits=logits,
                                      value=value), {"state": state}
    return fn
