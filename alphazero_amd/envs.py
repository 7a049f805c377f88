"""The built-in environments (CartPole-v1, MountainCar-v0, Acrobot-v1 and its
never-terminating variant Acrobot-v2, gymnax registration.py:28-37).

Mirrors the training environment the reference builds at alphazero.py:884-890:
``AutoResetWrapper(VmapWrapper(EpisodeWrapper(GymnaxToBraxWrapper(env))))`` for the
real environment and ``VmapWrapper(GymnaxToBraxWrapper(env))`` inside the search
(the latter is fused into the search kernels, see networks.FusedRecurrentFn).
Dynamics, reset and wrapper logic run in HIP kernels (az_env_reset /
az_real_env_step); this module only owns the per-lane tensors.
"""
import ctypes as C
import dataclasses
from typing import Any

import numpy as np
import torch

from . import _native as nat
from . import prng

ENV_IDS = {"CartPole-v1": nat.ENV_CARTPOLE, "MountainCar-v0": nat.ENV_MOUNTAINCAR,
           "Acrobot-v1": nat.ENV_ACROBOT,
           "Acrobot-v2": nat.ENV_ACROBOT_NOTERM}   # AcrobotNoTerm (gymnax registration.py:36-37)
_OBS = {nat.ENV_CARTPOLE: 4, nat.ENV_MOUNTAINCAR: 2, nat.ENV_ACROBOT: 6, nat.ENV_ACROBOT_NOTERM: 6}
_ACT = {nat.ENV_CARTPOLE: 2, nat.ENV_MOUNTAINCAR: 3, nat.ENV_ACROBOT: 3, nat.ENV_ACROBOT_NOTERM: 3}


def env_code(env_id):
    if isinstance(env_id, str):
        if env_id not in ENV_IDS:
            raise NotImplementedError(f"environment {env_id!r} is not built in; "
                                      f"available: {sorted(ENV_IDS)}")
        return ENV_IDS[env_id]
    return int(env_id)


def observation_size(env_id):
    return _OBS[env_code(env_id)]


def action_size(env_id):
    return _ACT[env_code(env_id)]


def _key_words(key):
    return (C.c_uint32 * 2)(*prng.as_key(key))


@dataclasses.dataclass(frozen=True)
class State:
    """Per-lane environment state: the fields of the reference's brax-like State
    (gymnax/gymnax/wrappers/brax.py:16-24) that the three environments and the wrappers read.

    pipeline_state = (state f32[B,4], time i32[B]); info = {_rng, truncation, steps}."""
    state: Any        # f32 [B, 4]  native pipeline state
    time: Any         # i32 [B]     pipeline_state.time
    obs: Any          # f32 [B, obs]
    real_obs: Any     # f32 [B, obs] observation before auto-reset
    reward: Any       # f32 [B]
    done: Any         # f32 [B]
    truncation: Any   # f32 [B] info['truncation']
    steps: Any        # f32 [B] info['steps']
    rng: Any          # i32 [B, 2] info['_rng'] (bit pattern of the uint32 key words)

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    @property
    def embedding(self):
        """What a search needs as RootFnOutput.embedding."""
        return {"state": self.state, "time": self.time}


class Env:
    """The wrapped training environment (alphazero.py:884-889)."""

    def __init__(self, env_id, episode_length=500, action_repeat=1):
        if action_repeat != 1:
            raise NotImplementedError("action_repeat != 1 is not supported")
        self.env_id = env_id
        self.code = env_code(env_id)
        self.episode_length = int(episode_length)
        self.observation_size = _OBS[self.code]
        self.action_size = _ACT[self.code]

    def reset(self, rng, batch_size=None, device="cuda"):
        """`rng`: one key (split into `batch_size` lane keys, as alphazero.py does with
        jax.random.split) or an int32/uint32 tensor [B,2] of lane keys."""
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
        f = dict(dtype=torch.float32, device=device)
        state = torch.empty((B, 4), **f)
        time = torch.empty((B,), dtype=torch.int32, device=device)
        obs = torch.empty((B, self.observation_size), **f)
        rng_out = torch.empty((B, 2), dtype=torch.int32, device=device)
        nat.check(lib.az_env_reset(self.code, B, nat.ptr(keys), nat.ptr(state), nat.ptr(time),
                                   nat.ptr(obs), nat.ptr(rng_out), nat.stream_ptr()))
        z = torch.zeros((B,), **f)
        return State(state=state, time=time, obs=obs, real_obs=obs.clone(), reward=z,
                     done=z.clone(), truncation=z.clone(), steps=z.clone(), rng=rng_out)

    def step(self, st, action):
        """alphazero.py:383: one auto-resetting step; returns a new State."""
        B = st.state.shape[0]
        new = {f.name: getattr(st, f.name).clone() for f in dataclasses.fields(State)}
        action = action.to(torch.int32).contiguous()
        nat.check(nat.lib().az_real_env_step(
            self.code, B, self.episode_length, nat.ptr(new["state"]), nat.ptr(new["time"]),
            nat.ptr(new["obs"]), nat.ptr(new["real_obs"]), nat.ptr(new["reward"]),
            nat.ptr(new["done"]), nat.ptr(new["truncation"]), nat.ptr(new["steps"]),
            nat.ptr(new["rng"]), nat.ptr(action), nat.stream_ptr()))
        return State(**new)

    def step_(self, st, action):
        """In-place variant of `step` (no allocation): updates the tensors of `st`.  `action` must
        already be what the kernel reads, int32 [B] contiguous (it is not converted here, which would
        allocate): any other integer width would be read as interleaved 32-bit words."""
        B = st.state.shape[0]
        if action.dtype != torch.int32 or not action.is_contiguous() or action.shape != (B,):
            raise TypeError(f"step_ needs a contiguous int32 action tensor of shape [{B}], got "
                            f"{action.dtype} {tuple(action.shape)}")
        nat.check(nat.lib().az_real_env_step(
            self.code, B, self.episode_length, nat.ptr(st.state), nat.ptr(st.time),
            nat.ptr(st.obs), nat.ptr(st.real_obs), nat.ptr(st.reward), nat.ptr(st.done),
            nat.ptr(st.truncation), nat.ptr(st.steps), nat.ptr(st.rng), nat.ptr(action),
            nat.stream_ptr()))
        return st


def make(env_id, **kwargs):
    if env_id == "Breakout-MinAtar":   # alphazero.py:48; its own state layout and kernels
        from . import minatar
        return minatar.BreakoutEnv(**kwargs)
    return Env(env_id, **kwargs)
