"""GPU replay queues behind the reference's replay-buffer interface.

Mirrors replay_buffers.py: `QueueBase` (:69-170), `UniformSamplingQueue` (:250-277) and
`PrioritizedSamplingQueue` (:280-339) -- same constructor arguments, `init`, `insert`,
`sample`, `set_priorities`, `size`, same error messages.  Differences a caller sees:

* rows are flat float32 tensors `[n, D]` (the reference ravels its `MCTSTransition`
  pytree into exactly such rows, replay_buffers.py:86-95, priority and weight last);
  `dummy_data_sample` may be a 1-D tensor or an int giving D;
* the state is updated in place and returned (the reference returns a new pytree);
* `state.data` is produced on demand in the reference's logical row order -- the rows
  themselves live in a ring in HBM and are never rolled;
* keys are pairs of uint32 words (see `prng`).

All compute is in libaz_mcts.so (include/az_replay.h); there is no fallback.
"""
import ctypes as C

import torch

from . import _native as nat
from . import prng


class ReplayBufferState:
    """replay_buffers.py:58-66."""

    def __init__(self, desc, cstate, tensors, key, bs):
        self._desc, self._c, self._tensors = desc, cstate, tensors
        self.key = prng.as_key(key)
        self.current_sample_indices = torch.zeros(bs, dtype=torch.int32,
                                                  device=tensors["rows"].device)

    @property
    def insert_position(self):
        return int(self._c.insert_position)

    @property
    def sample_position(self):
        return int(self._c.sample_position)

    @property
    def data(self):
        """The reference's `data[cap, D]` (logical order, priority and weight columns filled)."""
        out = torch.empty((self._desc.cap, self._desc.D), dtype=torch.float32,
                          device=self._tensors["rows"].device)
        nat.check(nat.lib().az_replay_export(C.byref(self._desc), C.byref(self._c), nat.ptr(out),
                                             nat.stream_ptr()))
        return out


class QueueBase:
    """replay_buffers.py:69-170."""

    def __init__(self, max_replay_size, dummy_data_sample, sample_batch_size):
        if isinstance(dummy_data_sample, int):
            data_size = dummy_data_sample
        else:
            data_size = int(torch.as_tensor(dummy_data_sample).numel())
        self._data_shape = (int(max_replay_size), data_size)
        self._sample_batch_size = int(sample_batch_size)
        self._size = 0

    def init(self, key, device="cuda"):
        cap, D = self._data_shape
        dev = torch.device(device)
        lib = nat.lib()
        t = {
            "rows": torch.empty((cap, D), dtype=torch.float32, device=dev),
            "priority": torch.empty(cap, dtype=torch.float32, device=dev),
            "weight": torch.empty(cap, dtype=torch.float32, device=dev),
            "cum": torch.empty(cap, dtype=torch.float32, device=dev),
            "partial": torch.zeros(int(lib.az_replay_partial_floats(cap)), dtype=torch.float32,
                                   device=dev),
            "owner": torch.empty(cap, dtype=torch.int32, device=dev),
        }
        desc = nat.Replay(*[nat.ptr(t[k]) for k in ("rows", "priority", "weight", "cum", "partial",
                                                    "owner")], cap, D)
        cstate = nat.ReplayState()
        nat.check(lib.az_replay_init(C.byref(desc), C.byref(cstate), nat.stream_ptr()))
        return ReplayBufferState(desc, cstate, t, key, self._sample_batch_size)

    def check_can_insert(self, buffer_state, samples, shards):
        """replay_buffers.py:108-118."""
        assert isinstance(shards, int), 'This method should not be JITed.'
        insert_size = samples.shape[0] // shards
        if self._data_shape[0] < insert_size:
            raise ValueError(
                'Trying to insert a batch of samples larger than the maximum replay'
                f' size. num_samples: {insert_size}, max replay size'
                f' {self._data_shape[0]}')
        self._size = min(self._data_shape[0], self._size + insert_size)

    def check_can_sample(self, buffer_state, shards):
        pass

    def insert(self, buffer_state, samples):
        """replay_buffers.py:120-159."""
        if tuple(samples.shape[1:]) != (self._data_shape[1],):
            raise ValueError(f"samples must be [n, {self._data_shape[1]}] float32 rows")
        self.check_can_insert(buffer_state, samples, 1)
        samples = samples.to(torch.float32).contiguous()
        nat.check(nat.lib().az_replay_insert(C.byref(buffer_state._desc), C.byref(buffer_state._c),
                                             nat.ptr(samples), samples.shape[0], nat.stream_ptr()))
        return buffer_state

    def sample(self, buffer_state):
        self.check_can_sample(buffer_state, 1)
        return self.sample_internal(buffer_state)

    def sample_internal(self, buffer_state):
        raise NotImplementedError(f'{self.__class__}.sample() is not implemented.')

    def size(self, buffer_state):
        return buffer_state.insert_position - buffer_state.sample_position

    def set_priorities(self, buffer_state, priorities):
        return buffer_state

    def _out(self, buffer_state):
        dev = buffer_state._tensors["rows"].device
        return (torch.empty((self._sample_batch_size, self._data_shape[1]), dtype=torch.float32,
                            device=dev), (C.c_uint32 * 2))


class UniformSamplingQueue(QueueBase):
    """replay_buffers.py:250-277."""

    def sample_internal(self, buffer_state):
        key, sample_key = prng.split(buffer_state.key, 2)
        batch, K2 = self._out(buffer_state)
        nat.check(nat.lib().az_replay_sample_uniform(
            C.byref(buffer_state._desc), C.byref(buffer_state._c), C.byref(K2(*sample_key)),
            self._sample_batch_size, nat.ptr(buffer_state.current_sample_indices), nat.ptr(batch),
            nat.stream_ptr()))
        buffer_state.key = key
        return buffer_state, batch


class PrioritizedSamplingQueue(QueueBase):
    """replay_buffers.py:280-339."""

    def __init__(self, max_replay_size, dummy_data_sample, sample_batch_size,
                 per_importance_sampling_beta=0):
        super().__init__(max_replay_size, dummy_data_sample, sample_batch_size)
        self.per_importance_sampling_beta = per_importance_sampling_beta

    def sample_internal(self, buffer_state):
        key, sample_key = prng.split(buffer_state.key, 2)
        batch, K2 = self._out(buffer_state)
        nat.check(nat.lib().az_replay_sample_prioritized(
            C.byref(buffer_state._desc), C.byref(buffer_state._c), C.byref(K2(*sample_key)),
            float(self.per_importance_sampling_beta), self._sample_batch_size,
            nat.ptr(buffer_state.current_sample_indices), nat.ptr(batch), nat.stream_ptr()))
        buffer_state.key = key
        return buffer_state, batch

    def set_priorities(self, buffer_state, priorities):
        pri = priorities.to(torch.float32).reshape(-1).contiguous()
        if pri.numel() != self._sample_batch_size:
            raise ValueError("one priority per sampled row is required")
        nat.check(nat.lib().az_replay_set_priorities(
            C.byref(buffer_state._desc), C.byref(buffer_state._c),
            nat.ptr(buffer_state.current_sample_indices), nat.ptr(pri), pri.numel(),
            nat.stream_ptr()))
        return buffer_state

    def scan(self, buffer_state):
        """(cum[cap], total) of the masked priorities as the sampler computes them."""
        dev = buffer_state._tensors["rows"].device
        cum = torch.empty(self._data_shape[0], dtype=torch.float32, device=dev)
        total = torch.empty(1, dtype=torch.float32, device=dev)
        nat.check(nat.lib().az_replay_scan(C.byref(buffer_state._desc), C.byref(buffer_state._c),
                                           nat.ptr(cum), nat.ptr(total), nat.stream_ptr()))
        return cum, total
