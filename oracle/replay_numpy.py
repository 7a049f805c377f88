"""CPU restatement of the reference's replay queues -- TEST INFRASTRUCTURE ONLY.

Follows replay_buffers.py literally (np.roll and all): QueueBase.insert_internal
(:120-159), UniformSamplingQueue.sample_internal (:260-277), PrioritizedSamplingQueue
.sample_internal (:304-335) and .set_priorities (:337-339), with jax.random.choice /
randint / uniform / split restated from JAX's public algorithm on the shared threefry.

PARITY UNPINNED against the reference itself: it has no tests or fixtures for this path
and JAX is not installed here.  Two things are a choice of this restatement, because XLA
does not document them: the association order of the float32 sum / cumulative sum (fixed
here as the chunked order of include/az_replay.h) and which write wins when
set_priorities sees an index twice (the last one, as NumPy does).
"""
import numpy as np

from . import c_oracle

SCAN_T, SCAN_B = 16, 256
SCAN_CHUNK = SCAN_T * SCAN_B


def _excl_cumsum(x, axis):
    """Exclusive left-to-right float32 running sum along `axis` (first element +0)."""
    c = np.cumsum(x, axis=axis, dtype=np.float32)
    z = np.zeros_like(np.take(c, [0], axis=axis))
    return np.concatenate([z, np.delete(c, -1, axis=axis)], axis=axis)


def chunked_cumsum(x):
    """float32 inclusive prefix sums in the association order of include/az_replay.h."""
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    nb = max(1, -(-n // SCAN_CHUNK))
    pad = np.zeros(nb * SCAN_CHUNK, np.float32)
    pad[:n] = x
    v = pad.reshape(nb, SCAN_B, SCAN_T)
    run = np.cumsum(v, axis=2, dtype=np.float32)             # left to right inside a run
    run_tot = run[:, :, -1]
    run_off = _excl_cumsum(run_tot, axis=1)                  # runs of a chunk, left to right
    chunk_tot = (run_off[:, -1] + run_tot[:, -1]).astype(np.float32)
    chunk_off = _excl_cumsum(chunk_tot, axis=0)              # chunks, left to right
    cum = (chunk_off[:, None, None] + run_off[:, :, None]).astype(np.float32) + run
    return cum.astype(np.float32).reshape(-1)[:n]


def split(key):
    """jax.random.split(key) -> (new_key, sub_key)."""
    k = c_oracle.key_split(np.asarray(key, np.uint32), 2)
    return k[0].copy(), k[1].copy()


def randint(key, n, minval, maxval):
    """jax.random.randint(key, (n,), minval, maxval) for int32 bounds."""
    k1, k2 = split(key)
    hi = c_oracle.random_bits(k1, n).astype(np.uint64)
    lo = c_oracle.random_bits(k2, n).astype(np.uint64)
    span = np.uint64((maxval - minval) & 0xFFFFFFFF) if maxval > minval else np.uint64(1)
    m = np.uint64(1 << 16) % span
    mult = (m * m) % span
    off = (((hi % span) * mult) & np.uint64(0xFFFFFFFF)) + (lo % span)
    off = (off & np.uint64(0xFFFFFFFF)) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


class State:
    def __init__(self, cap, D, key, bs):
        self.data = np.zeros((cap, D), np.float32)            # replay_buffers.py:99-106
        self.insert_position = 0
        self.sample_position = 0
        self.key = np.asarray(key, np.uint32).copy()
        self.current_sample_indices = np.zeros(bs, np.int32)


def insert(st, update):
    """replay_buffers.py:120-159."""
    data = st.data
    update = np.asarray(update, np.float32)
    if len(update) > len(data):
        raise ValueError("Trying to insert a batch of samples larger than the maximum replay size.")
    position = st.insert_position
    roll = min(0, len(data) - position - len(update))
    if roll:
        data = np.roll(data, roll, axis=0)
    position = position + roll
    data[position:position + len(update)] = update
    st.data = data
    st.insert_position = (position + len(update)) % (len(data) + 1)
    st.sample_position = max(0, st.sample_position + roll)
    return st


def sample_prioritized(st, bs, beta):
    """replay_buffers.py:304-335; jax.random.choice(replace=True, p) is cumsum +
    searchsorted of cum[-1] * (1 - uniform)."""
    st.key, sample_key = split(st.key)
    data = st.data
    n = len(data)
    probs = np.where(np.arange(n) < st.insert_position, data[:, -2], np.float32(0)).astype(np.float32)
    total = chunked_cumsum(probs)[-1]
    probs = (probs / total).astype(np.float32)
    cum = chunked_cumsum(probs)
    u = c_oracle.uniform01(sample_key, bs)
    r = (cum[-1] * (np.float32(1) - u)).astype(np.float32)
    idx = np.searchsorted(cum, r, side="left").astype(np.int32)
    base = (np.float32(st.insert_position) * probs).astype(np.float32) + np.float32(1e-10)
    w = (np.float32(1) / c_oracle.powf(base, beta)).astype(np.float32)
    w = (w / w.max()).astype(np.float32)
    data[:, -1] = w
    st.current_sample_indices = idx
    return st, data[idx % n].copy(), dict(cum=cum, total=total, u=u)


def sample_uniform(st, bs):
    """replay_buffers.py:260-277."""
    st.key, sample_key = split(st.key)
    idx = randint(sample_key, bs, st.sample_position, st.insert_position)
    st.current_sample_indices = idx
    return st, st.data[idx % len(st.data)].copy()


def set_priorities(st, priorities):
    """replay_buffers.py:337-339 (a repeated index keeps its last value)."""
    st.data[st.current_sample_indices, -2] = np.asarray(priorities, np.float32)
    return st


def size(st):
    return st.insert_position - st.sample_position
