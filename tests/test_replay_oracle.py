"""CPU checks of the replay-queue restatement (oracle/replay_numpy.py) against facts that
follow from the reference's source (replay_buffers.py) -- the reference has no tests or
fixtures for this path."""
import numpy as np
import pytest

from oracle import replay_numpy as rp


def rows(n, D, tag, prio=1.0):
    x = np.full((n, D), tag, np.float32)
    x[:, -2] = prio
    x[:, -1] = 0
    return x


def test_fifo_insert_rolls_out_the_oldest_rows():
    # replay_buffers.py:141-153: the queue keeps the newest `cap` rows in arrival order
    st = rp.State(8, 5, (0, 1), 4)
    for tag in (1, 2):
        rp.insert(st, rows(3, 5, tag))
    assert (st.insert_position, st.sample_position) == (6, 0)
    rp.insert(st, rows(3, 5, 3))          # needs one row of room: roll by -1
    assert st.data[:, 0].tolist() == [1, 1, 2, 2, 2, 3, 3, 3]
    assert st.insert_position == 8 and rp.size(st) == 8
    rp.insert(st, rows(3, 5, 4))
    assert st.data[:, 0].tolist() == [2, 2, 3, 3, 3, 4, 4, 4]
    with pytest.raises(ValueError):
        rp.insert(st, rows(9, 5, 5))


def test_insert_position_wraps_modulo_cap_plus_one():
    # replay_buffers.py:152: (position + len(update)) % (len(data) + 1)
    st = rp.State(4, 3, (0, 1), 2)
    rp.insert(st, rows(4, 3, 1))
    assert st.insert_position == 4
    rp.insert(st, rows(1, 3, 2))
    assert st.insert_position == 4 and st.data[:, 0].tolist() == [1, 1, 1, 2]


def test_chunked_cumsum_is_a_prefix_sum():
    rng = np.random.default_rng(0)
    for n in (1, 15, 16, 17, 4095, 4096, 4097, 3 * 4096 + 5):
        x = rng.random(n).astype(np.float32)
        c = rp.chunked_cumsum(x)
        np.testing.assert_allclose(c, np.cumsum(x, dtype=np.float64), rtol=2e-6)
        assert c[0] == x[0] and (np.diff(c) >= 0).all()
    # integers that float32 represents exactly: any association order gives the same sums
    x = rng.integers(0, 8, 20000).astype(np.float32)
    assert np.array_equal(rp.chunked_cumsum(x), np.cumsum(x, dtype=np.float32))


def test_prioritized_sampling_follows_the_inverse_cdf():
    st = rp.State(64, 6, (0, 7), 4096)
    rp.insert(st, rows(16, 6, 1, prio=1.0))
    rp.insert(st, rows(16, 6, 2, prio=3.0))
    st, batch, aux = rp.sample_prioritized(st, 4096, beta=1.0)
    idx = st.current_sample_indices
    assert idx.min() >= 0 and idx.max() < 32          # empty rows have probability 0
    assert np.array_equal(idx, np.searchsorted(aux["cum"], aux["cum"][-1] * (1 - aux["u"])))
    frac = (idx >= 16).mean()
    assert abs(frac - 0.75) < 0.03                     # priorities 3 : 1
    # replay_buffers.py:329-330: weights of ALL rows; the empty rows (p = 0) give
    # 1e10^beta and hold the maximum while the queue is not full
    w = st.data[:, -1]
    assert w[32:].max() == 1.0 and (w[:32] < 1e-6).all()
    assert np.array_equal(batch, st.data[idx])


def test_beta_zero_gives_unit_weights_and_full_queue_normalises_by_rarest():
    st = rp.State(32, 4, (1, 2), 8)
    p = np.linspace(1, 4, 32).astype(np.float32)
    up = rows(32, 4, 1)
    up[:, -2] = p
    rp.insert(st, up)
    st, _, _ = rp.sample_prioritized(st, 8, beta=0.0)
    assert (st.data[:, -1] == 1.0).all()
    st, _, _ = rp.sample_prioritized(st, 8, beta=1.0)
    w = st.data[:, -1]
    assert w.argmax() == 0 and w[0] == 1.0
    np.testing.assert_allclose(w, p[0] / p, rtol=1e-5)


def test_set_priorities_last_write_wins():
    st = rp.State(8, 4, (0, 3), 4)
    rp.insert(st, rows(8, 4, 1))
    st.current_sample_indices = np.array([2, 5, 2, 7], np.int32)
    rp.set_priorities(st, [10, 20, 30, 40])
    assert st.data[[2, 5, 7], -2].tolist() == [30, 20, 40]


def test_randint_range_and_split():
    k = np.array([0, 11], np.uint32)
    r = rp.randint(k, 5000, 3, 10)
    assert r.min() == 3 and r.max() == 9
    assert np.abs(np.bincount(r - 3) / 5000 - 1 / 7).max() < 0.03
    assert (rp.randint(k, 10, 5, 5) == 5).all()       # empty span returns minval
    st = rp.State(16, 4, k, 8)
    rp.insert(st, rows(10, 4, 1))
    st, b = rp.sample_uniform(st, 8)
    assert st.current_sample_indices.max() < 10 and b.shape == (8, 4)
    assert not np.array_equal(st.key, k)
