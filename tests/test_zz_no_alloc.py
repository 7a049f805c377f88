"""The entry point that allocates nothing (az_replay_sample_prioritized_noalloc) against the
entry point it is a variant of (az_replay_sample_prioritized), through the C ABI: the same
queue contents and key must give bit-identical indices, batches and per-row weights, on two
calls in a row, the second on a smaller queue (the library's scratch array keeps the first
call's contents), and a queue beyond the stated maximum must be refused."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _queue(torch, nat, cap, D, n_rows, seed, with_scratch):
    """A filled queue through the C ABI: (descriptor, state, tensors kept alive)."""
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed)
    t = dict(rows=torch.zeros((cap, D), device=dev), priority=torch.zeros(cap, device=dev),
             weight=torch.zeros(cap, device=dev),
             owner=torch.full((cap,), -1, dtype=torch.int32, device=dev))
    if with_scratch:
        t["cum"] = torch.zeros(cap, device=dev)
        t["partial"] = torch.zeros(int(nat.lib().az_replay_partial_floats(cap)), device=dev)
    desc = nat.Replay(nat.ptr(t["rows"]), nat.ptr(t["priority"]), nat.ptr(t["weight"]),
                      nat.ptr(t.get("cum")), nat.ptr(t.get("partial")), nat.ptr(t["owner"]), cap, D)
    st = nat.ReplayState()
    # init / insert check every pointer of the descriptor: run them on one that has scratch
    full = desc
    if not with_scratch:
        t["_cum"], t["_partial"] = torch.zeros(cap, device=dev), torch.zeros(
            int(nat.lib().az_replay_partial_floats(cap)), device=dev)
        full = nat.Replay(nat.ptr(t["rows"]), nat.ptr(t["priority"]), nat.ptr(t["weight"]),
                          nat.ptr(t["_cum"]), nat.ptr(t["_partial"]), nat.ptr(t["owner"]), cap, D)
    nat.check(nat.lib().az_replay_init(C.byref(full), C.byref(st), nat.stream_ptr()))
    up = rng.standard_normal((n_rows, D)).astype(np.float32)
    up[:, -2] = (rng.random(n_rows).astype(np.float32) + np.float32(1e-3)) ** 2
    up[:, -1] = 0
    upd = torch.from_numpy(up).to(dev)
    nat.check(nat.lib().az_replay_insert(C.byref(full), C.byref(st), nat.ptr(upd), n_rows, nat.stream_ptr()))
    return desc, st, t


def _sample(torch, nat, fn, desc, st, key, beta, bs, D):
    idx = torch.empty(bs, dtype=torch.int32, device="cuda")
    batch = torch.empty((bs, D), device="cuda")
    k = (C.c_uint32 * 2)(*key)
    nat.check(fn(C.byref(desc), C.byref(st), C.byref(k), beta, bs, nat.ptr(idx), nat.ptr(batch),
                 nat.stream_ptr()))
    torch.cuda.synchronize()
    return idx.cpu().numpy(), batch.cpu().numpy()


def test_noalloc_sampling_equals_the_allocating_entry_point():
    import torch
    from alphazero_amd import _native as nat
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lib = nat.lib()
    # first call: several scan chunks and a partly filled queue; second call: a smaller queue
    for cap, D, n_rows, bs, seed in ((20000, 9, 15000, 700, 1), (3000, 5, 3000, 64, 2)):
        ref_desc, ref_st, ref_t = _queue(torch, nat, cap, D, n_rows, seed, True)
        new_desc, new_st, new_t = _queue(torch, nat, cap, D, n_rows, seed, False)   # cum / partial NULL
        for key in ((3, 4), (5, 6)):
            ri, rb = _sample(torch, nat, lib.az_replay_sample_prioritized, ref_desc, ref_st, key, 0.6, bs, D)
            ni, nb = _sample(torch, nat, lib.az_replay_sample_prioritized_noalloc, new_desc, new_st, key,
                             0.6, bs, D)
            assert np.array_equal(ri, ni)
            assert np.array_equal(rb, nb)
            assert np.array_equal(ref_t["weight"].cpu().numpy(), new_t["weight"].cpu().numpy())
        assert len(set(ri.tolist())) > 1


def test_noalloc_sampling_allocates_no_device_memory_and_covers_the_maximum_capacity():
    """A queue of exactly the stated maximum (2^20 rows: every chunk slot of the library's scratch
    array, up to its last words, is in use) against the allocating entry point, and the free
    device memory before and after a call, which must not move (the outputs are allocated by
    the test beforehand; one earlier call has looked the scratch address up)."""
    import torch
    from alphazero_amd import _native as nat
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    lib = nat.lib()
    cap, D, n_rows, bs = 1 << 20, 3, (1 << 20) - 5, 512
    ref_desc, ref_st, ref_t = _queue(torch, nat, cap, D, n_rows, 7, True)
    new_desc, new_st, new_t = _queue(torch, nat, cap, D, n_rows, 7, False)
    ri, rb = _sample(torch, nat, lib.az_replay_sample_prioritized, ref_desc, ref_st, (9, 9), 0.5, bs, D)
    ni, nb = _sample(torch, nat, lib.az_replay_sample_prioritized_noalloc, new_desc, new_st, (9, 9), 0.5, bs, D)
    assert np.array_equal(ri, ni) and np.array_equal(rb, nb)
    assert np.array_equal(ref_t["weight"].cpu().numpy(), new_t["weight"].cpu().numpy())
    assert int(ni.max()) > cap - cap // 64                      # draws reach the last chunks
    idx = torch.empty(bs, dtype=torch.int32, device="cuda")
    batch = torch.empty((bs, D), device="cuda")
    k = (C.c_uint32 * 2)(11, 12)
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    nat.check(lib.az_replay_sample_prioritized_noalloc(C.byref(new_desc), C.byref(new_st), C.byref(k), 0.5, bs,
                                                       nat.ptr(idx), nat.ptr(batch), nat.stream_ptr()))
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before


def test_noalloc_sampling_refuses_queues_beyond_its_scratch():
    import torch
    from alphazero_amd import _native as nat
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    cap = (1 << 20) + 1
    dummy = torch.zeros(8, device="cuda")
    desc = nat.Replay(nat.ptr(dummy), nat.ptr(dummy), nat.ptr(dummy), None, None, nat.ptr(dummy), cap, 4)
    st = nat.ReplayState()
    k = (C.c_uint32 * 2)(1, 2)
    rc = nat.lib().az_replay_sample_prioritized_noalloc(C.byref(desc), C.byref(st), C.byref(k), 1.0, 4,
                                                        nat.ptr(dummy), nat.ptr(dummy), nat.stream_ptr())
    assert rc == -2
    assert b"at most" in nat.lib().az_last_error()
