"""The N > 1 path on CPU: two gloo ranks shard the lanes, reduce the elapsed time with
MAX and the processed units with SUM, exactly as bench.py does on RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from alphazero_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r, w, _ = parallel.env_rank()
        shard = parallel.shard_lanes(8192, r, w, key=20)
        weak = parallel.shard_lanes(4096, r, w, key=20, weak=True)
        dist.barrier()
        elapsed = parallel.max_over_ranks(1.0 + rank)          # slowest rank decides
        units = parallel.sum_over_ranks(weak.lanes * 64)        # whole-job simulations
        out.put((rank, shard.lanes, shard.lane_offset, shard.key, weak.lanes, elapsed, units))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(out.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, l0, o0, k0, w0, e0, u0), (r1, l1, o1, k1, w1, e1, u1) = results
    assert (l0, o0) == (4096, 0) and (l1, o1) == (4096, 4096)   # contiguous lane blocks
    assert k0 != k1 and k0 == parallel.fold_in(20, 0) and k1 == parallel.fold_in(20, 1)
    assert w0 == w1 == 4096
    assert e0 == e1 == 2.0                                      # MAX over ranks
    assert u0 == u1 == 2 * 4096 * 64                            # SUM over ranks


def test_shard_lanes_rules():
    s = parallel.shard_lanes(16, 3, 4, key=(1, 2))
    assert (s.lanes, s.lane_offset) == (4, 12)
    with pytest.raises(AssertionError):   # alphazero.py:1436: num_envs % device_count == 0
        parallel.shard_lanes(10, 0, 4)
    with pytest.raises(ValueError):
        parallel.shard_lanes(8, 4, 4)
    assert parallel.max_over_ranks(3.5) == 3.5 and parallel.sum_over_ranks(2) == 2.0


def test_trainer_refuses_a_launcher_environment_without_a_process_group(monkeypatch):
    """WORLD_SIZE > 1 without torch.distributed initialised would make every collective of the
    learner a silent identity: the trainer's rank lookup raises instead."""
    import pytest
    from alphazero_amd import parallel
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    with pytest.raises(RuntimeError, match="not initialised"):
        parallel.trainer_ranks()
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    assert parallel.trainer_ranks()[:2] == (0, 1)
