"""Sharding of the self-play hot path over the GPUs of one node.

The reference data-parallelises with `jax.pmap` (alphazero.py:1193): every device
owns `num_envs // device_count` environments, their search trees and their share
of the replay buffer (alphazero.py:893-896, 935-943), and derives its own random
stream by folding the process index into the key (alphazero.py:870).  Lanes never
exchange data during self-play, so here each rank (one process per GPU) simply owns
a contiguous block of lanes; the only collectives of the search path are the timing
barrier and a max-reduction of the elapsed time, both through torch.distributed
(RCCL on GPUs, gloo on CPU).
"""
import dataclasses
import os

from . import prng


@dataclasses.dataclass(frozen=True)
class Shard:
    rank: int
    world: int
    lanes: int        # lanes owned by this rank
    lane_offset: int  # global index of the rank's first lane
    key: tuple        # this rank's random key


def env_rank():
    """(rank, world, local_rank) from the torch.distributed launcher's environment."""
    return (int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)),
            int(os.environ.get("LOCAL_RANK", 0)))


def trainer_ranks():
    """(rank, world, local_rank) for a data-parallel learner: the process group decides, and a
    launcher environment that announces several ranks without an initialised group is an error --
    every collective would silently be the identity and the replicas would drift apart while
    their steps are still counted as one job."""
    import torch.distributed as dist
    rank, world, local = env_rank()
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), local
    if world > 1:
        raise RuntimeError(f"WORLD_SIZE={world} but torch.distributed is not initialised: call "
                           "torch.distributed.init_process_group() before building the Trainer")
    return 0, 1, local


def shard_lanes(num_envs, rank, world, key=0, *, weak=False):
    """Strong scaling (`weak=False`): `num_envs` global lanes split as the reference does,
    `num_envs // world` per rank, which requires divisibility (alphazero.py:893, 1436).
    Weak scaling: every rank owns `num_envs` lanes.  The rank's key is fold_in(key, rank)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} for world size {world}")
    if weak:
        lanes = int(num_envs)
    else:
        if num_envs % world:
            raise AssertionError(f"num_envs={num_envs} must be divisible by the {world} devices")
        lanes = num_envs // world
    return Shard(rank=rank, world=world, lanes=lanes, lane_offset=rank * lanes,
                 key=fold_in(key, rank))


def fold_in(key, data):
    """jax.random.fold_in: threefry block of the key over the counter (0, data)."""
    return prng.threefry2x32(prng.as_key(key), (0, int(data) & 0xFFFFFFFF))


def max_over_ranks(value, device=None):
    """Max of a Python float over all ranks (identity without a process group)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
