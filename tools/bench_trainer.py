"""End-to-end training throughput (self-play search + replay + SGD) of the data-parallel
Trainer, in environment steps per second (the reference's `training/sps`,
alphazero.py:1210-1212), with the share of each part of a training step.

  python tools/bench_trainer.py                      # one GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      tools/bench_trainer.py                         # N GPUs, gradients averaged over RCCL
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphazero_amd import parallel  # noqa: E402
from alphazero_amd.trainer import Config, Trainer  # noqa: E402

CASES = {
    # the reference's own CartPole configuration, scaled to one GPU's batch
    "cartpole_sampled": dict(env_id="CartPole-v1", policy="sampled", num_envs=4096, num_simulations=64,
                             num_samples=16, per_alpha=0.0),
    # Acrobot / MountainCar with prioritized replay and n-step / GAE targets
    "acrobot_per_nstep": dict(env_id="Acrobot-v1", policy="gumbel", num_envs=8192, num_simulations=64,
                              per_alpha=0.6, per_importance_sampling_beta=0.4),
    "mountaincar_per_gae": dict(env_id="MountainCar-v0", policy="gumbel", num_envs=8192,
                                num_simulations=64, per_alpha=0.6, use_gae=True),
    # sampled search with a quantile-regression value head
    "cartpole_sampled_qr": dict(env_id="CartPole-v1", policy="sampled", num_envs=4096, num_simulations=64,
                                num_samples=16, num_atoms=8),
    # MinAtar Breakout with the convolutional torso: the search runs on the per-simulation kernels,
    # the networks through library convolutions (rows of 1616 floats: a smaller replay)
    "breakout_torso": dict(env_id="Breakout-MinAtar", policy="gumbel", num_envs=2048, num_simulations=32,
                           normalize_observations=False, max_replay_size=1 << 17),
}


def timed(fn, device):
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(device)
    return out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="all", choices=["all"] + list(CASES))
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    rank, world, local = parallel.env_rank()
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    for name, kw in CASES.items():
        if args.case not in ("all", name):
            continue
        kw = dict(kw)
        kw["num_envs"] *= world          # weak scaling: the same lanes per GPU
        kw["max_replay_size"] = kw.get("max_replay_size", 1 << 20) * world
        cfg = Config(unroll_length=8, replay_buffer_batch_size=1024, num_minibatches=4, **kw)
        tr = Trainer(cfg)
        for _ in range(args.warmup):
            tr.training_step()
        # shares: time the collection alone, then whole steps
        k = (0, 12345)
        _, t_collect = timed(lambda: tr.collect(k, k), tr.device)
        m = tr.train(args.steps)
        step_s = cfg.unroll_length * cfg.num_envs / m["training/sps"]
        if rank == 0:
            print(json.dumps({
                "case": name, "n_gpus": world, "env_steps_per_s": round(m["training/sps"], 1),
                "simulations_per_s": round(m["training/sps"] * cfg.num_simulations * (cfg.unroll_length + 1)
                                           / cfg.unroll_length, 1),
                "ms_per_training_step": round(step_s * 1e3, 2),
                "collect_share": round(t_collect / step_s, 3),
                "lanes_per_gpu": cfg.num_envs // world, "num_simulations": cfg.num_simulations,
                "unroll_length": cfg.unroll_length, "sgd_batch": cfg.replay_buffer_batch_size * cfg.num_minibatches,
                "total_loss": round(m.get("total_loss", float("nan")), 5)}), flush=True)
        del tr
        torch.cuda.empty_cache()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
