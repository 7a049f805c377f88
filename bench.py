"""Benchmark of the batched-MCTS self-play hot path on MI355X.

One step = one actor step (alphazero.py:287-403) on B lanes per GPU: root forward,
gumbel_muzero_policy with the fused env + MLP + tree search kernel (S simulations),
greedy/sampled action post-processing and the auto-resetting real environment
step.  metric = MCTS simulations/s (B*S node expansions per step), whole job.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started without a launcher and with N > 1 it launches itself: the parent starts the N ranks as a
torch.distributed.run child (before it touches the GPU itself) and exits with the child's code.

Lanes are independent, so ranks shard them with no data-path collective (weak
scaling: B lanes per GPU); the only collectives are the timing barrier and the
max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_EXPANSION = {  # 2 * sum(in*out) over both MLPs (4x32 policy, 5x256 value)
    "CartPole-v1": 533376, "Acrobot-v1": 534592, "MountainCar-v0": 532288}
PEAK_F32_MFMA_TFLOPS = 157.3   # f32-input MFMA, MI355X
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=4096, help="lanes per GPU")
    p.add_argument("--sims", type=int, default=64)
    p.add_argument("--env", default="CartPole-v1")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="leave out the sampled_muzero_policy row (profiling runs: one search kernel only)")
    p.add_argument("--cpu-lanes", type=int, default=0, help="lanes of the CPU baseline sample")
    p.add_argument("--rehearse-on-one-gpu", action="store_true",
                   help="all ranks share cuda:0 and reduce over gloo (checks the N > 1 code path "
                        "on a one-GPU box; not a measurement)")
    p.add_argument("--dry-run", action="store_true",
                   help="launcher, process group (gloo), barriers and reductions only, no GPU work "
                        "(checks the N > 1 plumbing on a machine without GPUs; not a measurement)")
    return p.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) without a launcher's environment: run the N ranks as a
    torch.distributed.run child process and exit with its code.  Called before anything in this
    process initialises the GPU; nothing is exec'ed."""
    # (--standalone: the launcher's own rendezvous picks a free port itself; probing one here
    # and handing its number on would leave a window in which another process can take it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--local-addr", "127.0.0.1",
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))


def make_networks(env_id, seed, device):
    import alphazero_amd as mctx
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    obs, act = mctx.envs.observation_size(env_id), mctx.envs.action_size(env_id)
    pol = mctx.init_mlp((obs, 32, 32, 32, 32, act), generator=g, device=device)
    val = mctx.init_mlp((obs, 256, 256, 256, 256, 256, 1), generator=g, device=device)
    return pol, val


class Actor:
    """alphazero.py:287-403 actor_step with the search on the fused HIP path."""

    def __init__(self, env_id, B, S, device, seed=0, variant="gumbel", hidden_bf16=False):
        import alphazero_amd as mctx
        self.mctx, self.B, self.S, self.variant = mctx, B, S, variant
        self.env = mctx.envs.make(env_id, episode_length=500)
        self.pol, self.val = make_networks(env_id, 1, device)
        self.model = mctx.FusedModel(env_id, self.pol, self.val, gamma=0.99, hidden_bf16=hidden_bf16)
        self.recurrent_fn = mctx.FusedRecurrentFn(self.model)
        self.state = self.env.reset(seed, batch_size=B, device=device)
        self.key = (0, seed)
        self.qtransform = mctx.qtransform_completed_by_mix_value

    def step(self, stats=None):
        mctx = self.mctx
        from alphazero_amd import prng
        self.key, search_key = prng.split(self.key, 2)
        logits, value = self.model.forward(self.state.obs)
        root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=self.state.embedding)
        if self.variant == "sampled":  # the reference trainer's default (alphazero.py:312-326)
            out = mctx.sampled_muzero_policy((), search_key, root, self.recurrent_fn, self.S,
                                             num_samples=100, stats=stats)
        else:
            out = mctx.gumbel_muzero_policy((), search_key, root, self.recurrent_fn, self.S,
                                            max_num_considered_actions=16, qtransform=self.qtransform,
                                            stats=stats)
        action, search_value, log_prob = mctx.actor_post(out.search_tree, out.action_weights,
                                                         out.action, logits, deterministic=False)
        self.env.step_(self.state, action)
        return out


def secondary_sampled(args, B, S, device, steps=10):
    """The same actor step with the reference trainer's default policy, sampled_muzero_policy with
    100 samples (PUCT with Dirichlet noise at the root and tie-break noise in the tree)."""
    actor = Actor(args.env, B, S, device, seed=7, variant="sampled")
    for _ in range(3):
        actor.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ktimes = []
    for _ in range(steps):
        stats = {"record_events": True}
        actor.step(stats)
        ktimes.append(stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = float(np.mean([s["events"][0][0].elapsed_time(s["events"][0][1]) for s in ktimes]))
    return {"value": round(B * S * steps / dt, 1), "unit": "simulations/s", "num_samples": 100,
            "ms_per_step": round(dt / steps * 1e3, 4), "search_kernel_ms_per_launch": round(kt, 4)}


def secondary_bf16(args, B, S, device, steps=10):
    """The headline actor step with the networks' hidden layers on the bfloat16 matrix
    instructions (bfloat16 weights and layer inputs, float32 accumulation, float32 first and last
    layers): NOT the reference's arithmetic, reported beside the float32 headline with how far
    its search results are from the float32 kernel's on the same roots and root noise."""
    import alphazero_amd as mctx
    actor = Actor(args.env, B, S, device, seed=11, hidden_bf16=True)
    ref = Actor(args.env, B, S, device, seed=11)
    # one search each from the same states, prior logits and root Gumbel noise
    logits, value = ref.model.forward(ref.state.obs)
    root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=ref.state.embedding)
    gum = torch.empty_like(logits).exponential_().log_().neg_()
    kw = dict(max_num_considered_actions=16, qtransform=ref.qtransform, root_gumbel=gum)
    a = mctx.gumbel_muzero_policy((), 0, root, actor.recurrent_fn, S, **kw)
    b = mctx.gumbel_muzero_policy((), 0, root, ref.recurrent_fn, S, **kw)
    agree = float((a.action == b.action).float().mean())
    dw = float((a.action_weights - b.action_weights).abs().max())
    dw_mean = float((a.action_weights - b.action_weights).abs().mean())
    dv = float((a.search_tree.node_values[:, 0] - b.search_tree.node_values[:, 0]).abs().max())
    for _ in range(2):
        actor.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ktimes = []
    for _ in range(steps):
        stats = {"record_events": True}
        actor.step(stats)
        ktimes.append(stats)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kt = float(np.mean([s["events"][0][0].elapsed_time(s["events"][0][1]) for s in ktimes]))
    return {"value": round(B * S * steps / dt, 1), "unit": "simulations/s", "dtype": "bf16w/f32acc",
            "ms_per_step": round(dt / steps * 1e3, 4), "search_kernel_ms_per_launch": round(kt, 4),
            "action_agreement_with_f32": round(agree, 4), "max_abs_diff_action_weights": round(dw, 5),
            "mean_abs_diff_action_weights": round(dw_mean, 6),
            "max_abs_diff_root_value": round(dv, 5),
            "note": "hidden layers v_mfma_f32_16x16x32_bf16; first and output layers float32; never the parity path"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC passes (FETCH_SIZE / WRITE_SIZE,
    collected separately under rocprofv3 with the headline workload), or None."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r03_hbm_traffic.json"))).get(
            kernel + "_bytes_per_launch")
    except Exception:
        return None


def pmc_mfma_util(kernel):
    """Matrix-pipe busy share of `kernel` from the committed SQ counter pass, or None."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", "r03_sq_counters.json")))["kernels"]
        return next((round(v["mfma_util"], 4) for k, v in ks.items() if kernel in k), None)
    except Exception:
        return None


def tree_kernel_roofline(B, S, A, device, reps=5):
    """The persistent tree kernel alone (injected evaluations) against the HBM roofline:
    algorithmic bytes/sim = d*(24A+56) + 2E + 4A + 32 with E = 0 (no embedding), K = 1."""
    import alphazero_amd as mctx
    g = torch.Generator(device=device)
    g.manual_seed(2)
    rf = mctx.InjectedRecurrentFn(
        torch.randn((B, S, A), generator=g, device=device),
        torch.rand((B, S, 1), generator=g, device=device) * 2 - 1,
        torch.ones((B, S), device=device), torch.full((B, S), 0.99, device=device))
    root = mctx.RootFnOutput(prior_logits=torch.randn((B, A), generator=g, device=device),
                             value=torch.rand((B, 1), generator=g, device=device) * 2 - 1,
                             embedding=None)
    times, sum_depth = [], 0
    for i in range(reps + 1):
        stats = {"record_events": True}
        mctx.gumbel_muzero_policy((), i, root, rf, S, stats=stats)
        torch.cuda.synchronize()
        if i:
            times.append(stats["events"][0][0].elapsed_time(stats["events"][0][1]) * 1e-3)
            sum_depth = int(stats["sum_depth"].item())
    t = float(np.mean(times))
    alg_bytes = sum_depth * (24 * A + 56) + B * S * (4 * A + 32)
    achieved = alg_bytes / t / 1e9
    return {"kernel": "k_search_wave_aos21", "bound": "hbm", "achieved": round(achieved, 2),
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 5),
            "traffic": pmc_traffic("k_search_wave_aos21"), "ms_per_launch": round(t * 1e3, 4),
            "mean_depth": round(sum_depth / (B * S), 3),
            "sims_per_s": round(B * S / t, 1)}


def cpu_baseline(env_id, S, lanes, threads, runs=5):
    """The plain-C restatement (oracle/, built -O3 for x86-64-v3) of the same fused search, on
    `threads` host threads over `lanes` lanes of the same synthetic workload: one warm-up run,
    then the median of `runs` timed runs."""
    from oracle import runners
    per = max(1, lanes // threads)
    cases = [runners.synthetic_case("gumbel", B=per, S=S, K=1, seed=100 + i, env_id=env_id)
             for i in range(threads)]

    def once():
        t0 = time.perf_counter()
        workers = [threading.Thread(target=runners.run_policy_c, args=(c,)) for c in cases]
        for w in workers:
            w.start()
        for w in workers:
            w.join()
        return time.perf_counter() - t0

    once()
    times = sorted(once() for _ in range(runs))
    dt = times[len(times) // 2]
    n = per * threads
    return {"value": round(n * S / dt, 1), "unit": "simulations/s", "cores": threads, "kind": "port",
            "sample": f"{n} lanes x {S} simulations of the same fused {env_id} Gumbel search "
                      f"(plain-C checker, gcc -O3 x86-64-v3, {threads} threads): median of {runs} runs "
                      f"after one warm-up, {dt:.2f} s each ({sum(times):.1f} s in all)",
            "env_steps_per_s": round(n / dt, 2)}


def dry_run(args, rank, world):
    """The multi-rank plumbing without a GPU: gloo group, the two barriers, max / sum over ranks,
    one JSON line from rank 0."""
    from alphazero_amd import parallel
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    shard = parallel.shard_lanes(args.batch, rank, world, key=1000, weak=True)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0)
    total = parallel.sum_over_ranks(shard.lanes * args.sims * args.steps)
    if rank == 0:
        print(json.dumps({"metric": "MCTS simulations/s", "value": round(total / elapsed, 1),
                          "unit": "simulations/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "data": "none"}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    from alphazero_amd import parallel
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)
    rank, world, local = parallel.env_rank()
    if world != args.gpus:
        raise SystemExit(f"the launcher started {world} rank(s) but --gpus is {args.gpus}: "
                         "--nproc-per-node must equal --gpus")
    if args.dry_run:
        return dry_run(args, rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    red_device = torch.device("cpu") if args.rehearse_on_one_gpu else device

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    S = args.sims
    shard = parallel.shard_lanes(args.batch, rank, world, key=1000, weak=True)
    B = shard.lanes
    actor = Actor(args.env, B, S, device, seed=shard.key[1])
    for _ in range(args.warmup):
        actor.step()
    barrier()
    stats_list = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        stats = {"record_events": True}
        actor.step(stats)
        stats_list.append(stats)
    barrier()
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, device=red_device)
    total_sims = parallel.sum_over_ranks(B * S * args.steps, device=red_device)

    # dominant kernel: the fused persistent search, timed with HIP events on its stream
    ktimes = [s["events"][0][0].elapsed_time(s["events"][0][1]) * 1e-3 for s in stats_list]
    kt = float(np.mean(ktimes))
    flop = FLOP_PER_EXPANSION[args.env] * B * S
    achieved = flop / kt / 1e12
    sum_depth = int(stats_list[-1]["sum_depth"].item())

    if rank == 0:
        line = {
            "metric": "MCTS simulations/s", "value": round(total_sims / elapsed, 1),
            "unit": "simulations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.env} gumbel_muzero_policy actor step, fused env+MLP+tree "
                                   f"search, batch {B} lanes/GPU x {S} simulations, policy MLP 4x32, "
                                   f"value MLP 5x256, random-init weights",
                       "lanes_per_gpu": B, "num_simulations": S, "max_num_considered_actions": 16,
                       "parallelism": f"lanes sharded over {world} GPU(s), no data-path collective"},
            "env_steps_per_s": round(total_sims / S / elapsed, 1),
            "roofline": {"kernel": "k_search_fused_wave", "bound": "mfma", "achieved": round(achieved, 3),
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 5),
                         "traffic": pmc_traffic("k_search_fused_wave"),
                         "mfma_util_pmc": pmc_mfma_util("k_search_fused_wave"),
                         "ms_per_launch": round(kt * 1e3, 4),
                         "mean_depth": round(sum_depth / (B * S), 3)},
        }
        if world == 1:
            line["roofline_tree_kernel"] = tree_kernel_roofline(B, S, 2, device)
            if not args.no_secondary:
                line["secondary"] = {"sampled_muzero_policy": secondary_sampled(args, B, S, device),
                                     "gumbel_muzero_policy_bf16_hidden_layers": secondary_bf16(args, B, S, device)}
            if not args.no_cpu_baseline:
                threads = min(16, os.cpu_count() or 1)
                lanes = args.cpu_lanes or 192 * threads  # six runs, ~15 s of wall time in all
                line["cpu_baseline"] = cpu_baseline(args.env, S, lanes, threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
