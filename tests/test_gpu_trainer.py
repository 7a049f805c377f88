"""End-to-end training steps on the GPU: self-play search -> targets -> replay -> SGD."""
import math

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("policy,use_gae,atoms,per_alpha", [("gumbel", False, 1, 0.0),
                                                           ("sampled", True, 4, 0.6),
                                                           ("muzero", False, 1, -2.0)])
def test_training_steps_run_and_learn_signal(policy, use_gae, atoms, per_alpha):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(env_id="CartPole-v1", num_envs=64, unroll_length=6, num_simulations=16, policy=policy,
                 use_gae=use_gae,
                 num_atoms=atoms, per_alpha=per_alpha, max_replay_size=2048, min_replay_size=0,
                 replay_buffer_batch_size=32, num_minibatches=4, value_hidden=(64, 64),
                 policy_hidden=(32, 32), num_samples=8)
    tr = Trainer(cfg, device="cuda")
    before = [p.clone() for p in tr.optimizer.params]
    m = tr.train(3)
    assert all(math.isfinite(v) for v in m.values()), m
    assert m["buffer_current_size"] == 3 * 6 * 64 and m["training/sps"] > 0
    assert tr.optimizer.count == 3 * cfg.num_minibatches
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.optimizer.params))
    assert tr.stats.count == 3 * 6 * 64
    # the self-play model sees the updated parameters and statistics
    lg, _ = tr.model.forward(tr.env_state.obs)
    from alphazero_amd import learner as L
    ref, _ = L.forward_networks(tr.policy_layers, tr.value_layers, tr.stats.mean, tr.stats.std,
                                tr.env_state.obs)
    torch.testing.assert_close(lg, ref, rtol=1e-4, atol=1e-5)
    ev = tr.evaluate(num_eval_envs=16, episode_length=12)
    assert ev["eval/episode_reward"] > 0 and ev["eval/avg_episode_length"] > 0


@pytest.mark.parametrize("env_id,use_gae", [("Acrobot-v1", False), ("MountainCar-v0", True),
                                            ("Acrobot-v1", True), ("MountainCar-v0", False)])
def test_training_steps_on_acrobot_and_mountaincar_with_prioritized_replay(env_id, use_gae):
    """Acrobot / MountainCar (three actions) with prioritized replay (alpha 0.6, beta 0.4) and
    n-step or GAE targets: the training step runs end to end on the fused search of those
    environments, priorities stay positive and finite and are refreshed by the SGD step."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(env_id=env_id, num_envs=128, unroll_length=6, num_simulations=16, policy="gumbel",
                 use_gae=use_gae, per_alpha=0.6, per_importance_sampling_beta=0.4, max_replay_size=4096,
                 min_replay_size=0, replay_buffer_batch_size=64, num_minibatches=4,
                 value_hidden=(64, 64), policy_hidden=(32, 32), seed=3)
    tr = Trainer(cfg, device="cuda")
    before = [p.clone() for p in tr.optimizer.params]
    m = tr.train(3)
    assert all(math.isfinite(v) for v in m.values()), m
    assert m["buffer_current_size"] == 3 * 6 * 128
    assert any(not torch.equal(a, b) for a, b in zip(before, tr.optimizer.params))
    data = tr.buffer_state.data[:3 * 6 * 128]
    pri, w = data[:, -2], data[:, -1]
    assert torch.isfinite(pri).all() and (pri > 0).all() and pri.unique().numel() > 16
    assert torch.isfinite(w).all() and (w > 0).all() and float(w.max()) <= 1.0
    assert (data[:, tr.layout["reward"][0]] == -1).all()          # both environments pay -1 per step
    # three actions: action weights are a distribution over them, actions in range
    a0, _ = tr.layout["action"]
    assert set(data[:, a0].unique().tolist()) <= {0.0, 1.0, 2.0}
    p0, pw = tr.layout["target_policy_probs"]
    assert pw == 3
    torch.testing.assert_close(data[:, p0:p0 + pw].sum(-1), torch.ones_like(pri), rtol=1e-5, atol=1e-6)


def _dp_worker(rank, world, port, out):
    """One data-parallel rank of the trainer; both ranks share cuda:0 and reduce over gloo
    (on an 8-GPU node the same code runs one rank per GPU over RCCL)."""
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alphazero_amd.trainer import Config, Trainer
        cfg = Config(env_id="CartPole-v1", num_envs=64, unroll_length=4, num_simulations=8, policy="gumbel",
                     use_gae=False, min_replay_size=0,
                     max_replay_size=2048, replay_buffer_batch_size=32, num_minibatches=2,
                     value_hidden=(64, 64), policy_hidden=(32, 32), per_alpha=0.6)
        tr = Trainer(cfg, device="cuda:0")
        obs0 = tr.env_state.obs.clone()
        m = tr.train(2)
        flat = torch.cat([p.detach().reshape(-1) for p in tr.optimizer.params]).cpu()
        out.put((rank, tr.lanes, tr.key, float(m["total_loss"]), m["training/sps"], tr.env_steps,
                 float(tr.stats.count), flat.numpy(), tr.stats.mean.cpu().numpy(), obs0.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_two_rank_trainer_keeps_replicas_identical():
    """alphazero.py:1193 (pmap over devices): every rank plays its own lanes with its own key,
    and after the gradient / statistics all-reduces the replicas hold the same parameters."""
    import socket
    import numpy as np
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((out.get(timeout=240) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    assert a[1] == b[1] == 32 and a[2] != b[2]                 # lanes split, keys folded per rank
    assert not np.array_equal(a[9], b[9])                        # different self-play streams
    assert math.isfinite(a[3]) and math.isfinite(b[3])
    assert a[5] == b[5] == 2 * 4 * 64                            # env steps counted over the whole job
    assert a[6] == b[6] == 2 * 4 * 64                            # statistics merged over both ranks
    assert np.array_equal(a[7], b[7]), np.abs(a[7] - b[7]).max()  # replicas bit-identical
    assert np.array_equal(a[8], b[8])


@pytest.mark.parametrize("use_gae,atoms", [(False, 1), (True, 1), (False, 4)])
def test_training_step_plumbing_vs_restatement(use_gae, atoms):
    """The assembly of the training step (transition record -> targets -> priorities -> replay rows
    -> sample -> SGD -> priority refresh with the pre-update networks, alphazero.py:385-403,
    1104-1180) against its NumPy restatement in float64, with the search outputs and environment
    transitions of the run injected: rows, priorities and refreshed priorities to 1e-5."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd import learner as L
    from alphazero_amd import prng
    from alphazero_amd.trainer import Config, Trainer
    from oracle import trainer_numpy as tn
    T, B = 7, 48
    cfg = Config(env_id="CartPole-v1", num_envs=B, unroll_length=T, num_simulations=8, policy="gumbel",
                 use_gae=use_gae, num_atoms=atoms, per_alpha=0.6, per_importance_sampling_beta=0.4,
                 max_replay_size=1024, min_replay_size=0, replay_buffer_batch_size=32, num_minibatches=4,
                 value_hidden=(64, 64), policy_hidden=(32, 32), episode_length=5, seed=11)
    tr = Trainer(cfg, device="cuda")
    # a few plain steps first: non-trivial statistics, episode ends (length 5) inside the unroll
    tr.train(2)
    n0 = int(tr.replay.size(tr.buffer_state))
    tr.trace = []
    k1, k2, _ = prng.split(tr.key, 3)
    flat = tr.collect(k1, k2)
    steps = [{k: v.double().cpu().numpy() for k, v in s.items()} for s in tr.trace]
    tr.trace = None
    assert len(steps) == T + 1
    ref = tn.assemble(steps[:T], steps[T], use_gae=use_gae, gae_lambda=cfg.gae_lambda,
                      n_step_n=cfg.n_step_n, gamma=cfg.n_step_gamma, per_alpha=cfg.per_alpha)
    assert float(ref["truncation"].sum()) > 0 and float((1 - ref["discount"]).sum()) > 0
    ref_rows = tn.rows(ref)
    got_rows = tr.buffer_state.data[n0:n0 + T * B].double().cpu().numpy()
    assert got_rows.shape == ref_rows.shape == (T * B, tr.layout["_width"])
    np.testing.assert_allclose(got_rows[:, :-1], ref_rows[:, :-1], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(flat["priority"].double().cpu().numpy(), ref["priority"], rtol=1e-5, atol=1e-7)

    # the learner half: sample, SGD, refresh.  Pre-update networks and statistics are snapshotted.
    mean0, std0 = tr.stats.mean.double().cpu().numpy(), tr.stats.std.double().cpu().numpy()
    # (the refresh runs on the self-play model, which holds the statistics of before this step)
    model_mean, model_std = mean0.copy(), std0.copy()
    val0 = [(w.detach().double().cpu().numpy(), b.detach().double().cpu().numpy())
            for w, b in tr.value_layers]
    before = tr.buffer_state.data.double().cpu().numpy()
    tr.learn(flat)
    idx = tr.buffer_state.current_sample_indices.cpu().numpy()
    after = tr.buffer_state.data.double().cpu().numpy()
    sampled = before[idx]
    def col(name, keep_axis=False):
        o, w = tr.layout[name]
        x = sampled[:, o:o + w]
        return x if keep_axis else x[:, 0]
    batch = dict(observation=col("observation", True), bootstrap_value=col("bootstrap_value", True),
                 value_prefix_target=col("value_prefix_target"),
                 bootstrap_discount=col("bootstrap_discount"))
    ref_pri = tn.refreshed_priorities(batch, val0, model_mean, model_std, cfg.per_alpha)
    last = {int(i): p for p, i in enumerate(idx)}          # a repeated index keeps its last write
    rows_i = np.array(sorted(last))
    want = np.array([ref_pri[last[int(i)]] for i in rows_i])
    np.testing.assert_allclose(after[rows_i, -2], want, rtol=2e-5, atol=1e-7)
    untouched = np.setdiff1d(np.arange(after.shape[0]), rows_i)
    assert np.array_equal(after[untouched, -2], before[untouched, -2])
    # rows themselves are not modified by the learner
    assert np.array_equal(after[:, :-2], before[:, :-2])


def test_prefill_is_one_unroll_and_updates_the_normaliser():
    """alphazero.py:845, 1224-1296, 1427-1431: the prefill is ONE unroll of
    ceil(min_replay_size / num_envs) actor steps with one bootstrap search; afterwards the
    observation statistics hold the prefilled observations and self-play uses them; the trainer's
    key advances by one 2-way split.  train() prefills only while the queue is below
    min_replay_size.  (Parity unpinned: the reference has no test of this path.)"""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd import learner as L
    from alphazero_amd import prng
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(env_id="CartPole-v1", num_envs=64, unroll_length=4, num_simulations=8, policy="gumbel",
                 use_gae=True, max_replay_size=2048, min_replay_size=1000, replay_buffer_batch_size=32,
                 num_minibatches=4, value_hidden=(64, 64), policy_hidden=(32, 32), seed=5)
    tr = Trainer(cfg, device="cuda")
    key0 = tr.key
    tr.trace = []
    assert tr.prefill() == 16                                   # ceil(1000 / 64)
    assert len(tr.trace) == 16 + 1                              # one unroll + ONE bootstrap search
    tr.trace = None
    assert int(tr.replay.size(tr.buffer_state)) == 16 * 64
    assert tr.env_steps == 16 * 64 and float(tr.stats.count) == 16 * 64
    assert tr.key == prng.split(key0, 2)[1]
    assert tr.optimizer.count == 0                              # no learning during the prefill
    lg, _ = tr.model.forward(tr.env_state.obs)                  # self-play sees the new statistics
    ref, _ = L.forward_networks(tr.policy_layers, tr.value_layers, tr.stats.mean, tr.stats.std,
                                tr.env_state.obs)
    torch.testing.assert_close(lg, ref, rtol=1e-4, atol=1e-5)
    assert float(tr.stats.std.max()) < 0.9                      # no longer the initial ones
    m = tr.train(1)                                             # the queue is full enough: no second prefill
    assert m["buffer_current_size"] == 16 * 64 + 4 * 64
    tr2 = Trainer(cfg, device="cuda")                           # train() on an empty queue prefills first
    m2 = tr2.train(1)
    assert m2["buffer_current_size"] == 16 * 64 + 4 * 64 and tr2.env_steps == tr.env_steps


def test_evaluate_follows_the_eval_wrapper_accumulation():
    """alphazero.py:433-508 with envs/brax_wrappers.py:153-197: per evaluation lane the return and
    the length of its FIRST episode, against the NumPy restatement on the recorded transitions
    (episodes of at most 9 steps inside an unroll of 30, so lanes auto-reset several times)."""
    import numpy as np
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    from oracle import trainer_numpy as tn
    cfg = Config(env_id="CartPole-v1", num_envs=16, unroll_length=2, num_simulations=8, policy="gumbel",
                 min_replay_size=0, max_replay_size=256, replay_buffer_batch_size=8, num_minibatches=2,
                 value_hidden=(32, 32), policy_hidden=(16, 16), episode_length=9, num_eval_envs=48, seed=2)
    tr = Trainer(cfg, device="cuda")
    state_before = tr.env_state
    tr.trace = []
    ev = tr.evaluate(episode_length=30)
    steps = [{k: v.double().cpu().numpy() for k, v in s.items() if k in ("reward", "done", "steps")}
             for s in tr.trace]
    tr.trace = None
    assert len(steps) == 30 and steps[0]["reward"].shape == (48,)
    assert tr.env_state is state_before                        # the training environments are untouched
    ref = tn.eval_metrics(steps)
    assert np.array_equal(ev["episode_reward"].cpu().numpy(), ref["episode_reward"].astype(np.float32))
    assert np.array_equal(ev["episode_steps"].cpu().numpy(), ref["episode_steps"].astype(np.float32))
    assert np.array_equal(ev["active_episodes"].cpu().numpy(), ref["active_episodes"].astype(np.float32))
    assert ev["eval/episode_reward"] == pytest.approx(ref["eval/episode_reward"], rel=1e-6)
    assert ev["eval/episode_reward_std"] == pytest.approx(ref["eval/episode_reward_std"], rel=1e-5, abs=1e-6)
    assert ev["eval/avg_episode_length"] == pytest.approx(ref["eval/avg_episode_length"], rel=1e-6)
    # every lane finished its first episode within 9 steps (EpisodeWrapper truncates there)
    assert float(ev["active_episodes"].sum()) == 0 and float(ev["episode_steps"].max()) <= 9
    assert ev["eval/sps"] > 0


def test_training_step_at_the_largest_configuration():
    """One training step at the size of the largest baseline configuration (16 384 lanes x 256
    simulations, Gumbel policy; unroll of one actor step and a small SGD batch to keep it short):
    finite metrics, one row per lane in the queue."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(env_id="CartPole-v1", num_envs=16384, unroll_length=1, num_simulations=256, policy="gumbel",
                 use_gae=True, min_replay_size=0, max_replay_size=32768, replay_buffer_batch_size=64,
                 num_minibatches=2, per_alpha=0.6, seed=4)
    tr = Trainer(cfg, device="cuda")
    m = tr.training_step()
    torch.cuda.synchronize()
    assert all(math.isfinite(float(v)) for v in m.values()), m
    assert int(m["buffer_current_size"]) == 16384 and tr.env_steps == 16384
    assert float(tr.stats.count) == 16384


@pytest.mark.parametrize("atoms,per_alpha", [(1, 0.0), (4, 0.6)])
def test_graph_replayed_minibatch_updates_equal_the_eager_ones(atoms, per_alpha):
    """Config.graph_sgd replays the captured minibatch update (loss, gradient, clip + Adam with the
    update count and the learning-rate schedule on the device); two trainers from one seed, one
    eager and one replaying, must hold the same parameters, moments and metrics after several
    training steps (annealed learning rate, so the schedule matters)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd.trainer import Config, Trainer
    kw = dict(env_id="CartPole-v1", num_envs=64, unroll_length=4, num_simulations=8, policy="gumbel",
              num_atoms=atoms, per_alpha=per_alpha, max_replay_size=2048, min_replay_size=0,
              replay_buffer_batch_size=32, num_minibatches=4, value_hidden=(64, 64), policy_hidden=(32, 32),
              anneal_lr=True, total_timesteps=64 * 4 * 10)
    eager, graphed = Trainer(Config(graph_sgd=False, **kw), device="cuda"), Trainer(Config(graph_sgd=True, **kw), device="cuda")
    for _ in range(3):
        me, mg = eager.training_step(), graphed.training_step()
        for k in me:
            torch.testing.assert_close(torch.as_tensor(mg[k]).float().cpu(), torch.as_tensor(me[k]).float().cpu(),
                                       rtol=1e-5, atol=1e-7, msg=k)
    assert graphed._graphed is not None and graphed._graphed.graph is not None and eager._graphed is None
    assert eager.optimizer.count == graphed.optimizer.count == 12
    assert float(graphed._graphed.count) == 12.0
    for a, b in zip(eager.optimizer.params + eager.optimizer.m + eager.optimizer.v,
                    graphed.optimizer.params + graphed.optimizer.m + graphed.optimizer.v):
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-8)


def test_minibatch_permutation_on_the_device_equals_the_host_one():
    """learner.permutation on the GPU (az_random_sort_keys + stable sorts) = prng.permutation (NumPy),
    for one and two rounds of sorting."""
    import numpy as np
    from alphazero_amd import learner as L, prng
    for key, n in (((0, 3), 1), ((7, 8), 37), ((0, 9), 1625), ((1, 2), 4096), ((5, 5), 16384)):
        got = L.permutation(key, n, "cuda").cpu().numpy()
        assert np.array_equal(got, prng.permutation(key, n)), (key, n)
