"""MinAtar Breakout on the GPU (include/az_minatar.h, alphazero_amd/minatar.py) against the NumPy
restatement of gymnax's MinBreakout under the reference's wrappers, the convolutional torso
against a NumPy evaluation, and a Gumbel search around both on the per-simulation kernels against
the C checker fed with the recorded evaluations."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch


def test_real_environment_steps_equal_the_restatement():
    """Random play on 48 lanes through many short episodes: states, observations, rewards, done /
    truncation flags, step counters and keys after every step, bit for bit (integer game)."""
    torch = _torch()
    from alphazero_amd import minatar
    from oracle import minatar_numpy as bk
    B, T = 48, 160
    env = minatar.BreakoutEnv(episode_length=23, max_steps_in_episode=17)
    keys = np.random.default_rng(1).integers(0, 2 ** 32, (B, 2), dtype=np.uint64).astype(np.uint32)
    st = env.reset(torch.from_numpy(keys.view(np.int32)).cuda())
    lanes = [bk.Wrapped(keys[b], episode_length=23, max_steps_in_episode=17) for b in range(B)]
    assert np.array_equal(st.state.cpu().numpy(), np.stack([l.state.words() for l in lanes]))
    assert np.array_equal(st.obs.cpu().numpy(), np.stack([l.obs for l in lanes]))
    assert np.array_equal(st.rng.cpu().numpy().view(np.uint32), np.stack([l.rng for l in lanes]))
    rng = np.random.default_rng(2)
    seen_done = seen_trunc = seen_reward = 0
    for t in range(T):
        a = rng.integers(0, 3, B).astype(np.int32)
        st = env.step(st, torch.from_numpy(a).cuda())
        for b in range(B):
            lanes[b].step(a[b])
        got = {k: getattr(st, k).cpu().numpy() for k in ("state", "obs", "real_obs", "reward", "done",
                                                         "truncation", "steps")}
        assert np.array_equal(got["state"], np.stack([l.state.words() for l in lanes])), t
        assert np.array_equal(got["obs"], np.stack([l.obs for l in lanes])), t
        assert np.array_equal(got["real_obs"], np.stack([l.real_obs for l in lanes])), t
        for k, attr in (("reward", "reward"), ("done", "done"), ("truncation", "truncation"), ("steps", "steps")):
            assert np.array_equal(got[k], np.array([getattr(l, attr) for l in lanes], np.float32)), (t, k)
        assert np.array_equal(st.rng.cpu().numpy().view(np.uint32), np.stack([l.rng for l in lanes])), t
        seen_done += int(got["done"].sum())
        seen_trunc += int(got["truncation"].sum())
        seen_reward += int(got["reward"].sum())
    assert seen_done > B and seen_trunc > 0 and seen_reward > 0


def test_model_step_on_crafted_states():
    """The branches random play seldom reaches: brick hits with and without the strike flag, the
    ceiling, both paddle bounces, a miss, brick respawn on an empty map, the time limit, and a
    ball that has left the grid after a terminal step (gathers clamp, scatters are dropped)."""
    torch = _torch()
    from alphazero_amd import minatar
    from oracle import minatar_numpy as bk
    rng = np.random.default_rng(3)
    states, actions = [], []
    for _ in range(600):
        s = bk.EnvState()
        s.ball_y, s.ball_x = int(rng.integers(0, 12)), int(rng.integers(0, 10))
        s.ball_dir, s.pos = int(rng.integers(0, 4)), int(rng.integers(0, 10))
        s.last_y, s.last_x = int(rng.integers(0, 10)), int(rng.integers(0, 10))
        s.strike = bool(rng.integers(0, 2))
        s.time = int(rng.choice([0, 5, 998, 999, 1000]))
        s.terminal = bool(rng.integers(0, 2))
        mode = rng.integers(0, 4)
        if mode == 1:
            s.brick_map[1:4, :] = 1
        elif mode == 2:
            s.brick_map[:] = rng.integers(0, 2, (10, 10))
        elif mode == 3:
            s.brick_map[int(rng.integers(0, 9)), int(rng.integers(0, 10))] = 1
        states.append(s)
        actions.append(int(rng.integers(0, 3)))
    env = minatar.BreakoutEnv()
    packed = torch.from_numpy(np.stack([s.words() for s in states])).cuda()
    out, obs, reward, done, trunc = env.model_step(packed, torch.tensor(actions, dtype=torch.int32).cuda())
    ref = [bk.step_env(s, a) for s, a in zip(states, actions)]
    assert np.array_equal(out.cpu().numpy(), np.stack([r[1].words() for r in ref]))
    assert np.array_equal(obs.cpu().numpy(), np.stack([r[0] for r in ref]))
    assert np.array_equal(reward.cpu().numpy(), np.array([r[2] for r in ref], np.float32))
    assert np.array_equal(done.cpu().numpy(), np.array([r[3] for r in ref], np.float32))
    assert np.array_equal(trunc.cpu().numpy(), np.array([r[4] for r in ref], np.float32))
    assert done.sum() > 0 and reward.sum() > 0 and trunc.sum() > 0


def _torso_numpy(torso, obs):
    """networks.py:95-130 in NumPy float64 (VALID convolutions as explicit window sums)."""
    x = obs.astype(np.float64)
    for w, b in torso.conv:
        w, b = w.cpu().numpy().astype(np.float64), b.cpu().numpy().astype(np.float64)
        kh, kw = w.shape[:2]
        H, W = x.shape[1] - kh + 1, x.shape[2] - kw + 1
        y = np.zeros((x.shape[0], H, W, w.shape[3]))
        for i in range(kh):
            for j in range(kw):
                y += x[:, i:i + H, j:j + W, :] @ w[i, j]
        x = np.maximum(y + b, 0)
    x = x.reshape(x.shape[0], -1)
    for i, (w, b) in enumerate(torso.dense):
        x = x @ w.cpu().numpy().astype(np.float64) + b.cpu().numpy().astype(np.float64)
        if i > 0:
            x = np.maximum(x, 0)
    return x


def test_torso_forward_and_initialisation():
    torch = _torch()
    from alphazero_amd import minatar
    g = torch.Generator(device="cuda").manual_seed(4)
    nets = minatar.AtariNetworks(num_atoms=4, generator=g)
    t = nets.torso
    assert [tuple(w.shape) for w, _ in t.conv] == [(3, 3, 4, 32), (2, 2, 32, 64)]
    assert [tuple(w.shape) for w, _ in t.dense] == [(3136, 512), (512, 512)]
    # orthogonal(sqrt 2): the flattened kernel's columns (or rows) are orthogonal with norm sqrt 2
    w = t.dense[1][0]
    torch.testing.assert_close(w.T @ w, 2 * torch.eye(512, device="cuda"), rtol=1e-4, atol=1e-4)
    w = t.conv[0][0].reshape(36, 32)
    torch.testing.assert_close(w.T @ w, 2 * torch.eye(32, device="cuda"), rtol=1e-4, atol=1e-4)
    assert all(float(b.abs().sum()) == 0 for _, b in t.conv + t.dense)
    env = minatar.BreakoutEnv()
    st = env.reset(0, batch_size=32)
    for _ in range(6):
        st = env.step(st, torch.randint(0, 3, (32,), device="cuda", dtype=torch.int32))
    feats = t(st.obs)
    ref = _torso_numpy(t, st.obs.cpu().numpy())
    np.testing.assert_allclose(feats.cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    logits, value = nets.forward(st.obs)
    assert logits.shape == (32, 3) and value.shape == (32, 4)


def test_gumbel_search_around_breakout_and_the_torso(oracle=None):
    """gumbel_muzero_policy with recurrent_fn = Breakout model step + torso networks on the
    per-simulation kernels: the evaluations the search asked for are recorded and handed to the C
    checker's injected search, whose tree, action and weights must be the same; the embeddings
    stored per node are the states the NumPy game reaches along each node's action path."""
    torch = _torch()
    import alphazero_amd as mctx
    from alphazero_amd import minatar
    from oracle import c_oracle as co, minatar_numpy as bk
    B, S = 24, 20
    g = torch.Generator(device="cuda").manual_seed(5)
    nets = minatar.AtariNetworks(num_atoms=1, generator=g)
    env = minatar.BreakoutEnv()
    st = env.reset(7, batch_size=B)
    for _ in range(9):   # a few moves in, so that lanes differ
        st = env.step(st, torch.randint(0, 3, (B,), device="cuda", dtype=torch.int32, generator=g))
    inner = minatar.recurrent_fn(env, nets, gamma=0.99)
    rec = []

    def recording(params, key, action, emb):
        out, new = inner(params, key, action, emb)
        rec.append((out, action.clone()))
        return out, new
    logits, value = nets.forward(st.obs)
    root = mctx.RootFnOutput(prior_logits=logits, value=value, embedding=st.embedding)
    gum = torch.randn((B, 3), device="cuda", generator=g)
    out = mctx.gumbel_muzero_policy((), 0, root, recording, S, root_gumbel=gum, max_num_considered_actions=3)
    torch.cuda.synchronize()
    assert len(rec) == S
    evals = dict(prior_logits=np.stack([r[0].prior_logits.cpu().numpy() for r in rec], 1),
                 value=np.stack([r[0].value.cpu().numpy() for r in rec], 1),
                 reward=np.stack([r[0].reward.cpu().numpy() for r in rec], 1),
                 discount=np.stack([r[0].discount.cpu().numpy() for r in rec], 1))
    cfg = co.make_cfg(B, S, 3, 1, variant="gumbel", max_num_considered=3)
    tree = co.alloc_tree(B, S, 3, 1)
    prior, g_scaled = co.root_gumbel(logits.cpu().numpy(), None, gumbel=gum.cpu().numpy(), gumbel_scale=1.0)
    co.tree_init(cfg, tree, prior, value.cpu().numpy())
    co.search_injected(cfg, tree, evals, root_gumbel=g_scaled)
    action, weights = co.epilogue_gumbel(cfg, tree, prior, g_scaled, None)
    t = out.search_tree
    for f in ("node_visits", "parents", "action_from_parent", "children_index", "children_visits",
              "raw_values", "node_values", "children_prior_logits", "children_rewards",
              "children_discounts", "children_values"):
        got = getattr(t, f).cpu().numpy()
        assert np.array_equal(got, tree[f].reshape(got.shape)), f
    assert np.array_equal(out.action.cpu().numpy(), action)
    assert np.array_equal(out.action_weights.cpu().numpy(), weights)
    # every node's stored state = the root's state stepped along the node's action path
    emb = t.embeddings["state"].cpu().numpy()
    par, afp = tree["parents"], tree["action_from_parent"]
    for b in range(B):
        for n in range(1, S + 1):
            if par[b, n] < 0:
                continue
            _, s, _, _, _ = bk.step_env(bk.EnvState.from_words(emb[b, par[b, n]]), afp[b, n])
            assert np.array_equal(emb[b, n], s.words()), (b, n)


@pytest.mark.parametrize("policy,atoms,per_alpha,normalize", [("gumbel", 1, 0.0, True), ("sampled", 4, 0.6, False)])
def test_trainer_on_breakout_with_the_convolutional_torso(policy, atoms, per_alpha, normalize):
    """alphazero.py with env_id 'Breakout-MinAtar' (is_atari_env, :878): training steps run end to
    end -- search on the per-simulation kernels with the HIP model step and the torso networks,
    flattened observations in the replay rows, L2 and the optimiser over the torso's parameters
    too (:787-789, :1304-1308).  The loss the learner minimises is checked against a direct
    restatement with torch's functional convolutions on the same minibatch.
    (normalize_observations off in one case: a cell that never lit up during the first unrolls
    has the minimum deviation of 1e-6, running_statistics.py:75-180, and the first time it does the
    networks see 1e6 -- the reference's arithmetic, but the metrics are no longer finite.)"""
    import torch
    import torch.nn.functional as F
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from alphazero_amd import learner as L, minatar
    from alphazero_amd.trainer import Config, Trainer
    cfg = Config(env_id="Breakout-MinAtar", num_envs=32, unroll_length=5, num_simulations=8, policy=policy,
                 num_atoms=atoms, per_alpha=per_alpha, max_replay_size=1024, min_replay_size=0,
                 replay_buffer_batch_size=16, num_minibatches=4, num_samples=6, episode_length=50,
                 atari_dense_layer_sizes=(64,), use_gae=(atoms == 1), normalize_observations=normalize)
    tr = Trainer(cfg, device="cuda")
    nets = tr.policy_layers
    assert isinstance(nets, minatar.AtariNetworks) and tr.value_layers is None
    assert len(tr.optimizer.params) == len(nets.leaves()) == 2 * (2 + 2 + 1 + 1)
    assert tr.layout["observation"][1] == 400 and tr.layout["_width"] == 4 * 400 + 3 + 2 * atoms + 10
    before = [p.clone() for p in tr.optimizer.params]
    tr.trace = []
    m = tr.train(2)
    assert all(math.isfinite(v) for v in m.values()), m
    assert m["buffer_current_size"] == 2 * 5 * 32 and tr.optimizer.count == 2 * cfg.num_minibatches
    assert tr.stats.count == (2 * 5 * 32 if normalize else 0) and tr.stats.mean.shape == (400,)
    changed = [not torch.equal(a, b) for a, b in zip(before, tr.optimizer.params)]
    assert all(changed[i] for i in (0, 2, 4)), changed          # both convolutions and the first dense layer moved
    # the replay rows hold the flattened observation the search saw, and 0/1 cells only
    step0 = tr.trace[0]
    assert step0["obs"].shape == (32, 400) and set(step0["obs"].unique().tolist()) <= {0.0, 1.0}
    assert step0["action_weights"].shape == (32, 3)
    torch.testing.assert_close(step0["action_weights"].sum(-1), torch.ones(32, device="cuda"))
    # the loss on one minibatch against a direct restatement
    _, rows = tr.replay.sample(tr.buffer_state)
    data = L.unpack_rows(tr.layout, rows[:16])
    mean, std = tr.stats.mean, tr.stats.std
    total, met = L.compute_muzero_loss(nets, None, mean, std, data, atoms, l2_coef=cfg.l2_coef,
                                       vf_cost=cfg.vf_cost, qr_kappa=cfg.qr_kappa)
    x = ((data["observation"] - mean) / std).reshape(-1, 10, 10, 4).permute(0, 3, 1, 2)
    for w, b in nets.torso.conv:
        x = F.relu(F.conv2d(x, w.permute(3, 2, 0, 1), b))
    x = x.permute(0, 2, 3, 1).reshape(16, -1)
    x = x @ nets.torso.dense[0][0] + nets.torso.dense[0][1]
    x = F.relu(x @ nets.torso.dense[1][0] + nets.torso.dense[1][1])
    logits = x @ nets.policy[0][0] + nets.policy[0][1]
    value = x @ nets.value[0][0] + nets.value[0][1]
    vs = data["value_prefix_target"][:, None] + data["bootstrap_discount"][:, None] * data["bootstrap_value"]
    vl = L.quantile_regression_loss(value, vs, cfg.qr_kappa, atoms) if atoms > 1 else L.mse_value_loss(value, vs)
    want = -(data["target_policy_probs"] * F.log_softmax(logits, -1)).sum(-1).mean() \
        + cfg.vf_cost * (vl * data["weight"]).mean() \
        + cfg.l2_coef * 0.5 * sum((p.double() ** 2).sum() for p in nets.leaves()).float()
    torch.testing.assert_close(total, want, rtol=1e-5, atol=1e-6)
    ev = tr.evaluate(num_eval_envs=8, episode_length=20)
    assert ev["eval/avg_episode_length"] > 0 and math.isfinite(ev["eval/episode_reward"])
