"""Hand-checkable cases of the training-step assembly restatement (oracle/trainer_numpy.py)."""
import numpy as np

from oracle import trainer_numpy as tn


def _step(obs, action, reward, done, trunc, sv, value, nxt):
    f = np.float64
    return dict(obs=np.array([obs], f), real_obs=np.array([obs], f), value=np.array([[value]], f),
                action=np.array([action], np.int32), action_weights=np.array([[0.25, 0.75]], f),
                search_value=np.array([[sv]], f), prior_log_prob=np.array([-0.5], f),
                reward=np.array([reward], f), done=np.array([done], f), truncation=np.array([trunc], f),
                next_real_obs=np.array([nxt], f))


def test_row_order_and_one_step_targets():
    """Two steps, one lane, n = 1: target_t = r_t + gamma * (1 - done_t) * v_{t+1} (search values),
    priority = (|target - prior value| + 1e-10) ** alpha, columns in the ravel order of
    envs/transition.py:18-34 with priority and weight last."""
    s0 = _step([1.0, 2.0], 1, 1.0, 0.0, 0.0, 0.5, 0.4, [3.0, 4.0])
    s1 = _step([3.0, 4.0], 0, 2.0, 1.0, 0.0, 0.7, 0.6, [5.0, 6.0])
    boot = _step([5.0, 6.0], 0, 0.0, 0.0, 0.0, 0.9, 0.0, [0.0, 0.0])
    flat = tn.assemble([s0, s1], boot, use_gae=False, gae_lambda=0.95, n_step_n=1, gamma=0.9, per_alpha=0.5)
    # n = 1: prefix = reward, bootstrap value = next search value, bootstrap discount = gamma * (1 - done)
    np.testing.assert_allclose(flat["value_prefix_target"], [1.0, 2.0])
    np.testing.assert_allclose(flat["bootstrap_value"][:, 0], [0.7, 0.9])
    np.testing.assert_allclose(flat["bootstrap_discount"], [0.9, 0.0])
    targets = np.array([1.0 + 0.9 * 0.7, 2.0])
    np.testing.assert_allclose(flat["priority"], (np.abs(targets - np.array([0.4, 0.6])) + 1e-10) ** 0.5)
    rows = tn.rows(flat)
    assert rows.shape == (2, 4 * 2 + 2 + 2 * 1 + 10)          # 4 obs + A + 2K + 10 (K = 1, A = 2)
    r = rows[0]
    assert list(r[:4]) == [1.0, 2.0, 1.0, 2.0] and r[4] == 1.0 and r[5] == 1.0 and r[6] == 1.0
    assert list(r[7:9]) == [3.0, 4.0] and list(r[9:11]) == [0.25, 0.75] and r[11] == 0.5
    assert r[12] == 1.0 and list(r[13:15]) == [3.0, 4.0] and r[15] == 0.7 and r[16] == 0.9
    assert r[17] == -0.5 and r[18] == 1.0 and r[19] == 0.0 and r[21] == 1.0
    np.testing.assert_allclose(r[20], flat["priority"][0])


def test_truncation_cuts_the_bootstrap_like_termination():
    """discounts = (1 - done) * (1 - truncation): a truncated step stops the n-step accumulation."""
    s0 = _step([0.0, 0.0], 0, 1.0, 1.0, 1.0, 0.5, 0.1, [1.0, 1.0])     # EpisodeWrapper: done and truncated
    s1 = _step([1.0, 1.0], 0, 1.0, 0.0, 0.0, 0.5, 0.1, [2.0, 2.0])
    boot = _step([2.0, 2.0], 0, 0.0, 0.0, 0.0, 0.5, 0.0, [0.0, 0.0])
    flat = tn.assemble([s0, s1], boot, use_gae=False, gae_lambda=0.95, n_step_n=2, gamma=0.5, per_alpha=1.0)
    assert flat["value_prefix_target"][0] == 1.0 and flat["bootstrap_discount"][0] == 0.0
    assert flat["value_prefix_target"][1] == 1.0       # last row of the unroll: padded with zero rewards


def test_refreshed_priorities_use_the_given_networks_and_statistics():
    w = np.array([[2.0], [0.0]])
    layers = [(w, np.array([1.0]))]
    batch = dict(observation=np.array([[3.0, 7.0]]), value_prefix_target=np.array([1.0]),
                 bootstrap_discount=np.array([0.5]), bootstrap_value=np.array([[4.0]]))
    p = tn.refreshed_priorities(batch, layers, mean=np.array([1.0, 0.0]), std=np.array([2.0, 1.0]), per_alpha=1.0)
    # value = 2 * (3 - 1) / 2 + 1 = 3; target = 1 + 0.5 * 4 = 3
    np.testing.assert_allclose(p, [1e-10])


def test_eval_metrics_count_the_first_episode_only():
    """Two lanes over four steps: lane 0 ends its first episode at step 2 (what follows belongs to
    the auto-reset episode and is not counted), lane 1 never ends."""
    f = np.float64
    steps = [dict(reward=np.array([1.0, 2.0], f), done=np.array([0.0, 0.0], f), steps=np.array([1.0, 1.0], f)),
             dict(reward=np.array([1.0, 2.0], f), done=np.array([1.0, 0.0], f), steps=np.array([2.0, 2.0], f)),
             dict(reward=np.array([5.0, 2.0], f), done=np.array([0.0, 0.0], f), steps=np.array([1.0, 3.0], f)),
             dict(reward=np.array([5.0, 2.0], f), done=np.array([1.0, 0.0], f), steps=np.array([2.0, 4.0], f))]
    m = tn.eval_metrics(steps)
    assert list(m["episode_reward"]) == [2.0, 8.0] and list(m["episode_steps"]) == [2.0, 4.0]
    assert list(m["active_episodes"]) == [0.0, 1.0]
    assert m["eval/episode_reward"] == 5.0 and m["eval/episode_reward_std"] == 3.0
    assert m["eval/avg_episode_length"] == 3.0
