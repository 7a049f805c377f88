"""CPU restatement of how the reference's training step ASSEMBLES its data -- TEST INFRASTRUCTURE
ONLY (the product never imports this).

Follows alphazero.py: the transition record of actor_step (:385-403), the stacking over the
unroll and the extra bootstrap search (:1104-1118), targets and priorities (:1122-1138), the
flattening to replay rows (:1143-1151, ravel order of envs/transition.py:18-34 with the extras
dictionaries in key order) and the priority refresh after the SGD step with the pre-update
networks (:1177-1180).  The search itself is not restated here: its outputs (actions, action
weights, search values) and the environment's transitions are INJECTED, as recorded from a run.
The formulas it calls are the restatements of learner_numpy.py.

PARITY UNPINNED against the reference itself: it has no tests or fixtures for this path and JAX
is not installed here.
"""
import numpy as np

from . import learner_numpy as ln

# ravel order of MCTSTransition (envs/transition.py:18-34); extras = {'policy_extras':
# {'prior_log_prob', 'raw_action'}, 'state_extras': {'truncation'}} flatten in key order
FIELDS = ("observation", "real_obs", "action", "reward", "discount", "next_observation",
          "target_policy_probs", "search_value", "value_prefix_target", "bootstrap_observation",
          "bootstrap_value", "bootstrap_discount", "prior_log_prob", "raw_action", "truncation",
          "priority", "weight")


def transition(step):
    """alphazero.py:385-403 from one recorded actor step: `step` holds the environment state
    before the step (obs, real_obs), the prior value, the search outputs (action, action_weights,
    search_value, prior_log_prob) and the stepped environment (reward, done, truncation,
    next_real_obs)."""
    reward, done = step["reward"], step["done"]
    return dict(
        observation=step["obs"], real_obs=step["real_obs"], action=step["action"].astype(reward.dtype),
        reward=reward, discount=1 - done, next_observation=step["next_real_obs"],
        target_policy_probs=step["action_weights"], search_value=step["search_value"],
        value_prefix_target=np.zeros_like(reward), bootstrap_observation=np.zeros_like(step["obs"]),
        bootstrap_value=step["value"],            # overloaded with the prior value (:396)
        bootstrap_discount=np.ones_like(1 - done), prior_log_prob=step["prior_log_prob"],
        raw_action=step["action"].astype(reward.dtype), truncation=step["truncation"],
        priority=np.ones_like(done), weight=np.ones_like(done))


def assemble(steps, bootstrap_step, *, use_gae, gae_lambda, n_step_n, gamma, per_alpha):
    """alphazero.py:1104-1151: `steps` are the unroll's recorded actor steps (in order),
    `bootstrap_step` the extra one whose search value closes the value sequence.  Returns the
    flat [T * B, ...] fields that go into the replay queue."""
    trs = [transition(s) for s in steps]
    data = {k: np.stack([t[k] for t in trs], 0) for k in trs[0]}
    values = np.concatenate([data["search_value"], bootstrap_step["search_value"][None]], 0)
    fn = ln.compute_gae if use_gae else ln.n_step_bootstrapped_targets
    kw = dict(lambda_=gae_lambda, discount=gamma) if use_gae else dict(n=n_step_n, gamma=gamma)
    prefix, b_obs, b_val, b_disc = fn(
        rewards=data["reward"], discounts=data["discount"] * (1 - data["truncation"]),
        termination_discount=data["discount"], observations=data["next_observation"],
        values=values, **kw)
    targets = prefix[..., None] + b_disc[..., None] * b_val
    data["priority"] = ln.priorities(targets, data["bootstrap_value"], per_alpha)
    data.update(value_prefix_target=prefix, bootstrap_observation=b_obs, bootstrap_value=b_val,
                bootstrap_discount=b_disc)
    return {k: v.reshape((-1,) + v.shape[2:]) for k, v in data.items()}


def rows(flat):
    """The flat float rows [T * B, D] of the replay queue."""
    cols = []
    for name in FIELDS:
        x = np.asarray(flat[name])
        cols.append(x.reshape(x.shape[0], -1))
    return np.concatenate(cols, 1)


def refreshed_priorities(batch, value_layers, mean, std, per_alpha):
    """alphazero.py:1177-1180: priorities of the sampled rows from the value network as it was
    BEFORE the SGD step, on observations normalised with the statistics of before the step."""
    targets = batch["value_prefix_target"][..., None] + batch["bootstrap_discount"][..., None] * \
        batch["bootstrap_value"]
    values = ln.mlp(value_layers, (batch["observation"] - mean) / std)
    return ln.priorities(targets, values, per_alpha)


def eval_metrics(steps):
    """alphazero.py:476-506 over an evaluation unroll whose environment is wrapped in EvalWrapper
    (envs/brax_wrappers.py:153-197): `steps` are the recorded stepped states in order, each with
    reward [B], done [B] and steps [B] (info['steps'] of EpisodeWrapper, :81-97).  A lane is
    active until its first done; while active its rewards are added up and its step counter
    is copied."""
    n = steps[0]["reward"].shape[0]
    active = np.ones(n)                   # reset (:156-167)
    episode_reward = np.zeros(n)
    episode_steps = np.zeros(n)
    for s in steps:                       # step (:169-197)
        episode_steps = np.where(active != 0, s["steps"], episode_steps)
        episode_reward = episode_reward + s["reward"] * active
        active = active * (1 - s["done"])
    return {"episode_reward": episode_reward, "episode_steps": episode_steps, "active_episodes": active,
            "eval/episode_reward": float(np.mean(episode_reward)),
            "eval/episode_reward_std": float(np.std(episode_reward)),
            "eval/avg_episode_length": float(np.mean(episode_steps))}
