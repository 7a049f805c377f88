"""CPU restatement of the reference's learner-side formulas -- TEST INFRASTRUCTURE ONLY.

Follows alphazero.py: n_step_bootstrapped_targets (:544-602), compute_gae (:605-663),
quantile_regression_loss (:667-706), mse_value_loss (:709-720), compute_muzero_loss
(:723-807), the priorities (:1134-1135, 1177-1179), the optimiser chain
clip_by_global_norm + adam with a linear schedule (:968-979, optax's published update
rules) and running_statistics.update (running_statistics.py:75-180).  Everything is plain
NumPy in the dtype of its inputs (tests feed float64 to get a reference that is more
accurate than the float32 GPU path).

PARITY UNPINNED against the reference itself: it has no tests or fixtures for these
functions and JAX / optax are not installed here.
"""
import numpy as np


def n_step_bootstrapped_targets(rewards, discounts, termination_discount, observations, values,
                                n=5, gamma=1.0):
    values = values[1:]
    T = rewards.shape[0]
    pad = min(n - 1, T)
    boot_obs = np.concatenate([observations[n - 1:], np.stack([observations[-1]] * pad)]) if pad else observations[n - 1:]
    boot_val = np.concatenate([values[n - 1:], np.stack([values[-1]] * pad)]) if pad else values[n - 1:]
    tail = (n - 1,) + rewards.shape[1:]
    rewards_p = np.concatenate([rewards, np.zeros(tail, rewards.dtype)])
    discounts_p = np.concatenate([discounts, np.ones(tail, rewards.dtype)]) * gamma
    prefix = rewards_p[n - 1:n - 1 + T]
    boot_disc = (np.concatenate([termination_discount, np.ones(tail, rewards.dtype)]) * gamma)[n - 1:n - 1 + T]
    for i in range(n - 2, -1, -1):
        r_ = rewards_p[i:i + T]
        d_ = discounts_p[i:i + T]
        prefix = r_ + d_ * prefix
        boot_disc = boot_disc * d_
    return prefix, boot_obs, boot_val, boot_disc


def compute_gae(rewards, discounts, termination_discount, observations, values, lambda_=1.0,
                discount=0.99):
    termination = (1 - termination_discount)[..., None]
    truncation = (1 - discounts)[..., None] * (1 - termination)
    mask = 1 - truncation
    deltas = rewards[..., None] + discount * (1 - termination) * values[1:] - values[:-1]
    deltas = deltas * mask
    acc = np.zeros_like(values[0])
    out = np.zeros_like(values[:-1])
    for t in range(rewards.shape[0] - 1, -1, -1):
        acc = deltas[t] + discount * (1 - termination[t]) * mask[t] * lambda_ * acc
        out[t] = acc
    vs = out + values[:-1]
    return np.zeros_like(rewards), np.zeros_like(observations), vs, np.ones_like(discounts)


def quantile_regression_loss(values, targets, kappa, num_atoms):
    err = targets[..., None, :] - values[..., :, None]
    a = np.abs(err)
    huber = (a <= kappa) * 0.5 * err ** 2 + (a > kappa) * kappa * (a - 0.5 * kappa)
    tau = ((np.arange(num_atoms, dtype=values.dtype) + 0.5) / num_atoms)[None, :, None]
    w = np.abs(tau - (err < 0))
    return np.sum(np.mean(w * huber, 2), 1)


def mse_value_loss(values, targets):
    e = targets - values
    return np.squeeze(e * e * 0.5)


def swish(x):
    return x / (1 + np.exp(-x))


def mlp(layers, x):
    for i, (w, b) in enumerate(layers):
        x = x @ w + b
        if i + 1 < len(layers):
            x = swish(x)
    return x


def log_softmax(x):
    m = x.max(-1, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(-1, keepdims=True))


def muzero_loss(policy_layers, value_layers, mean, std, batch, num_atoms, kappa=1.0, vf_cost=0.5,
                l2_coef=1e-4, per_importance_sampling=True):
    """alphazero.py:723-807 on a dict batch; returns (total, metrics)."""
    x = (batch["observation"] - mean) / std
    logits = mlp(policy_layers, x)
    baseline = mlp(value_layers, x)
    lsm = log_softmax(logits)
    policy_loss = -np.mean(np.sum(batch["target_policy_probs"] * lsm, -1))
    vs = batch["value_prefix_target"][:, None] + batch["bootstrap_discount"][:, None] * batch["bootstrap_value"]
    if num_atoms > 1:
        vl = quantile_regression_loss(baseline, vs, kappa, num_atoms)
    else:
        vl = mse_value_loss(baseline, vs)
    if per_importance_sampling:
        vl = vl * batch["weight"]
    v_loss = vf_cost * np.mean(vl)
    l2 = l2_coef * 0.5 * sum(np.sum(w ** 2) + np.sum(b ** 2) for w, b in policy_layers + value_layers)
    a = batch["raw_action"].astype(np.int64)
    logp = np.take_along_axis(lsm, a[:, None], 1)[:, 0]
    log_ratio = logp - batch["prior_log_prob"]
    approx_kl = np.mean((np.exp(log_ratio) - 1) - log_ratio)
    p = np.exp(lsm)
    entropy = np.mean(-np.sum(p * lsm, -1))
    total = policy_loss + v_loss + l2
    return total, dict(total_loss=total, policy_loss=policy_loss, value_loss=v_loss, l2_penalty=l2,
                       entropy=entropy, approx_kl=approx_kl)


def priorities(targets, values, per_alpha):
    """alphazero.py:1135 / :1179."""
    return (np.mean(np.abs(targets - values), -1) + 1e-10) ** per_alpha


def linear_schedule(init, end, steps, count):
    frac = 1 - min(count, steps) / steps
    return (init - end) * frac + end


def clip_adam_step(params, grads, m, v, count, lr, max_norm, b1=0.9, b2=0.999, eps=1e-8):
    """optax.chain(clip_by_global_norm(max_norm), adam(lr)) on lists of arrays; `count` is
    the number of updates done before this one.  Returns (params, m, v)."""
    gnorm = np.sqrt(sum(np.sum(g ** 2) for g in grads))
    scale = max_norm / gnorm if gnorm >= max_norm else 1.0   # optax: where(g_norm < max, g, g * max / g_norm)
    grads = [g * scale for g in grads]
    t = count + 1
    m = [b1 * mi + (1 - b1) * g for mi, g in zip(m, grads)]
    v = [b2 * vi + (1 - b2) * g * g for vi, g in zip(v, grads)]
    mh = [mi / (1 - b1 ** t) for mi in m]
    vh = [vi / (1 - b2 ** t) for vi in v]
    params = [p - lr * a / (np.sqrt(b) + eps) for p, a, b in zip(params, mh, vh)]
    return params, m, v


def running_statistics_update(count, mean, summed_variance, batches, std_min=1e-6, std_max=1e6):
    """running_statistics.py:75-180 with one batch per rank in `batches` (psum over ranks)."""
    step = sum(int(np.prod(b.shape[:-1])) for b in batches)
    count = count + step
    mean_update = sum(np.sum(b - mean, axis=tuple(range(b.ndim - 1))) / count for b in batches)
    new_mean = mean + mean_update
    var_update = sum(np.sum((b - mean) * (b - new_mean), axis=tuple(range(b.ndim - 1))) for b in batches)
    summed_variance = summed_variance + var_update
    std = np.clip(np.sqrt(np.maximum(summed_variance, 0) / count), std_min, std_max)
    return count, new_mean, summed_variance, std
