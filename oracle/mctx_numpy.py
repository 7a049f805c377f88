"""NumPy restatement of the reference's batched MCTS (TEST INFRASTRUCTURE ONLY).

A second, independent CPU checker: one lane at a time, plain Python loops,
``np.float32`` arithmetic with NumPy's own exp / log / sin / cos.  It follows
the reference source line by line (paths relative to the reference checkout are
cited per function) and exists to cross-check the plain-C checker
(``mcts_oracle.c``), whose elementary functions come from the build's numerics
specification instead of libm.  Discrete results of the two agree except on
floating-point near-ties, which the tests detect and report.

Only ``tests/`` imports this module.  PARITY UNPINNED: the reference has no
tests or golden vectors and its JAX runtime is not installable here; the pins
are the hand-derived known-answer cases in ``tests/test_oracle_kats.py``.
"""
import math

import numpy as np

f32 = np.float32
ROOT_INDEX, NO_PARENT, UNVISITED = 0, -1, -1
TINY = np.finfo(np.float32).tiny
FMIN = np.finfo(np.float32).min


# ---------------------------------------------------------------- seq_halving
def get_sequence_of_considered_visits(m, num_simulations):
    """mctx_dist/_src/seq_halving.py:36-70"""
    if m <= 1:
        return tuple(range(num_simulations))
    log2max = int(math.ceil(math.log2(m)))
    sequence, visits, num_considered = [], [0] * m, m
    while len(sequence) < num_simulations:
        num_extra = max(1, int(num_simulations / (log2max * num_considered)))
        for _ in range(num_extra):
            sequence.extend(visits[:num_considered])
            for i in range(num_considered):
                visits[i] += 1
        num_considered = max(2, num_considered // 2)
    return tuple(sequence[:num_simulations])


def get_table_of_considered_visits(max_m, num_simulations):
    """mctx_dist/_src/seq_halving.py:73-88"""
    return tuple(get_sequence_of_considered_visits(m, num_simulations)
                 for m in range(max_m + 1))


def score_considered(considered_visit, gumbel, logits, normalized_qvalues, visit_counts):
    """mctx_dist/_src/seq_halving.py:23-33"""
    logits = logits - np.max(logits)
    penalty = np.where(visit_counts == considered_visit, f32(0), f32(-np.inf))
    with np.errstate(invalid="ignore"):
        return np.maximum(f32(-1e9), gumbel + logits + normalized_qvalues) + penalty


# ---------------------------------------------------------------- helpers
def softmax(x):
    e = np.exp(x - np.max(x))
    s = f32(0)
    for v in e:
        s = f32(s + v)
    return (e / s).astype(f32)


def mean_atoms(x):
    """jnp.mean(x, axis=-1) with a left-to-right sum."""
    s = np.zeros(x.shape[:-1], f32)
    for k in range(x.shape[-1]):
        s = (s + x[..., k]).astype(f32)
    return (s / f32(x.shape[-1])).astype(f32)


def masked_argmax(to_argmax, invalid):
    """mctx_dist/_src/action_selection.py:243-253"""
    if invalid is not None:
        to_argmax = np.where(invalid != 0, f32(-np.inf), to_argmax)
    return int(np.argmax(to_argmax))


class LaneTree:
    """One lane of mctx_dist/_src/tree.py:29-91 (arrays without the batch axis)."""

    def __init__(self, S, A, K):
        N = S + 1
        self.S, self.A, self.K = S, A, K
        self.node_visits = np.zeros(N, np.int32)
        self.raw_values = np.zeros((N, K), f32)
        self.node_values = np.zeros((N, K), f32)
        self.parents = np.full(N, NO_PARENT, np.int32)
        self.action_from_parent = np.full(N, NO_PARENT, np.int32)
        self.children_index = np.full((N, A), UNVISITED, np.int32)
        self.children_prior_logits = np.zeros((N, A), f32)
        self.children_visits = np.zeros((N, A), np.int32)
        self.children_rewards = np.zeros((N, A), f32)
        self.children_discounts = np.zeros((N, A), f32)
        self.children_values = np.zeros((N, A, K), f32)
        self.embeddings = [None] * N

    def qvalues(self, node):
        """tree.py:142-147"""
        return (self.children_rewards[node][:, None]
                + self.children_discounts[node][:, None] * self.children_values[node]).astype(f32)


# ---------------------------------------------------------------- qtransforms
def qtransform_by_min_max(tree, node, *, min_value, max_value):
    """mctx_dist/_src/qtransforms.py:24-51"""
    q = mean_atoms(tree.qvalues(node))
    vis = tree.children_visits[node]
    s = np.where(vis > 0, q, f32(min_value)).astype(f32)
    return ((s - f32(min_value)) / (f32(max_value) - f32(min_value))).astype(f32)


def qtransform_by_parent_and_siblings(tree, node, *, epsilon=1e-8):
    """mctx_dist/_src/qtransforms.py:54-87"""
    q = mean_atoms(tree.qvalues(node))
    vis = tree.children_visits[node]
    v = mean_atoms(tree.node_values[node])
    safe = np.where(vis > 0, q, v).astype(f32)
    lo = np.minimum(v, np.min(safe))
    hi = np.maximum(v, np.max(safe))
    completed = np.where(vis > 0, q, lo).astype(f32)
    return ((completed - lo) / np.maximum(f32(hi - lo), f32(epsilon))).astype(f32)


def qtransform_completed_by_mix_value(tree, node, *, value_scale=0.1, maxvisit_init=50.0,
                                      rescale_values=True, use_mixed_value=True, epsilon=1e-8):
    """mctx_dist/_src/qtransforms.py:90-149, 152-198"""
    q = mean_atoms(tree.qvalues(node))
    vis = tree.children_visits[node]
    raw = mean_atoms(tree.raw_values[node])
    probs = softmax(tree.children_prior_logits[node])
    if use_mixed_value:
        sum_visits = int(np.sum(vis))
        probs = np.maximum(TINY, probs)
        sum_probs = f32(0)
        for a in range(len(vis)):
            sum_probs = f32(sum_probs + (probs[a] if vis[a] > 0 else f32(0)))
        weighted_q = f32(0)
        for a in range(len(vis)):
            term = f32(probs[a] * q[a]) / (sum_probs if vis[a] > 0 else f32(1)) if vis[a] > 0 else f32(0)
            weighted_q = f32(weighted_q + f32(term))
        value = f32(f32(raw + f32(f32(sum_visits) * weighted_q)) / f32(sum_visits + 1))
    else:
        value = raw
    cq = np.where(vis > 0, q, value).astype(f32)
    if rescale_values:
        mn, mx = np.min(cq), np.max(cq)
        cq = ((cq - mn) / np.maximum(f32(mx - mn), f32(epsilon))).astype(f32)
    visit_scale = f32(f32(maxvisit_init) + f32(np.max(vis)))
    return (f32(visit_scale * f32(value_scale)) * cq).astype(f32)


# ---------------------------------------------------------------- action selection
def puct_action_selection(noise, tree, node, depth, *, invalid, sampled, pb_c_init=1.25,
                          pb_c_base=19652.0, qtransform=qtransform_by_parent_and_siblings):
    """action_selection.py:48-90 (sampled=False) and :93-135 (sampled=True).
    ``noise`` is the uniform(key, (A,)) draw."""
    vis = tree.children_visits[node]
    n = tree.node_visits[node]
    pb_c = f32(f32(pb_c_init) + np.log(f32(f32(f32(n) + f32(pb_c_base)) + f32(1)) / f32(pb_c_base)))
    if sampled:
        probs = tree.children_prior_logits[node]
    else:
        probs = softmax(tree.children_prior_logits[node])
    policy_score = (f32(np.sqrt(f32(n)) * pb_c) * probs / (vis + 1).astype(f32)).astype(f32)
    value_score = qtransform(tree, node)
    node_noise_score = (f32(1e-7) * noise).astype(f32)
    to_argmax = ((value_score + policy_score).astype(f32) + node_noise_score).astype(f32)
    return masked_argmax(to_argmax, invalid * f32(depth == 0))


def gumbel_root_action_selection(tree, node, *, invalid, gumbel, num_simulations,
                                 max_num_considered_actions, qtransform):
    """action_selection.py:148-200"""
    vis = tree.children_visits[node]
    logits = tree.children_prior_logits[node]
    cq = qtransform(tree, node)
    table = np.array(get_table_of_considered_visits(max_num_considered_actions, num_simulations))
    num_valid = int(np.sum(1 - invalid))
    num_considered = min(max_num_considered_actions, num_valid)
    sim_index = min(int(np.sum(vis)), num_simulations - 1)
    considered_visit = table[num_considered, sim_index]
    return masked_argmax(score_considered(considered_visit, gumbel, logits, cq, vis), invalid)


def gumbel_interior_action_selection(tree, node, *, qtransform):
    """action_selection.py:203-240, 256-277"""
    vis = tree.children_visits[node]
    logits = tree.children_prior_logits[node]
    cq = qtransform(tree, node)
    probs = softmax((logits + cq).astype(f32))
    to_argmax = (probs - vis.astype(f32) / f32(1 + int(np.sum(vis)))).astype(f32)
    return int(np.argmax(to_argmax))


class MaskedLaneTree:
    """mctx_dist/_src/policies.py:642-673 (_mask_tree) on one lane: columns [lo, hi) of every
    per-action array, everything else shared."""

    def __init__(self, t, lo, hi):
        self.node_visits, self.raw_values, self.node_values = t.node_visits, t.raw_values, t.node_values
        self.children_index = t.children_index[:, lo:hi]
        self.children_prior_logits = t.children_prior_logits[:, lo:hi]
        self.children_visits = t.children_visits[:, lo:hi]
        self.children_rewards = t.children_rewards[:, lo:hi]
        self.children_discounts = t.children_discounts[:, lo:hi]
        self.children_values = t.children_values[:, lo:hi]

    qvalues = LaneTree.qvalues


def stochastic_action_selection(noise, tree, node, depth, *, invalid, num_actions, **puct_kwargs):
    """mctx_dist/_src/policies.py:676-708.  The node's embedding is (state, afterstate,
    is_decision_node); ``noise`` is the uniform(key, (num_actions,)) draw of the decision rule."""
    if tree.embeddings[node][2]:
        return puct_action_selection(noise, MaskedLaneTree(tree, 0, num_actions), node, depth,
                                     invalid=invalid, sampled=False, **puct_kwargs)
    chance = MaskedLaneTree(tree, num_actions, tree.A)
    prob_chance = softmax(chance.children_prior_logits[node])
    return int(np.argmax((prob_chance / (chance.children_visits[node] + 1).astype(f32)).astype(f32))) + num_actions


def make_stochastic_recurrent_fn(decision_fn, chance_fn, num_actions, num_chance_outcomes):
    """mctx_dist/_src/policies.py:569-639 for one lane: ``decision_fn(sim, action, state)`` returns
    (chance_logits[C], afterstate_value[K], afterstate), ``chance_fn(sim, outcome, afterstate)``
    (action_logits[A], value[K], reward, discount, state); both are evaluated and the parent's
    kind picks."""
    def recurrent_fn(sim, action_or_chance, emb):
        state, afterstate, is_decision = emb
        chance_logits, afterstate_value, new_afterstate = decision_fn(sim, action_or_chance, state)
        action_logits, value, reward, discount, new_state = chance_fn(
            sim, action_or_chance - num_actions, afterstate)
        neg_a = np.full(num_actions, -np.inf, f32)
        neg_c = np.full(num_chance_outcomes, -np.inf, f32)
        new_emb = (new_state, new_afterstate, not is_decision)
        if is_decision:
            return (f32(0), f32(1), np.concatenate([neg_a, chance_logits]).astype(f32),
                    afterstate_value, new_emb)
        return reward, discount, np.concatenate([action_logits, neg_c]).astype(f32), value, new_emb
    return recurrent_fn


# ---------------------------------------------------------------- search
def search_lane(*, root_logits, root_value, root_embedding, recurrent_fn, num_simulations,
                select, max_depth=None):
    """mctx_dist/_src/search.py:31-114 for one lane.

    ``select(tree, node, depth, sim)`` returns the action; ``recurrent_fn(sim, action,
    embedding)`` returns ``(reward, discount, prior_logits[A], value[K], embedding)``.
    Returns the LaneTree and the sum over simulations of the expanded node's depth."""
    S = num_simulations
    A, K = len(root_logits), len(root_value)
    if max_depth is None:
        max_depth = S
    t = LaneTree(S, A, K)
    # instantiate_tree_from_root, search.py:345-387
    t.children_prior_logits[0] = root_logits
    t.raw_values[0] = root_value
    t.node_values[0] = root_value
    t.node_visits[0] += 1
    t.embeddings[0] = root_embedding
    sum_depth = 0
    for sim in range(S):
        # simulate, search.py:127-187
        nxt, depth = ROOT_INDEX, 0
        while True:
            node = nxt
            action = select(t, node, depth, sim)
            nxt = int(t.children_index[node, action])
            depth += 1
            if not (nxt != UNVISITED and depth < max_depth):
                break
        parent = node
        new = sim + 1 if nxt == UNVISITED else nxt  # search.py:97-99
        sum_depth += depth
        # expand, search.py:190-244, 304-342
        reward, discount, logits, value, emb = recurrent_fn(sim, action, t.embeddings[parent])
        t.children_prior_logits[new] = logits
        t.raw_values[new] = value
        t.node_values[new] = value
        t.node_visits[new] += 1
        t.embeddings[new] = emb
        t.children_index[parent, action] = new
        t.children_rewards[parent, action] = reward
        t.children_discounts[parent, action] = discount
        t.parents[new] = parent
        t.action_from_parent[new] = action
        # backward, search.py:247-292
        index = new
        leaf_value = t.node_values[new].copy()
        while index != ROOT_INDEX:
            p = int(t.parents[index])
            count = int(t.node_visits[p])
            a = int(t.action_from_parent[index])
            leaf_value = (t.children_rewards[p, a] + t.children_discounts[p, a] * leaf_value).astype(f32)
            parent_value = ((t.node_values[p] * f32(count) + leaf_value).astype(f32)
                            / f32(f32(count) + f32(1))).astype(f32)
            children_values = t.node_values[index].copy()
            t.node_values[p] = parent_value
            t.node_visits[p] = count + 1
            t.children_values[p, a] = children_values
            t.children_visits[p, a] += 1
            index = p
    return t, sum_depth


# ---------------------------------------------------------------- policies
def mask_invalid_actions(logits, invalid):
    """policies.py:486-496"""
    if invalid is None:
        return logits
    logits = (logits - np.max(logits)).astype(f32)
    return np.where(invalid != 0, FMIN, logits).astype(f32)


def apply_temperature(logits, temperature):
    """policies.py:519-524"""
    logits = (logits - np.max(logits)).astype(f32)
    return (logits / np.maximum(TINY, f32(temperature))).astype(f32)


def root_logits_muzero(prior_logits, dirichlet, fraction, invalid):
    """policies.py:80-90, 499-516"""
    probs = softmax(prior_logits)
    noisy = (f32(1 - fraction) * probs + f32(fraction) * dirichlet).astype(f32)
    return mask_invalid_actions(np.log(np.maximum(noisy, TINY)).astype(f32), invalid)


def visit_epilogue(tree, temperature, gumbel):
    """policies.py:113-122 with tree.py:102-120"""
    counts = tree.children_visits[ROOT_INDEX].astype(f32)
    total = f32(np.sum(counts))
    probs = (counts / np.maximum(total, f32(1))).astype(f32)
    if not total > 0:
        probs = np.full_like(probs, f32(1 / len(counts)))
    logits = apply_temperature(np.log(np.maximum(probs, TINY)).astype(f32), temperature)
    return int(np.argmax((gumbel + logits).astype(f32))), probs


def gumbel_epilogue(tree, root_logits, gumbel, invalid, qtransform):
    """policies.py:321-343"""
    vis = tree.children_visits[ROOT_INDEX]
    cq = qtransform(tree, ROOT_INDEX)
    to_argmax = score_considered(np.max(vis), gumbel, root_logits, cq, vis)
    action = masked_argmax(to_argmax, invalid)
    weights = softmax(mask_invalid_actions((root_logits + cq).astype(f32), invalid))
    return action, weights


# ---------------------------------------------------------------- environments
def cartpole_is_terminal(s, time):
    """cartpole.py:114-135"""
    done1 = (s[0] < f32(-2.4)) or (s[0] > f32(2.4))
    th = f32(12 * 2 * np.pi / 360)
    done2 = (s[2] < -th) or (s[2] > th)
    terminated = f32(done1 or done2)
    truncated = f32(1) - terminated if time >= 50000 else f32(0)
    return terminated, truncated


def cartpole_step(s, time, action):
    """cartpole.py:49-90; returns (state, time, obs, reward, terminated, truncated)."""
    prev_terminal, prev_truncated = cartpole_is_terminal(s, time)
    x, x_dot, theta, theta_dot = (f32(v) for v in s)
    force = f32(f32(10.0) * f32(action) - f32(10.0) * f32(1 - action))
    costheta, sintheta = np.cos(theta), np.sin(theta)
    temp = f32(f32(force + f32(f32(f32(0.05) * f32(theta_dot * theta_dot)) * sintheta)) / f32(1.1))
    thetaacc = f32(f32(f32(f32(9.8) * sintheta) - f32(costheta * temp))
                   / f32(f32(0.5) * f32(f32(4.0 / 3.0) - f32(f32(f32(0.1) * f32(costheta * costheta)) / f32(1.1)))))
    xacc = f32(temp - f32(f32(f32(f32(0.05) * thetaacc) * costheta) / f32(1.1)))
    tau = f32(0.02)
    ns = np.array([x + tau * x_dot, x_dot + tau * xacc, theta + tau * theta_dot,
                   theta_dot + tau * thetaacc], f32)
    reward = f32(1.0) - max(prev_terminal, prev_truncated)
    terminated, truncated = cartpole_is_terminal(ns, time + 1)
    return ns, time + 1, ns.copy(), reward, terminated, truncated


def mountaincar_step(s, time, action):
    """mountain_car.py:43-72, 86-94"""
    position, velocity = f32(s[0]), f32(s[1])
    velocity = f32(f32(velocity + f32(f32(action - 1) * f32(0.001)))
                   - f32(np.cos(f32(f32(3) * position)) * f32(0.0025)))
    velocity = np.clip(velocity, f32(-0.07), f32(0.07))
    position = np.clip(f32(position + velocity), f32(-1.2), f32(0.6))
    velocity = f32(velocity * f32(1 - int((position == f32(-1.2)) and (velocity < 0))))
    done = f32((position >= f32(0.5)) and (velocity >= f32(0.0)))
    truncated = f32(time + 1 >= 5000)
    ns = np.array([position, velocity, 0, 0], f32)
    return ns, time + 1, ns[:2].copy(), f32(-1.0), done, truncated


def _acrobot_dsdt(s):
    """acrobot.py:202-233"""
    m1 = m2 = l1 = I1 = I2 = f32(1.0)
    lc1 = lc2 = f32(0.5)
    g = f32(9.8)
    a = s[4]
    theta1, theta2, dtheta1, dtheta2 = s[0], s[1], s[2], s[3]
    d1 = f32(f32(f32(m1 * lc1 ** 2) + f32(m2 * f32(f32(l1 ** 2 + lc2 ** 2) + f32(2 * l1 * lc2 * np.cos(theta2))))) + I1) + I2
    d1 = f32(d1)
    d2 = f32(f32(m2 * f32(f32(lc2 ** 2) + f32(l1 * lc2 * np.cos(theta2)))) + I2)
    phi2 = f32(f32(m2 * lc2 * g) * np.cos(f32(f32(theta1 + theta2) - f32(np.pi / 2.0))))
    phi1 = f32(f32(f32(f32(f32(-m2 * l1 * lc2) * f32(dtheta2 ** 2)) * np.sin(theta2))
                   - f32(f32(f32(f32(2 * m2 * l1 * lc2) * dtheta2) * dtheta1) * np.sin(theta2)))
               + f32(f32(f32(m1 * lc1 + m2 * l1) * g) * np.cos(f32(theta1 - f32(np.pi / 2)))))
    phi1 = f32(phi1 + phi2)
    ddtheta2 = f32(f32(f32(f32(a + f32(f32(d2 / d1) * phi1))
                           - f32(f32(f32(m2 * l1 * lc2) * f32(dtheta1 ** 2)) * np.sin(theta2))) - phi2)
                   / f32(f32(f32(m2 * lc2 ** 2) + I2) - f32(f32(d2 ** 2) / d1)))
    ddtheta1 = f32(-f32(f32(d2 * ddtheta2) + phi1) / d1)
    return np.array([dtheta1, dtheta2, ddtheta1, ddtheta2, 0.0], f32)


def _acrobot_wrap(x, m, M):
    """acrobot.py:236-247"""
    diff = f32(M - m)
    go_up, go_down = f32(x < m), f32(x >= M)
    how_often = f32(go_up * np.ceil(f32(f32(m - x) / diff))
                    + go_down * np.floor(f32(f32(f32(x - M) / diff) + f32(1))))
    return f32(f32(x - f32(f32(how_often * diff) * go_down)) + f32(f32(how_often * diff) * go_up))


def acrobot_step(s, time, action):
    """acrobot.py:52-106, 137-147, 250-258"""
    torque = f32(action - 1)
    y0 = np.array([s[0], s[1], s[2], s[3], torque], f32)
    dt = f32(0.2)
    dt2 = f32(dt / f32(2.0))
    k1 = _acrobot_dsdt(y0)
    k2 = _acrobot_dsdt((y0 + dt2 * k1).astype(f32))
    k3 = _acrobot_dsdt((y0 + dt2 * k2).astype(f32))
    k4 = _acrobot_dsdt((y0 + dt * k3).astype(f32))
    ns = (y0 + f32(dt / f32(6.0)) * (((k1 + f32(2) * k2).astype(f32) + f32(2) * k3).astype(f32) + k4).astype(f32)).astype(f32)
    pi = f32(np.pi)
    ja1, ja2 = _acrobot_wrap(ns[0], -pi, pi), _acrobot_wrap(ns[1], -pi, pi)
    v1 = np.clip(ns[2], f32(-4 * np.pi), f32(4 * np.pi))
    v2 = np.clip(ns[3], f32(-9 * np.pi), f32(9 * np.pi))
    done_angle = f32(-np.cos(ja1) - np.cos(f32(ja2 + ja1))) > f32(1.0)
    reward = f32(-1.0) * f32(1 - int(done_angle))
    st = np.array([ja1, ja2, v1, v2], f32)
    obs = np.array([np.cos(ja1), np.sin(ja1), np.cos(ja2), np.sin(ja2), v1, v2], f32)
    return st, time + 1, obs, reward, f32(done_angle), f32(time + 1 >= 5000)


def acrobot_no_term_step(s, time, action):
    """acrobot_no_term.py:52-108, 151-157: Acrobot's step with done = 0."""
    st, time, obs, reward, _, truncated = acrobot_step(s, time, action)
    return st, time, obs, reward, f32(0.0), truncated


ENV_STEP = {"CartPole-v1": cartpole_step, "MountainCar-v0": mountaincar_step,
            "Acrobot-v1": acrobot_step, "Acrobot-v2": acrobot_no_term_step}


# ---------------------------------------------------------------- networks
def swish(x):
    return (x * (f32(1) / (f32(1) + np.exp(-x)))).astype(f32)


def mlp_forward(layers, x):
    """networks/networks.py:25-48 with swish hidden activations."""
    h = x.astype(f32)
    for i, (w, b) in enumerate(layers):
        h = (h @ w.astype(f32) + b).astype(f32)
        if i != len(layers) - 1:
            h = swish(h)
    return h


def make_model_recurrent_fn(env_id, policy_layers, value_layers, mean, std, gamma=0.99):
    """alphazero.py:317-339 on the native (state, time) embedding."""
    step = ENV_STEP[env_id]

    def recurrent_fn(sim, action, emb):
        s, time = emb
        ns, ntime, obs, reward, terminated, truncated = step(s, time, action)
        x = ((obs - mean) / std).astype(f32)  # running_statistics.py:183-199
        logits = mlp_forward(policy_layers, x)
        value = mlp_forward(value_layers, x)
        discount = f32(gamma) * (f32(1) if truncated != 0 else f32(1) - terminated)
        return reward, f32(discount), logits, value, (ns, ntime)

    return recurrent_fn
