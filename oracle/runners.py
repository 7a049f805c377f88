"""Whole-policy runners over the two CPU checkers (TEST INFRASTRUCTURE ONLY).

A *case* is a dict describing one policy call with every random quantity
injected as an array, so that the C checker, the NumPy checker and the HIP
path can be driven with identical inputs:

  variant      'muzero' | 'sampled' | 'gumbel'
  B, S, A, K   lanes, simulations, actions, value atoms
  max_depth    None or int
  qtransform   None (variant default) or name; qt_kwargs dict of its options
  invalid      None or f32[B,A] (1 = invalid)
  prior_logits f32[B,A], value f32[B,K]
  evals        dict(prior_logits[B,S,A], value[B,S,K], reward[B,S], discount[B,S]), or
  model        dict(env_id, policy_layers, value_layers, mean, std, gamma) with
               root_state f32[B,4], root_time i32[B]
  dirichlet    f32[B,A]            (muzero, sampled)   fraction = dirichlet_fraction
  counts       f32[B,A]            (sampled: categorical draw counts), num_samples
  tie_noise    f32[B,S,Dn,A]       (muzero, sampled: uniform(0,1) tie-break draws)
  root_gumbel  f32[B,A]            (gumbel; already scaled)
  action_gumbel f32[B,A]           (muzero, sampled: final categorical draw)
  temperature, pb_c_init, pb_c_base, max_num_considered
"""
import numpy as np

from . import c_oracle as co
from . import mctx_numpy as mn

TREE_FIELDS = ("node_visits", "raw_values", "node_values", "parents", "action_from_parent",
               "children_index", "children_prior_logits", "children_visits", "children_rewards",
               "children_discounts", "children_values")


def _qt_defaults(case):
    qt = case.get("qtransform")
    if qt is None:
        qt = "completed_by_mix_value" if case["variant"] == "gumbel" else "parent_and_siblings"
    return qt, dict(case.get("qt_kwargs") or {})


def make_cfg(case):
    qt, kw = _qt_defaults(case)
    tie = case.get("tie_noise")
    return co.make_cfg(
        case["B"], case["S"], case["A"], case["K"], variant=case["variant"],
        max_depth=case.get("max_depth"), qtransform=qt,
        pb_c_init=case.get("pb_c_init", 1.25), pb_c_base=case.get("pb_c_base", 19652.0),
        q_epsilon=kw.get("epsilon", 1e-8), value_scale=kw.get("value_scale", 0.1),
        maxvisit_init=kw.get("maxvisit_init", 50.0),
        rescale_values=kw.get("rescale_values", True),
        use_mixed_value=kw.get("use_mixed_value", True),
        min_value=kw.get("min_value", 0.0), max_value=kw.get("max_value", 1.0),
        max_num_considered=case.get("max_num_considered", 16),
        noise_mode="injected" if tie is not None else "zero",
        noise_depth=0 if tie is None else tie.shape[2])


def root_prior(case):
    """Root prior array handed to search() and, for gumbel, the root Gumbel sample."""
    v = case["variant"]
    inv = case.get("invalid")
    if v == "muzero":
        return co.root_muzero(case["prior_logits"], case["dirichlet"],
                              case.get("dirichlet_fraction", 0.25), inv), None
    if v == "sampled":
        p, _ = co.root_sampled(case["prior_logits"], case["dirichlet"],
                               case.get("dirichlet_fraction", 0.25), inv,
                               case["num_samples"], case.get("sampling_temperature", 1.0),
                               counts=case["counts"])
        return p, None
    logits, g = co.root_gumbel(case["prior_logits"], inv, gumbel=case["root_gumbel"],
                               gumbel_scale=1.0)
    return logits, g


def run_policy_c(case):
    cfg = make_cfg(case)
    B, S, A, K = case["B"], case["S"], case["A"], case["K"]
    model = case.get("model")
    tree = co.alloc_tree(B, S, A, K, with_embedding=model is not None)
    prior, g = root_prior(case)
    inv = case.get("invalid")
    if model is not None:
        md = co.make_model(model["env_id"], model["policy_layers"], model["value_layers"],
                           model["mean"], model["std"], model.get("gamma", 0.99))
        co.tree_init(cfg, tree, prior, case["value"], case["root_state"], case["root_time"])
        sum_depth = co.search_model(cfg, tree, md, invalid=inv, root_gumbel=g,
                                    noise=case.get("tie_noise"))
    else:
        co.tree_init(cfg, tree, prior, case["value"])
        sum_depth = co.search_injected(cfg, tree, case["evals"], invalid=inv, root_gumbel=g,
                                       noise=case.get("tie_noise"))
    if case["variant"] == "gumbel":
        action, weights = co.epilogue_gumbel(cfg, tree, prior, g, inv)
    else:
        action, weights = co.epilogue_visits(cfg, tree, case.get("temperature", 1.0),
                                             gumbel=case["action_gumbel"])
    out = dict(tree)
    out.update(action=action, action_weights=weights, sum_depth=sum_depth, root_prior=prior)
    return out


def run_policy_numpy(case):
    """The same policy call on the NumPy checker; returns arrays stacked over lanes."""
    import functools
    B, S, A, K = case["B"], case["S"], case["A"], case["K"]
    v = case["variant"]
    qt_name, kw = _qt_defaults(case)
    qt = {"parent_and_siblings": mn.qtransform_by_parent_and_siblings,
          "completed_by_mix_value": mn.qtransform_completed_by_mix_value,
          "min_max": mn.qtransform_by_min_max}[qt_name]
    qt = functools.partial(qt, **kw)
    model = case.get("model")
    lanes, actions, weights, priors = [], [], [], []
    total_depth = 0
    for b in range(B):
        inv_in = None if case.get("invalid") is None else case["invalid"][b]
        inv = np.zeros(A, np.float32) if inv_in is None else inv_in
        if v == "gumbel":
            prior = mn.mask_invalid_actions(case["prior_logits"][b], inv_in)
            gumbel = case["root_gumbel"][b]
        else:
            prior = mn.root_logits_muzero(case["prior_logits"][b], case["dirichlet"][b],
                                          case.get("dirichlet_fraction", 0.25), inv_in)
            if v == "sampled":
                temp = case.get("sampling_temperature", 1.0)
                emp = (case["counts"][b] / np.float32(case["num_samples"])).astype(np.float32)
                prior = (emp * mn.softmax(prior) ** np.float32(1 - 1 / temp)).astype(np.float32)

        def select(tree, node, depth, sim, b=b, inv=inv, gumbel=None if v != "gumbel" else gumbel):
            if v == "gumbel":
                if depth == 0:
                    return mn.gumbel_root_action_selection(
                        tree, node, invalid=inv, gumbel=gumbel, num_simulations=S,
                        max_num_considered_actions=case.get("max_num_considered", 16),
                        qtransform=qt)
                return mn.gumbel_interior_action_selection(tree, node, qtransform=qt)
            tie = case.get("tie_noise")
            noise = (tie[b, sim, depth] if tie is not None and depth < tie.shape[2]
                     else np.zeros(A, np.float32))
            return mn.puct_action_selection(
                noise, tree, node, depth, invalid=inv, sampled=(v == "sampled"),
                pb_c_init=case.get("pb_c_init", 1.25), pb_c_base=case.get("pb_c_base", 19652.0),
                qtransform=qt)

        if model is not None:
            rf = mn.make_model_recurrent_fn(model["env_id"], model["policy_layers"],
                                            model["value_layers"], model["mean"], model["std"],
                                            model.get("gamma", 0.99))
            emb = (case["root_state"][b].copy(), int(case["root_time"][b]))
        else:
            ev = case["evals"]

            def rf(sim, action, emb, b=b, ev=ev):
                return (ev["reward"][b, sim], ev["discount"][b, sim], ev["prior_logits"][b, sim],
                        ev["value"][b, sim], None)
            emb = None
        tree, sd = mn.search_lane(root_logits=prior, root_value=case["value"][b],
                                  root_embedding=emb, recurrent_fn=rf, num_simulations=S,
                                  select=select, max_depth=case.get("max_depth"))
        total_depth += sd
        if v == "gumbel":
            a, w = mn.gumbel_epilogue(tree, prior, gumbel, inv_in, qt)
        else:
            a, w = mn.visit_epilogue(tree, case.get("temperature", 1.0), case["action_gumbel"][b])
        lanes.append(tree)
        actions.append(a)
        weights.append(w)
        priors.append(prior)
    out = {f: np.stack([getattr(t, f) for t in lanes]) for f in TREE_FIELDS}
    if model is not None:
        out["emb_state"] = np.stack([
            np.stack([e[0] if e is not None else np.zeros(4, np.float32) for e in t.embeddings])
            for t in lanes])
        out["emb_time"] = np.stack([
            np.array([e[1] if e is not None else 0 for e in t.embeddings], np.int32)
            for t in lanes])
    out.update(action=np.array(actions, np.int32), action_weights=np.stack(weights),
               sum_depth=total_depth, root_prior=np.stack(priors))
    return out


# ---------------------------------------------------------------- stochastic MuZero
def synthetic_stochastic_case(B, S, A, C, seed=0, *, max_depth=None, with_invalid=False):
    """Random inputs for one stochastic_muzero_policy call: what the decision and the chance
    node function return per (lane, simulation), scalar values (mctx_dist/_src/base.py:103-139)."""
    rng = np.random.default_rng(seed)
    f = np.float32
    case = dict(variant="stochastic", B=B, S=S, A=A, C=C, K=1, max_depth=max_depth, invalid=None,
                temperature=1.0)
    if with_invalid:
        inv = (rng.uniform(size=(B, A)) < 0.3).astype(f)
        inv[0, :] = 1.0
        if B > 1:
            inv[1, :] = 0.0
        case["invalid"] = inv
    case["prior_logits"] = rng.normal(size=(B, A)).astype(f)
    case["value"] = rng.uniform(-1, 1, size=(B, 1)).astype(f)
    case["evals"] = dict(
        chance_logits=rng.normal(size=(B, S, C)).astype(f),
        afterstate_value=rng.uniform(-1, 1, size=(B, S, 1)).astype(f),
        action_logits=rng.normal(size=(B, S, A)).astype(f),
        value=rng.uniform(-1, 1, size=(B, S, 1)).astype(f),
        reward=rng.uniform(0, 1, size=(B, S)).astype(f),
        discount=np.full((B, S), 0.99, f))
    case["dirichlet"] = rng.dirichlet([0.3] * A, size=B).astype(f)
    u = rng.uniform(1e-6, 1.0, size=(B, A))
    case["action_gumbel"] = (-np.log(-np.log(u))).astype(f)
    case["tie_noise"] = rng.uniform(size=(B, S, max_depth or S, A)).astype(f)
    return case


def _mask_decision(tree, A):
    """policies.py:476: the decision columns of every per-action array."""
    out = {}
    for k, v in tree.items():
        out[k] = np.ascontiguousarray(v[:, :, :A]) if k.startswith("children_") else v
    return out


def run_stochastic_c(case):
    """mctx_dist/_src/policies.py:346-483 on the C checker."""
    B, S, A, C_ = case["B"], case["S"], case["A"], case["C"]
    noise_mode = "injected" if case.get("tie_noise") is not None else "zero"
    if case.get("search_key") is not None:  # the device generator keyed as search() keys it
        noise_mode = "threefry"
    cfg = co.make_cfg(B, S, A + C_, 1, variant="stochastic", max_depth=case.get("max_depth"),
                      noise_mode=noise_mode, key=case.get("search_key") or (0, 0),
                      noise_depth=0 if case.get("tie_noise") is None else case["tie_noise"].shape[2],
                      num_decision_actions=A)
    inv = case.get("invalid")
    prior = co.root_muzero(case["prior_logits"], case["dirichlet"], 0.25, inv)
    padded = np.concatenate([prior, np.full((B, C_), -np.inf, np.float32)], axis=-1)
    tree = co.alloc_tree(B, S, A + C_, 1)
    co.tree_init(cfg, tree, padded, case["value"])
    kind, sum_depth = co.search_stochastic(cfg, tree, case["evals"], invalid=inv, noise=case.get("tie_noise"))
    masked = _mask_decision(tree, A)
    mcfg = co.make_cfg(B, S, A, 1, variant="muzero")
    action, weights = co.epilogue_visits(mcfg, masked, case.get("temperature", 1.0),
                                         gumbel=case["action_gumbel"])
    out = dict(masked)
    out.update(action=action, action_weights=weights, sum_depth=sum_depth, root_prior=prior,
               is_decision=kind, full_tree=tree)
    return out


def run_stochastic_numpy(case):
    """The same call on the NumPy checker."""
    B, S, A, C_ = case["B"], case["S"], case["A"], case["C"]
    ev = case["evals"]
    lanes, actions, weights, priors, kinds = [], [], [], [], []
    total_depth = 0
    for b in range(B):
        inv_in = None if case.get("invalid") is None else case["invalid"][b]
        inv = np.zeros(A, np.float32) if inv_in is None else inv_in
        prior = mn.root_logits_muzero(case["prior_logits"][b], case["dirichlet"][b], 0.25, inv_in)
        padded = np.concatenate([prior, np.full(C_, -np.inf, np.float32)])

        def select(tree, node, depth, sim, b=b, inv=inv):
            tie = case.get("tie_noise")
            noise = (tie[b, sim, depth] if tie is not None and depth < tie.shape[2]
                     else np.zeros(A, np.float32))
            return mn.stochastic_action_selection(noise, tree, node, depth, invalid=inv, num_actions=A)

        def decision_fn(sim, action, state, b=b):
            return ev["chance_logits"][b, sim], ev["afterstate_value"][b, sim], state + 1

        def chance_fn(sim, outcome, afterstate, b=b):
            return (ev["action_logits"][b, sim], ev["value"][b, sim], ev["reward"][b, sim],
                    ev["discount"][b, sim], afterstate + 10)
        rf = mn.make_stochastic_recurrent_fn(decision_fn, chance_fn, A, C_)
        tree, sd = mn.search_lane(root_logits=padded, root_value=case["value"][b],
                                  root_embedding=(0, 1, True), recurrent_fn=rf, num_simulations=S,
                                  select=select, max_depth=case.get("max_depth"))
        total_depth += sd
        kinds.append(np.array([e[2] if e is not None else False for e in tree.embeddings], np.uint8))
        a, w = mn.visit_epilogue(mn.MaskedLaneTree(tree, 0, A), case.get("temperature", 1.0),
                                 case["action_gumbel"][b])
        lanes.append(tree)
        actions.append(a)
        weights.append(w)
        priors.append(prior)
    out = {f: np.stack([getattr(t, f) for t in lanes]) for f in TREE_FIELDS}
    out = _mask_decision(out, A)
    out.update(action=np.array(actions, np.int32), action_weights=np.stack(weights),
               sum_depth=total_depth, root_prior=np.stack(priors), is_decision=np.stack(kinds))
    return out


# ---------------------------------------------------------------- synthetic cases
def lecun_uniform_mlp(rng, sizes):
    """flax lecun_uniform kernels U(+-sqrt(3/fan_in)), zero biases (networks.py:29)."""
    layers = []
    for i in range(len(sizes) - 1):
        lim = np.sqrt(3.0 / sizes[i])
        w = rng.uniform(-lim, lim, size=(sizes[i], sizes[i + 1])).astype(np.float32)
        layers.append((w, np.zeros(sizes[i + 1], np.float32)))
    return layers


ENV_DIMS = {"CartPole-v1": (4, 2), "MountainCar-v0": (2, 3), "Acrobot-v1": (6, 3), "Acrobot-v2": (6, 3)}


def synthetic_case(variant, B, S, A=None, K=1, seed=0, *, env_id=None, max_depth=None,
                   with_invalid=False, qtransform=None, qt_kwargs=None, num_samples=16,
                   policy_hidden=(32, 32, 32, 32), value_hidden=(256,) * 5, tie_depth=None,
                   random_bias=False):
    """Random inputs for one policy call (injected evaluations unless env_id is given)."""
    rng = np.random.default_rng(seed)
    if env_id is not None:
        obs_dim, A = ENV_DIMS[env_id]
    case = dict(variant=variant, B=B, S=S, A=A, K=K, max_depth=max_depth,
                qtransform=qtransform, qt_kwargs=qt_kwargs)
    if with_invalid:
        inv = (rng.uniform(size=(B, A)) < 0.3).astype(np.float32)
        inv[0, :] = 1.0           # an all-invalid lane
        if B > 1:
            inv[1, :] = 0.0       # an all-valid lane
        case["invalid"] = inv
    else:
        case["invalid"] = None
    if env_id is None:
        case["prior_logits"] = rng.normal(size=(B, A)).astype(np.float32)
        case["value"] = rng.uniform(-1, 1, size=(B, K)).astype(np.float32)
        case["evals"] = dict(
            prior_logits=rng.normal(size=(B, S, A)).astype(np.float32),
            value=rng.uniform(-1, 1, size=(B, S, K)).astype(np.float32),
            reward=rng.uniform(0, 1, size=(B, S)).astype(np.float32),
            discount=np.full((B, S), 0.99, np.float32))
    else:
        pol = lecun_uniform_mlp(rng, (obs_dim,) + tuple(policy_hidden) + (A,))
        val = lecun_uniform_mlp(rng, (obs_dim,) + tuple(value_hidden) + (K,))
        if random_bias:
            pol = [(w, rng.normal(scale=0.1, size=b.shape).astype(np.float32)) for w, b in pol]
            val = [(w, rng.normal(scale=0.1, size=b.shape).astype(np.float32)) for w, b in val]
        mean = rng.normal(scale=0.05, size=obs_dim).astype(np.float32)
        std = rng.uniform(0.5, 1.5, size=obs_dim).astype(np.float32)
        case["model"] = dict(env_id=env_id, policy_layers=pol, value_layers=val, mean=mean,
                             std=std, gamma=0.99)
        keys = rng.integers(0, 2 ** 32, size=(B, 2), dtype=np.uint64).astype(np.uint32)
        rs = co.env_reset(env_id, keys)
        case["root_state"], case["root_time"] = rs["state"], rs["time"]
        md = co.make_model(env_id, pol, val, mean, std)
        case["prior_logits"], case["value"] = co.forward(md, rs["obs"])
    if variant == "gumbel":
        u = rng.uniform(1e-6, 1.0, size=(B, A))
        case["root_gumbel"] = (-np.log(-np.log(u))).astype(np.float32)
    else:
        case["dirichlet"] = rng.dirichlet([0.3] * A, size=B).astype(np.float32)
        u = rng.uniform(1e-6, 1.0, size=(B, A))
        case["action_gumbel"] = (-np.log(-np.log(u))).astype(np.float32)
        md_ = tie_depth if tie_depth is not None else (max_depth or S)
        case["tie_noise"] = rng.uniform(size=(B, S, md_, A)).astype(np.float32)
        case["temperature"] = 1.0
    if variant == "sampled":
        case["num_samples"] = num_samples
        p = np.exp(case["prior_logits"] - case["prior_logits"].max(-1, keepdims=True))
        p /= p.sum(-1, keepdims=True)
        case["counts"] = np.stack([rng.multinomial(num_samples, pi) for pi in p]).astype(np.float32)
    return case
