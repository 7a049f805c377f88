"""ctypes front end of the plain-C CPU checker (oracle/libaz_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of ``bench.py``; the product package never
imports it.  PARITY UNPINNED (see mcts_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAX_LAYERS = 8
VARIANTS = {"muzero": 0, "sampled": 1, "gumbel": 2, "stochastic": 3}
QTRANSFORMS = {"parent_and_siblings": 0, "completed_by_mix_value": 1, "min_max": 2}
ENVS = {"none": 0, "CartPole-v1": 1, "MountainCar-v0": 2, "Acrobot-v1": 3, "Acrobot-v2": 4}
NOISE = {"threefry": 0, "injected": 1, "zero": 2}


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libaz_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


class Cfg(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("S", C.c_int32), ("A", C.c_int32), ("K", C.c_int32),
        ("variant", C.c_int32), ("max_depth", C.c_int32), ("qtransform", C.c_int32),
        ("pb_c_init", C.c_float), ("pb_c_base", C.c_float), ("q_epsilon", C.c_float),
        ("value_scale", C.c_float), ("maxvisit_init", C.c_float),
        ("rescale_values", C.c_int32), ("use_mixed_value", C.c_int32),
        ("min_value", C.c_float), ("max_value", C.c_float),
        ("max_num_considered", C.c_int32), ("noise_mode", C.c_int32),
        ("noise_depth", C.c_int32), ("key", C.c_uint32 * 2), ("num_decision_actions", C.c_int32),
    ]


class TreeS(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "node_visits", "raw_values", "node_values", "parents", "action_from_parent",
        "children_index", "children_prior_logits", "children_visits", "children_rewards",
        "children_discounts", "children_values", "emb_state", "emb_time")]


class Mlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("sizes", C.c_int32 * (MAX_LAYERS + 1)),
                ("w", C.c_void_p * MAX_LAYERS), ("b", C.c_void_p * MAX_LAYERS)]


class Model(C.Structure):
    _fields_ = [("env_id", C.c_int32), ("gamma", C.c_float), ("policy", Mlp), ("value", Mlp),
                ("norm_mean", C.c_void_p), ("norm_std", C.c_void_p)]


class Injected(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("prior_logits", "value", "reward", "discount")]


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def make_cfg(B, S, A, K=1, variant="gumbel", max_depth=None, qtransform=None,
             pb_c_init=1.25, pb_c_base=19652.0, q_epsilon=1e-8, value_scale=0.1,
             maxvisit_init=50.0, rescale_values=True, use_mixed_value=True,
             min_value=0.0, max_value=1.0, max_num_considered=16, noise_mode="zero",
             noise_depth=0, key=(0, 0), num_decision_actions=0):
    if qtransform is None:
        qtransform = "completed_by_mix_value" if variant == "gumbel" else "parent_and_siblings"
    c = Cfg()
    c.B, c.S, c.A, c.K = B, S, A, K
    c.variant = VARIANTS[variant]
    c.max_depth = S if max_depth is None else max_depth
    c.qtransform = QTRANSFORMS[qtransform]
    c.pb_c_init, c.pb_c_base, c.q_epsilon = pb_c_init, pb_c_base, q_epsilon
    c.value_scale, c.maxvisit_init = value_scale, maxvisit_init
    c.rescale_values, c.use_mixed_value = int(rescale_values), int(use_mixed_value)
    c.min_value, c.max_value = min_value, max_value
    c.max_num_considered = max_num_considered
    c.noise_mode = NOISE[noise_mode]
    c.noise_depth = noise_depth
    c.key[0], c.key[1] = int(key[0]), int(key[1])
    c.num_decision_actions = num_decision_actions
    return c


def alloc_tree(B, S, A, K, with_embedding=False):
    N = S + 1
    t = {
        "node_visits": np.zeros((B, N), np.int32),
        "raw_values": np.zeros((B, N, K), np.float32),
        "node_values": np.zeros((B, N, K), np.float32),
        "parents": np.zeros((B, N), np.int32),
        "action_from_parent": np.zeros((B, N), np.int32),
        "children_index": np.zeros((B, N, A), np.int32),
        "children_prior_logits": np.zeros((B, N, A), np.float32),
        "children_visits": np.zeros((B, N, A), np.int32),
        "children_rewards": np.zeros((B, N, A), np.float32),
        "children_discounts": np.zeros((B, N, A), np.float32),
        "children_values": np.zeros((B, N, A, K), np.float32),
    }
    if with_embedding:
        t["emb_state"] = np.zeros((B, N, 4), np.float32)
        t["emb_time"] = np.zeros((B, N), np.int32)
    return t


def _tree_struct(t):
    s = TreeS()
    for name, _ in TreeS._fields_:
        setattr(s, name, _p(t.get(name)))
    return s


def make_mlp(layers):
    """layers: list of (kernel[in,out], bias[out]) float32 arrays (flax layout)."""
    m = Mlp()
    keep = []
    m.n_layers = len(layers)
    m.sizes[0] = layers[0][0].shape[0]
    for i, (w, b) in enumerate(layers):
        w, b = _f32(w), _f32(b)
        keep += [w, b]
        m.sizes[i + 1] = w.shape[1]
        m.w[i] = _p(w)
        m.b[i] = _p(b)
    return m, keep


def make_model(env_id, policy_layers, value_layers, norm_mean, norm_std, gamma=0.99):
    md = Model()
    md.env_id = ENVS[env_id] if isinstance(env_id, str) else env_id
    md.gamma = gamma
    md.policy, k1 = make_mlp(policy_layers)
    md.value, k2 = make_mlp(value_layers)
    mean, std = _f32(norm_mean), _f32(norm_std)
    md.norm_mean, md.norm_std = _p(mean), _p(std)
    md._keep = k1 + k2 + [mean, std]
    return md


def considered_table(max_m, S):
    tab = np.zeros((max_m + 1, S), np.int32)
    lib().oz_considered_table(C.c_int(max_m), C.c_int(S), _p(tab))
    return tab


def tree_init(cfg, tree, root_logits, root_value, root_state=None, root_time=None):
    rl, rv = _f32(root_logits), _f32(root_value).reshape(cfg.B, cfg.K)
    rs, rt = _f32(root_state), _i32(root_time)
    ts = _tree_struct(tree)
    lib().oz_tree_init(C.byref(cfg), C.byref(ts), _p(rl), _p(rv), _p(rs), _p(rt))


def search_injected(cfg, tree, evals, invalid=None, root_gumbel=None, noise=None):
    """evals: dict(prior_logits[B,S,A], value[B,S,K], reward[B,S], discount[B,S])."""
    inj = Injected()
    keep = {k: _f32(v) for k, v in evals.items()}
    for k in ("prior_logits", "value", "reward", "discount"):
        setattr(inj, k, _p(keep[k]))
    inv, g, nz = _f32(invalid), _f32(root_gumbel), _f32(noise)
    ts = _tree_struct(tree)
    sd = C.c_int64(0)
    lib().oz_search_injected(C.byref(cfg), C.byref(ts), _p(inv), _p(g), _p(nz), C.byref(inj),
                             C.byref(sd))
    return sd.value


class InjectedStochastic(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("chance_logits", "afterstate_value", "action_logits", "value",
                                           "reward", "discount")]


def search_stochastic(cfg, tree, evals, invalid=None, noise=None):
    """evals: dict(chance_logits[B,S,C], afterstate_value[B,S,K], action_logits[B,S,AD], value[B,S,K],
    reward[B,S], discount[B,S]).  Returns (is_decision [B,N] uint8, sum of depths)."""
    inj = InjectedStochastic()
    keep = {k: _f32(v) for k, v in evals.items()}
    for k, _ in InjectedStochastic._fields_:
        setattr(inj, k, _p(keep[k]))
    inv, nz = _f32(invalid), _f32(noise)
    ts = _tree_struct(tree)
    kind = np.zeros((cfg.B, cfg.S + 1), np.uint8)
    sd = C.c_int64(0)
    lib().oz_search_stochastic(C.byref(cfg), C.byref(ts), _p(inv), _p(nz), C.byref(inj), _p(kind),
                               C.byref(sd))
    return kind, sd.value


def search_model(cfg, tree, model, invalid=None, root_gumbel=None, noise=None):
    inv, g, nz = _f32(invalid), _f32(root_gumbel), _f32(noise)
    ts = _tree_struct(tree)
    sd = C.c_int64(0)
    lib().oz_search_model(C.byref(cfg), C.byref(ts), _p(inv), _p(g), _p(nz), C.byref(model),
                          C.byref(sd))
    return sd.value


def forward(model, obs):
    obs = _f32(obs)
    B = obs.shape[0]
    A = model.policy.sizes[model.policy.n_layers]
    K = model.value.sizes[model.value.n_layers]
    logits = np.zeros((B, A), np.float32)
    value = np.zeros((B, K), np.float32)
    lib().oz_forward(C.byref(model), C.c_int(B), _p(obs), _p(logits), _p(value))
    return logits, value


def env_step(env_id, state, time, action):
    eid = ENVS[env_id]
    state, time, action = _f32(state).copy(), _i32(time).copy(), _i32(action)
    B = state.shape[0]
    od = lib().oz_env_obs_dim(C.c_int(eid))
    obs = np.zeros((B, od), np.float32)
    reward, term, trunc = (np.zeros(B, np.float32) for _ in range(3))
    lib().oz_env_step(C.c_int(eid), C.c_int(B), _p(state), _p(time), _p(action), _p(obs),
                      _p(reward), _p(term), _p(trunc))
    return dict(state=state, time=time, obs=obs, reward=reward, terminated=term, truncated=trunc)


def env_reset(env_id, keys):
    eid = ENVS[env_id]
    keys = np.ascontiguousarray(keys, np.uint32)
    B = keys.shape[0]
    od = lib().oz_env_obs_dim(C.c_int(eid))
    state, time = np.zeros((B, 4), np.float32), np.zeros(B, np.int32)
    obs = np.zeros((B, od), np.float32)
    lib().oz_env_reset(C.c_int(eid), C.c_int(B), _p(keys), _p(state), _p(time), _p(obs))
    return dict(state=state, time=time, obs=obs)


def real_env_step(env_id, episode_length, st, action):
    """st: dict(state, time, obs, real_obs, reward, done, truncation, steps, rng); updated copy
    is returned."""
    eid = ENVS[env_id]
    o = {k: np.ascontiguousarray(v).copy() for k, v in st.items()}
    B = o["state"].shape[0]
    action = _i32(action)
    lib().oz_real_env_step(C.c_int(eid), C.c_int(B), C.c_int(episode_length), _p(o["state"]),
                           _p(o["time"]), _p(o["obs"]), _p(o["real_obs"]), _p(o["reward"]),
                           _p(o["done"]), _p(o["truncation"]), _p(o["steps"]), _p(o["rng"]),
                           _p(action))
    return o


def root_muzero(prior_logits, dirichlet, fraction, invalid=None):
    pl, dn, inv = _f32(prior_logits), _f32(dirichlet), _f32(invalid)
    B, A = pl.shape
    out = np.zeros((B, A), np.float32)
    lib().oz_root_muzero(C.c_int(B), C.c_int(A), _p(pl), _p(dn), C.c_float(fraction), _p(inv),
                         _p(out))
    return out


def root_sampled(prior_logits, dirichlet, fraction, invalid, num_samples, temperature,
                 counts=None, key=None):
    pl, dn, inv, cn = _f32(prior_logits), _f32(dirichlet), _f32(invalid), _f32(counts)
    B, A = pl.shape
    out = np.zeros((B, A), np.float32)
    out_counts = np.zeros((B, A), np.float32)
    k = np.asarray(key if key is not None else (0, 0), np.uint32)
    exponent = np.float32(1 - 1 / temperature)
    lib().oz_root_sampled(C.c_int(B), C.c_int(A), _p(pl), _p(dn), C.c_float(fraction), _p(inv),
                          C.c_int(num_samples), C.c_float(temperature), C.c_float(exponent),
                          _p(cn), _p(k), _p(out), _p(out_counts))
    return out, out_counts


def root_gumbel(prior_logits, invalid=None, gumbel=None, key=None, gumbel_scale=1.0):
    pl, inv, g = _f32(prior_logits), _f32(invalid), _f32(gumbel)
    B, A = pl.shape
    out, og = np.zeros((B, A), np.float32), np.zeros((B, A), np.float32)
    k = np.asarray(key if key is not None else (0, 0), np.uint32)
    lib().oz_root_gumbel(C.c_int(B), C.c_int(A), _p(pl), _p(inv), _p(g), _p(k),
                         C.c_float(gumbel_scale), _p(out), _p(og))
    return out, og


def epilogue_visits(cfg, tree, temperature, gumbel=None, key=None):
    g = _f32(gumbel)
    k = np.asarray(key if key is not None else (0, 0), np.uint32)
    action = np.zeros(cfg.B, np.int32)
    aw = np.zeros((cfg.B, cfg.A), np.float32)
    ts = _tree_struct(tree)
    lib().oz_epilogue_visits(C.byref(cfg), C.byref(ts), C.c_float(temperature), _p(g), _p(k),
                             _p(action), _p(aw))
    return action, aw


def epilogue_gumbel(cfg, tree, root_logits, gumbel, invalid=None):
    rl, g, inv = _f32(root_logits), _f32(gumbel), _f32(invalid)
    action = np.zeros(cfg.B, np.int32)
    aw = np.zeros((cfg.B, cfg.A), np.float32)
    ts = _tree_struct(tree)
    lib().oz_epilogue_gumbel(C.byref(cfg), C.byref(ts), _p(rl), _p(g), _p(inv), _p(action),
                             _p(aw))
    return action, aw


def summary(cfg, tree):
    B, A, K = cfg.B, cfg.A, cfg.K
    vc, vp = np.zeros((B, A), np.float32), np.zeros((B, A), np.float32)
    val, q = np.zeros((B, K), np.float32), np.zeros((B, A, K), np.float32)
    ts = _tree_struct(tree)
    lib().oz_summary(C.byref(cfg), C.byref(ts), _p(vc), _p(vp), _p(val), _p(q))
    return dict(visit_counts=vc, visit_probs=vp, value=val, qvalues=q)


def actor_post(cfg, tree, action_weights, sampled_action, deterministic, prior_logits):
    aw, sa, pl = _f32(action_weights), _i32(sampled_action), _f32(prior_logits)
    action = np.zeros(cfg.B, np.int32)
    sv = np.zeros((cfg.B, cfg.K), np.float32)
    lp = np.zeros(cfg.B, np.float32)
    ts = _tree_struct(tree)
    lib().oz_actor_post(C.byref(cfg), C.byref(ts), _p(aw), _p(sa), C.c_int(int(deterministic)),
                        _p(pl), _p(action), _p(sv), _p(lp))
    return action, sv, lp


_MATH_FN = {"exp": 0, "log": 1, "sin": 2, "cos": 3, "swish": 4, "sqrt": 5, "rcp": 6,
            "gumbel_from_bits": 7, "sin_select": 9, "cos_select": 10}


def math_apply(fn, x):
    x = _f32(x)
    y = np.zeros_like(x)
    lib().oz_math_apply(C.c_int(_MATH_FN[fn]), C.c_int(x.size), _p(x), _p(y))
    return y


def powf(x, e):
    x = _f32(x)
    y = np.zeros_like(x)
    lib().oz_powf(C.c_int(x.size), _p(x), C.c_float(float(e)), _p(y))
    return y


def random_bits(key, n):
    key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros(n, np.uint32)
    lib().oz_random_bits(_p(key), C.c_int(n), _p(out))
    return out


def uniform01(key, n):
    key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros(n, np.float32)
    lib().oz_uniform01(_p(key), C.c_int(n), _p(out))
    return out


def threefry(key, ctr):
    key = np.ascontiguousarray(key, np.uint32).reshape(-1, 2)
    ctr = np.ascontiguousarray(ctr, np.uint32).reshape(-1, 2)
    out = np.zeros_like(key)
    lib().oz_threefry(C.c_int(key.shape[0]), _p(key), _p(ctr), _p(out))
    return out


def key_split(key, n):
    key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros((n, 2), np.uint32)
    lib().oz_key_split(_p(key), C.c_int(n), _p(out))
    return out
