"""Known-answer tests that pin the CPU checkers.

The reference ships no tests or golden vectors, so these cases are derived by
hand from its source (file:line cited per test).  Both checkers -- the plain-C
one and the NumPy one -- must reproduce them.
"""
import numpy as np
import pytest

from oracle import mctx_numpy as mn
from oracle import runners


# ---- KAT-1: sequential halving schedule (mctx_dist/_src/seq_halving.py:36-88)
def _rep(pairs):
    out = []
    for count, value in pairs:
        out += [value] * count
    return tuple(out)


SEQ_KATS = [
    (1, 5, (0, 1, 2, 3, 4)),
    (0, 3, (0, 1, 2)),
    (2, 4, (0, 0, 1, 1)),
    (4, 8, (0, 0, 0, 0, 1, 1, 2, 2)),
    (2, 64, _rep([(2, v) for v in range(32)])),
    (3, 64, _rep([(3, v) for v in range(10)] + [(2, v) for v in range(10, 27)])),
    (16, 64, _rep([(16, 0), (8, 1), (8, 2)] + [(4, v) for v in (3, 4, 5, 6)]
                  + [(2, v) for v in range(7, 15)])),
]


@pytest.mark.parametrize("m,n,expected", SEQ_KATS)
def test_kat1_considered_visits(oracle, m, n, expected):
    assert len(expected) == n
    assert mn.get_sequence_of_considered_visits(m, n) == expected
    table = oracle.considered_table(max(m, 1), n)
    assert tuple(int(v) for v in table[m]) == expected


def test_kat1_table_matches_between_checkers(oracle):
    for m, n in [(16, 64), (16, 256), (5, 17), (8, 3), (3, 32)]:
        assert np.array_equal(oracle.considered_table(m, n),
                              np.array(mn.get_table_of_considered_visits(m, n), np.int32))


# ---- KAT-2: fresh tree (mctx_dist/_src/search.py:345-387)
def test_kat2_fresh_tree(oracle):
    B, S, A, K = 3, 5, 4, 2
    cfg = oracle.make_cfg(B, S, A, K)
    tree = oracle.alloc_tree(B, S, A, K)
    for v in tree.values():
        v[...] = 7  # poison: init must overwrite everything
    logits = np.arange(B * A, dtype=np.float32).reshape(B, A)
    value = np.arange(B * K, dtype=np.float32).reshape(B, K) + 0.5
    oracle.tree_init(cfg, tree, logits, value)
    assert np.array_equal(tree["node_visits"], np.array([[1, 0, 0, 0, 0, 0]] * B))
    assert (tree["parents"] == -1).all() and (tree["action_from_parent"] == -1).all()
    assert (tree["children_index"] == -1).all()
    assert np.array_equal(tree["raw_values"][:, 0], value)
    assert np.array_equal(tree["node_values"][:, 0], value)
    assert np.array_equal(tree["children_prior_logits"][:, 0], logits)
    assert (tree["raw_values"][:, 1:] == 0).all() and (tree["node_values"][:, 1:] == 0).all()
    assert (tree["children_prior_logits"][:, 1:] == 0).all()
    for f in ("children_visits", "children_rewards", "children_discounts", "children_values"):
        assert (tree[f] == 0).all()


# ---- KAT-3: first simulation (search.py:88-105, 231-244, 265-286)
@pytest.mark.parametrize("variant", ["muzero", "sampled", "gumbel"])
def test_kat3_first_simulation(variant):
    case = runners.synthetic_case(variant, B=4, S=1, A=3, K=2, seed=3)
    for run in (runners.run_policy_c, runners.run_policy_numpy):
        out = run(case)
        for b in range(4):
            a = int(out["action_from_parent"][b, 1])
            assert 0 <= a < 3
            assert out["children_index"][b, 0, a] == 1
            assert out["parents"][b, 1] == 0
            assert list(out["node_visits"][b]) == [2, 1]
            assert out["children_visits"][b, 0, a] == 1
            assert out["children_visits"][b, 0].sum() == 1
            v1 = case["evals"]["value"][b, 0]
            r, d = case["evals"]["reward"][b, 0], case["evals"]["discount"][b, 0]
            assert np.array_equal(out["children_values"][b, 0, a], v1)
            v0 = case["value"][b]
            expect = ((v0 * np.float32(1) + (r + d * v1).astype(np.float32)).astype(np.float32)
                      / np.float32(2)).astype(np.float32)
            assert np.array_equal(out["node_values"][b, 0], expect)
            assert np.array_equal(out["raw_values"][b, 0], v0)
            assert out["children_rewards"][b, 0, a] == r
            assert out["children_discounts"][b, 0, a] == d


# ---- KAT-4: structural invariants after S simulations
def check_tree_invariants(out, S, depth_capped=False):
    nv, cv, ci = out["node_visits"], out["children_visits"], out["children_index"]
    par, afp = out["parents"], out["action_from_parent"]
    B, N = nv.shape
    assert (nv[:, 0] == S + 1).all()
    assert (cv[:, 0].sum(-1) == S).all()
    for b in range(B):
        created = [i for i in range(1, N) if par[b, i] != -1]
        if not depth_capped:
            assert created == list(range(1, N))  # nodes are numbered in creation order
        for i in created:
            assert 0 <= par[b, i] < i
            assert ci[b, par[b, i], afp[b, i]] == i
            if not depth_capped:
                assert nv[b, i] == 1 + cv[b, i].sum()
        # every child pointer refers back to its parent
        for n in range(N):
            for a in range(ci.shape[2]):
                c = ci[b, n, a]
                if c != -1:
                    assert par[b, c] == n and afp[b, c] == a
                else:
                    assert cv[b, n, a] == 0


@pytest.mark.parametrize("variant", ["muzero", "sampled", "gumbel"])
def test_kat4_invariants(variant):
    S = 24
    case = runners.synthetic_case(variant, B=6, S=S, A=4, K=1, seed=11)
    for run in (runners.run_policy_c, runners.run_policy_numpy):
        check_tree_invariants(run(case), S)


def test_kat4_gumbel_two_actions_even_split():
    """A=2, S=64, m=16: sequential halving visits both root actions exactly 32 times."""
    case = runners.synthetic_case("gumbel", B=8, S=64, A=2, K=1, seed=5)
    out = runners.run_policy_c(case)
    assert (out["children_visits"][:, 0] == 32).all()


# ---- KAT-5: single environment steps
def test_kat5_cartpole_step(oracle):
    """cartpole.py:49-90 from the all-zero state with action 1."""
    o = oracle.env_step("CartPole-v1", np.zeros((1, 4)), np.zeros(1), np.array([1]))
    np.testing.assert_allclose(o["state"][0], [0.0, 0.195122, 0.0, -0.292683], rtol=2e-6, atol=1e-9)
    assert o["reward"][0] == 1.0 and o["terminated"][0] == 0.0 and o["truncated"][0] == 0.0
    assert o["time"][0] == 1
    s, t, obs, r, term, trunc = mn.cartpole_step(np.zeros(4, np.float32), 0, 1)
    np.testing.assert_allclose(s, o["state"][0], rtol=1e-6, atol=1e-9)
    assert (r, term, trunc) == (1.0, 0.0, 0.0)


def test_kat5_cartpole_reward_uses_previous_terminal(oracle):
    """cartpole.py:53,77: reward 0 only once the *previous* state was terminal."""
    st = np.array([[2.39, 5.0, 0.0, 0.0]], np.float32)
    o1 = oracle.env_step("CartPole-v1", st, np.zeros(1), np.array([1]))
    assert o1["terminated"][0] == 1.0 and o1["reward"][0] == 1.0
    o2 = oracle.env_step("CartPole-v1", o1["state"], o1["time"], np.array([1]))
    assert o2["reward"][0] == 0.0
    # truncation: time >= 50000 and not terminated
    o3 = oracle.env_step("CartPole-v1", np.zeros((1, 4)), np.array([49999]), np.array([0]))
    assert o3["truncated"][0] == 1.0 and o3["terminated"][0] == 0.0


def test_kat5_mountaincar_step(oracle):
    """mountain_car.py:43-72 from (-0.5, 0) with action 2."""
    o = oracle.env_step("MountainCar-v0", np.array([[-0.5, 0, 0, 0]]), np.zeros(1), np.array([2]))
    np.testing.assert_allclose(o["state"][0, 1], 0.000823157, rtol=1e-5)
    np.testing.assert_allclose(o["state"][0, 0], -0.499176843, rtol=1e-6)
    assert o["reward"][0] == -1.0 and o["terminated"][0] == 0.0
    s, *_ = mn.mountaincar_step(np.array([-0.5, 0, 0, 0], np.float32), 0, 2)
    np.testing.assert_allclose(s[:2], o["state"][0, :2], rtol=1e-6)
    # left wall: position clipped to -1.2 and negative velocity zeroed
    o = oracle.env_step("MountainCar-v0", np.array([[-1.2, -0.07, 0, 0]]), np.zeros(1), np.array([0]))
    assert o["state"][0, 0] == np.float32(-1.2) and o["state"][0, 1] == 0.0
    # goal
    o = oracle.env_step("MountainCar-v0", np.array([[0.49, 0.07, 0, 0]]), np.zeros(1), np.array([2]))
    assert o["terminated"][0] == 1.0


def test_kat5_acrobot(oracle):
    """acrobot.py:236-247 wrap(pi, -pi, pi) = -pi; done <=> -cos(t1) - cos(t1+t2) > 1."""
    pi = np.float32(np.pi)
    assert mn._acrobot_wrap(pi, -pi, pi) == -pi
    assert mn._acrobot_wrap(np.float32(0.3), -pi, pi) == np.float32(0.3)
    # hanging at rest stays at rest; reward -1, not done
    o = oracle.env_step("Acrobot-v1", np.zeros((1, 4)), np.zeros(1), np.array([1]))
    np.testing.assert_allclose(o["state"][0], 0.0, atol=1e-6)
    assert o["reward"][0] == -1.0 and o["terminated"][0] == 0.0
    np.testing.assert_allclose(o["obs"][0], [1, 0, 1, 0, 0, 0], atol=1e-6)
    # upright: -cos(pi - eps) - cos(pi - eps + 0) > 1 -> done, reward -0
    st = np.array([[3.0, 0.0, 0.0, 0.0]], np.float32)
    o = oracle.env_step("Acrobot-v1", st, np.zeros(1), np.array([1]))
    t1, t2 = o["state"][0, 0], o["state"][0, 1]
    done = (-np.cos(t1) - np.cos(t1 + t2)) > 1.0
    assert bool(o["terminated"][0]) == bool(done)
    assert o["reward"][0] == (0.0 if done else -1.0)
    # C and NumPy agree on random states
    rng = np.random.default_rng(0)
    for _ in range(50):
        s = rng.uniform(-2, 2, size=4).astype(np.float32)
        a = int(rng.integers(0, 3))
        oc = oracle.env_step("Acrobot-v1", s[None], np.zeros(1), np.array([a]))
        sn, _, obs, r, term, trunc = mn.acrobot_step(s, 0, a)
        np.testing.assert_allclose(oc["state"][0], sn, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(oc["obs"][0], obs, rtol=2e-5, atol=2e-5)
    # Acrobot-v2 (acrobot_no_term.py:151-157): the same step, reward included, with done = 0
    st = np.array([[3.0, 0.0, 0.0, 0.0]], np.float32)
    o1 = oracle.env_step("Acrobot-v1", st, np.zeros(1), np.array([1]))
    o2 = oracle.env_step("Acrobot-v2", st, np.zeros(1), np.array([1]))
    assert o1["terminated"][0] == 1.0 and o2["terminated"][0] == 0.0    # (the upright case above)
    assert o2["reward"][0] == o1["reward"][0] == 0.0                     # the reward keeps the angle test
    assert np.array_equal(o1["state"], o2["state"]) and np.array_equal(o1["obs"], o2["obs"])
    o2 = oracle.env_step("Acrobot-v2", st, np.array([4999]), np.array([1]))
    assert o2["truncated"][0] == 1.0 and o2["terminated"][0] == 0.0
    assert mn.acrobot_no_term_step(st[0], 0, 1)[4] == 0.0


# ---- KAT-6: qtransforms on an all-unvisited node (qtransforms.py:54-149)
def test_kat6_qtransforms_unvisited():
    t = mn.LaneTree(4, 3, 2)
    t.children_prior_logits[0] = np.array([0.3, -1.0, 2.0], np.float32)
    t.raw_values[0] = t.node_values[0] = np.array([0.7, 0.1], np.float32)
    assert (mn.qtransform_by_parent_and_siblings(t, 0) == 0).all()
    assert (mn.qtransform_completed_by_mix_value(t, 0) == 0).all()
    # visit scale (50 + max visits) * 0.1 once something differs
    t.children_visits[0] = np.array([3, 0, 0], np.int32)
    t.children_rewards[0, 0] = 1.0
    out = mn.qtransform_completed_by_mix_value(t, 0, use_mixed_value=False)
    # q0 = 1.0, others = raw mean 0.4 -> rescaled (1, 0, 0) * (50+3)*0.1
    np.testing.assert_allclose(out, [5.3, 0.0, 0.0], rtol=1e-6)


def test_kat6_gumbel_first_step_zero_q(oracle):
    """With all q equal the first root selection is argmax(gumbel + logits) (seq_halving.py:23-33)."""
    case = runners.synthetic_case("gumbel", B=16, S=1, A=5, K=1, seed=2)
    out = runners.run_policy_c(case)
    expect = np.argmax(case["root_gumbel"] + case["prior_logits"], -1)
    assert np.array_equal(out["action_from_parent"][:, 1], expect)


# ---- threefry (Salmon et al. 2011 known answers, 20 rounds)
def test_threefry_known_answers(oracle):
    key = np.array([[0, 0], [0xffffffff, 0xffffffff], [0x13198a2e, 0x03707344]], np.uint32)
    ctr = np.array([[0, 0], [0xffffffff, 0xffffffff], [0x243f6a88, 0x85a308d3]], np.uint32)
    out = oracle.threefry(key, ctr)
    expect = np.array([[0x6b200159, 0x99ba4efe], [0x1cb996fc, 0xbb002be7],
                       [0xc4923a9c, 0x483df7a0]], np.uint32)
    assert np.array_equal(out, expect)


# ---- numerics specification vs float64 libm
@pytest.mark.parametrize("fn,ref,lo,hi,tol", [
    ("exp", np.exp, -87.0, 88.0, 2e-7),
    ("log", np.log, 1e-30, 1e30, 2e-7),
    ("sin", np.sin, -20.0, 20.0, 2e-7),
    ("cos", np.cos, -20.0, 20.0, 2e-7),
    ("swish", lambda x: x / (1 + np.exp(-x)), -30.0, 30.0, 4e-7),
])
def test_numerics_spec_accuracy(oracle, fn, ref, lo, hi, tol):
    rng = np.random.default_rng(1)
    if fn == "log":
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), 200000)).astype(np.float32)
    else:
        x = rng.uniform(lo, hi, 200000).astype(np.float32)
    y = oracle.math_apply(fn, x).astype(np.float64)
    r = ref(x.astype(np.float64))
    if fn in ("sin", "cos"):
        err = np.abs(y - r)          # absolute: relative error is unbounded near zeros
    else:
        err = np.abs(y - r) / np.maximum(np.abs(r), 1e-30)
    assert err.max() < tol, (fn, err.max(), x[err.argmax()])


def test_numerics_spec_special_values(oracle):
    inf = np.float32(np.inf)
    e = oracle.math_apply("exp", np.array([-inf, -200, 0, 100, inf, -3.4e38], np.float32))
    assert list(e) == [0, 0, 1, inf, inf, 0]
    l = oracle.math_apply("log", np.array([0, 1, inf, 1.1754944e-38, 1e-45], np.float32))
    assert l[0] == -inf and l[1] == 0 and l[2] == inf
    small = np.array([1.1754944e-38, 1e-45], np.float32)  # smallest normal, smallest subnormal
    np.testing.assert_allclose(l[3:], np.log(small.astype(np.float64)), rtol=1e-6)
    assert np.isnan(oracle.math_apply("log", np.array([-1], np.float32))[0])
    # swish = x * logistic(x), the logistic's exponent clamped to [-64, 64] by specification
    sw = oracle.math_apply("swish", np.array([0, 100, -100, inf, -inf, 64, -64, np.nan], np.float32))
    assert sw[0] == 0 and sw[1] == 100 and sw[3] == inf and sw[4] == -inf and np.isnan(sw[7])
    tail = 1.0 / (1.0 + np.exp(64.0))                      # logistic at the clamp
    np.testing.assert_allclose(sw[2], -100 * tail, rtol=1e-6)
    np.testing.assert_allclose(sw[6], -64 * tail, rtol=1e-6)
    assert sw[5] == 64 and abs(sw[2]) < 2e-26              # the clamp moves results by < 2e-26
    # subnormal results of exp are rounded once
    x = np.linspace(-103.9, -87.4, 1000).astype(np.float32)
    y = oracle.math_apply("exp", x).astype(np.float64)
    assert np.all(np.abs(y - np.exp(x.astype(np.float64))) <= 1.5e-45 + 2e-7 * y)


def test_select_only_trig_equals_the_early_return_forms(oracle):
    """az_trig_(x, want_cos) -- one body for sine and cosine, selects only, the form the Acrobot
    step evaluates four at a time -- gives the bits of az_sinf / az_cosf on random bit patterns of
    every exponent, on the arguments an environment sees, and on the special values."""
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** 32, 1 << 21, dtype=np.uint64).astype(np.uint32)
    x = np.concatenate([bits.view(np.float32), rng.uniform(-40, 40, 1 << 20).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 8192.0, 8192.5, -8193.0,
                                  1073741824.0, 1073741760.0, -1073741824.0, 1e-45, 3.4e38],
                                 np.float32)])
    for sel, plain in (("sin_select", "sin"), ("cos_select", "cos")):
        a, b = oracle.math_apply(sel, x), oracle.math_apply(plain, x)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), sel


# ---------------------------------------------------------------- MinAtar Breakout (hand-derived)
def _breakout_state(**kw):
    from oracle import minatar_numpy as bk
    s = bk.EnvState()
    s.brick_map[1:4, :] = 1
    s.pos, s.ball_y, s.ball_x, s.ball_dir, s.last_y, s.last_x = 4, 3, 0, 2, 3, 0
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def test_breakout_known_answers():
    """gymnax minatar/breakout.py worked by hand: directions 0..3 are up-left, up-right,
    down-right, down-left (:183-194)."""
    from oracle import minatar_numpy as bk
    # free flight: down-right from (3, 0); the trail takes the old position; 'l' moves the paddle
    obs, s, r, done, trunc = bk.step_env(_breakout_state(), 1)
    assert (s.ball_y, s.ball_x, s.ball_dir, s.last_y, s.last_x, s.pos, s.time) == (4, 1, 2, 3, 0, 3, 1)
    assert r == 0 and done == 0 and trunc == 0 and not s.strike
    assert obs[9, 3, 0] == 1 and obs[4, 1, 1] == 1 and obs[3, 0, 2] == 1 and obs[:, :, 3].sum() == 30
    # a brick: up-right from (4, 5) meets brick (3, 6): +1, the brick goes, the ball stays on its
    # row, moves on in x and heads down-right (:230-244)
    _, s, r, done, _ = bk.step_env(_breakout_state(ball_y=4, ball_x=5, ball_dir=1), 0)
    assert r == 1 and s.brick_map[3, 6] == 0 and s.brick_map.sum() == 29
    assert (s.ball_y, s.ball_x, s.ball_dir, s.strike) == (4, 6, 2, True)
    # side wall: up-left from (5, 0) stays in column 0 and turns up-right (:196-204)
    _, s, _, _, _ = bk.step_env(_breakout_state(ball_y=5, ball_x=0, ball_dir=0), 0)
    assert (s.ball_y, s.ball_x, s.ball_dir) == (4, 0, 1)
    # ceiling: up-right from (0, 4) over a cleared row 0 stays in row 0 and turns down-right (:223-228)
    _, s, _, _, _ = bk.step_env(_breakout_state(ball_y=0, ball_x=4, ball_dir=1), 0)
    assert (s.ball_y, s.ball_x, s.ball_dir) == (0, 5, 2)
    # paddle under the ball's NEW column: down-right from (8, 3) with the paddle at 4 bounces back
    # the way it came (:264-270: [2, 3, 0, 1])
    _, s, _, done, _ = bk.step_env(_breakout_state(ball_y=8, ball_x=3, ball_dir=2, pos=4), 0)
    assert (s.ball_y, s.ball_x, s.ball_dir, done) == (8, 4, 0, 0)
    # paddle under the ball's OLD column: straight mirror (:257-262: [3, 2, 1, 0])
    _, s, _, done, _ = bk.step_env(_breakout_state(ball_y=8, ball_x=3, ball_dir=2, pos=3), 0)
    assert (s.ball_y, s.ball_x, s.ball_dir, done) == (8, 4, 1, 0)
    # a miss ends the episode; the ball sits in the bottom row (:271-272)
    _, s, _, done, _ = bk.step_env(_breakout_state(ball_y=8, ball_x=3, ball_dir=2, pos=0), 0)
    assert done == 1 and s.terminal and (s.ball_y, s.ball_x) == (9, 4)
    # the last brick gone: new rows appear when the ball next reaches the bottom row (:249-255)
    empty = _breakout_state(ball_y=8, ball_x=3, ball_dir=2, pos=4)
    empty.brick_map[:] = 0
    _, s, _, _, _ = bk.step_env(empty, 0)
    assert s.brick_map.sum() == 30
    # time limit: truncation only when the step did not also end the game (:121-128)
    _, s, _, done, trunc = bk.step_env(_breakout_state(time=999), 0)
    assert done == 0 and trunc == 1
    # the paddle stops at the walls (:171-178)
    _, s, _, _, _ = bk.step_env(_breakout_state(pos=9), 2)
    assert s.pos == 9
    # packed form round trip
    w = s.words()
    assert bk.EnvState.from_words(w).words().tolist() == w.tolist() and w[11] == 0x3ff and w[10] == 0


def test_breakout_reset_and_wrappers():
    from oracle import minatar_numpy as bk
    starts = set()
    for seed in range(16):
        s = bk.reset_env(np.array([0, seed], np.uint32))
        assert (s.ball_y, s.pos, s.last_y, s.time, s.brick_map.sum()) == (3, 4, 3, 0, 30)
        assert (s.ball_x, s.ball_dir) in ((0, 2), (9, 3)) and s.last_x == s.ball_x
        starts.add(s.ball_x)
    assert starts == {0, 9}
    # EpisodeWrapper / AutoResetWrapper: an episode of 5 steps is cut and reset; the observation
    # before the reset is kept in real_obs
    w = bk.Wrapped((3, 4), episode_length=5)
    for t in range(5):
        w.step(0)
    assert w.done == 1 and w.truncation == 1 and w.steps == 5 and w.state.time == 0
    assert w.real_obs[:, :, 1].sum() == 1 and not np.array_equal(w.real_obs, w.obs)
    w.step(0)
    assert w.steps == 1 and w.done == 0
