"""Helpers that drive the HIP path with the same *case* dicts as oracle/runners.py."""
import functools

import numpy as np
import torch

import alphazero_amd as mctx
from oracle import runners

INT_FIELDS = ("node_visits", "parents", "action_from_parent", "children_index", "children_visits")
FLT_FIELDS = ("raw_values", "node_values", "children_prior_logits", "children_rewards",
              "children_discounts", "children_values")


def dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x)).cuda()
    return t if dtype is None else t.to(dtype)


def qtransform_of(case):
    name, kw = runners._qt_defaults(case)
    fn = {"parent_and_siblings": mctx.qtransform_by_parent_and_siblings,
          "completed_by_mix_value": mctx.qtransform_completed_by_mix_value,
          "min_max": mctx.qtransform_by_min_max}[name]
    return functools.partial(fn, **kw) if kw else fn


def make_model(model):
    pol = [(dev(w), dev(b)) for w, b in model["policy_layers"]]
    val = [(dev(w), dev(b)) for w, b in model["value_layers"]]
    return mctx.FusedModel(model["env_id"], pol, val, dev(model["mean"]), dev(model["std"]),
                           gamma=model.get("gamma", 0.99))


def run_policy_hip(case, path="persistent", literal=False):
    """path: 'persistent' (one launch: injected or fused kernel) or 'callback'
    (per-simulation kernels around a Python recurrent_fn).  literal: the one-thread-per-lane
    form of the persistent kernels instead of the cooperative one."""
    v = case["variant"]
    B, S, A, K = case["B"], case["S"], case["A"], case["K"]
    model = case.get("model")
    if model is not None:
        fm = make_model(model)
        frf = mctx.FusedRecurrentFn(fm)
        recurrent_fn = frf if path == "persistent" else (lambda p, k, a, e: frf(p, k, a, e))
        embedding = {"state": dev(case["root_state"]), "time": dev(case["root_time"])}
    else:
        ev = {k: dev(x) for k, x in case["evals"].items()}
        if path == "persistent":
            recurrent_fn = mctx.InjectedRecurrentFn(ev["prior_logits"], ev["value"], ev["reward"],
                                                    ev["discount"])
        else:
            counter = {"sim": 0}

            def recurrent_fn(params, rng_key, action, embedding):
                s = counter["sim"]
                counter["sim"] += 1
                out = mctx.RecurrentFnOutput(reward=ev["reward"][:, s], discount=ev["discount"][:, s],
                                             prior_logits=ev["prior_logits"][:, s],
                                             value=ev["value"][:, s])
                return out, embedding + 1
        embedding = torch.zeros((B,), dtype=torch.int32, device="cuda")
    root = mctx.RootFnOutput(prior_logits=dev(case["prior_logits"]), value=dev(case["value"]),
                             embedding=embedding)
    inv = None if case.get("invalid") is None else dev(case["invalid"])
    stats = {"kernel_form": "literal"} if literal else {}
    common = dict(params=(), rng_key=0, root=root, recurrent_fn=recurrent_fn, num_simulations=S,
                  invalid_actions=inv, max_depth=case.get("max_depth"),
                  qtransform=qtransform_of(case), stats=stats)
    if v == "gumbel":
        out = mctx.gumbel_muzero_policy(
            max_num_considered_actions=case.get("max_num_considered", 16), gumbel_scale=1.0,
            root_gumbel=dev(case["root_gumbel"]), **common)
    else:
        kw = dict(dirichlet_fraction=case.get("dirichlet_fraction", 0.25), dirichlet_alpha=0.3,
                  pb_c_init=case.get("pb_c_init", 1.25), pb_c_base=case.get("pb_c_base", 19652),
                  temperature=case.get("temperature", 1.0), dirichlet_noise=dev(case["dirichlet"]),
                  tiebreak_noise=dev(case["tie_noise"]), action_gumbel=dev(case["action_gumbel"]))
        if v == "muzero":
            out = mctx.muzero_policy(**common, **kw)
        else:
            out = mctx.sampled_muzero_policy(num_samples=case["num_samples"],
                                             sample_counts=dev(case["counts"]),
                                             sampling_dist_temperature=case.get("sampling_temperature", 1.0),
                                             **common, **kw)
    torch.cuda.synchronize()
    t = out.search_tree
    res = {f: getattr(t, f).cpu().numpy() for f in INT_FIELDS + FLT_FIELDS}
    res["action"] = out.action.cpu().numpy()
    res["action_weights"] = out.action_weights.cpu().numpy()
    res["sum_depth"] = int(stats["sum_depth"].item())
    if model is not None:
        res["emb_state"] = t.embeddings["state"].cpu().numpy()
        res["emb_time"] = t.embeddings["time"].cpu().numpy()
    res["_tree"] = t
    res["_out"] = out
    return res


def assert_same(hip, ref, exact_floats=True, rtol=1e-5, atol=1e-6):
    """Integer arrays bit-exact; float arrays bit-exact (or within tolerance)."""
    for f in INT_FIELDS + ("action",):
        assert np.array_equal(hip[f], ref[f]), f"{f} differs"
    for f in FLT_FIELDS + ("action_weights",):
        a, b = hip[f], ref[f].reshape(hip[f].shape)
        if exact_floats:
            assert np.array_equal(a, b), f"{f} differs: max abs {np.abs(a - b).max()}"
        else:
            np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=f)
    assert hip["sum_depth"] == ref["sum_depth"]
