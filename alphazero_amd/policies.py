"""Search policies (mirrors mctx_dist/_src/policies.py) on HIP kernels.

`muzero_policy`, `sampled_muzero_policy` and `gumbel_muzero_policy` keep the
reference signatures.  Keyword-only extras without a counterpart in the
reference inject the random quantities as tensors, so that a run can be
reproduced exactly against another implementation:

  dirichlet_noise  [B,A]   the Dirichlet sample mixed into the root prior
  sample_counts    [B,A]   per-action counts of the `num_samples` categorical draws
  tiebreak_noise   [B,S,D,A] or "zero": uniform draws behind the 1e-7 tie-break noise
  action_gumbel    [B,A]   Gumbel noise of the final categorical action draw
  root_gumbel      [B,A]   Gumbel noise of gumbel_muzero_policy (before scaling)
  stats            dict    receives `sum_depth` (see search.search)

Without them the noise comes from the device threefry generator keyed by
`rng_key` with the reference's split structure.
"""
import ctypes as C
import functools

import torch

from . import _native as nat
from . import action_selection
from . import base
from . import prng
from . import qtransforms
from . import search as search_lib


def _f32(x):
    return x.detach().to(torch.float32).contiguous()


def _key_words(key):
    return (C.c_uint32 * 2)(*key)


def _check_floats(**kw):
    for name, v in kw.items():
        if not isinstance(v, float):
            raise AssertionError(f"{name} must be a float")  # policies.py:508


def _root_muzero_logits(root, invalid_actions, dirichlet_noise, dirichlet_key,
                        dirichlet_fraction, dirichlet_alpha):
    """policies.py:82-90: Dirichlet-mixed, log'ed and masked root logits."""
    lib = nat.lib()
    logits = _f32(root.prior_logits)
    B, A = logits.shape
    stream = nat.stream_ptr()
    if dirichlet_noise is None:
        dirichlet_noise = torch.empty_like(logits)
        nat.check(lib.az_dirichlet(B, A, float(dirichlet_alpha), _key_words(dirichlet_key),
                                   nat.ptr(dirichlet_noise), stream))
    else:
        dirichlet_noise = _f32(dirichlet_noise)
    inv = None if invalid_actions is None else _f32(invalid_actions)
    return logits, dirichlet_noise, inv


def muzero_policy(params, rng_key, root, recurrent_fn, num_simulations, invalid_actions=None,
                  max_depth=None, loop_fn=None, *,
                  qtransform=qtransforms.qtransform_by_parent_and_siblings,
                  dirichlet_fraction=0.25, dirichlet_alpha=0.3, pb_c_init=1.25, pb_c_base=19652,
                  temperature=1.0, dirichlet_noise=None, tiebreak_noise=None, action_gumbel=None,
                  stats=None):
    """mctx_dist/_src/policies.py:30-122."""
    _check_floats(dirichlet_alpha=dirichlet_alpha, dirichlet_fraction=dirichlet_fraction)
    lib = nat.lib()
    rng_key, dirichlet_rng_key, search_rng_key = prng.split(rng_key, 3)
    logits, noise, inv = _root_muzero_logits(root, invalid_actions, dirichlet_noise,
                                             dirichlet_rng_key, dirichlet_fraction, dirichlet_alpha)
    B, A = logits.shape
    noisy_logits = torch.empty_like(logits)
    nat.check(lib.az_root_muzero(B, A, nat.ptr(logits), nat.ptr(noise), float(dirichlet_fraction),
                                 nat.ptr(inv), nat.ptr(noisy_logits), nat.stream_ptr()))
    root = root.replace(prior_logits=noisy_logits)

    interior = functools.partial(action_selection.muzero_action_selection,
                                 pb_c_base=pb_c_base, pb_c_init=pb_c_init, qtransform=qtransform)
    root_fn = functools.partial(interior, depth=0)
    search_tree = search_lib.search(
        params=params, rng_key=search_rng_key, root=root, recurrent_fn=recurrent_fn,
        root_action_selection_fn=root_fn, interior_action_selection_fn=interior,
        num_simulations=num_simulations, max_depth=max_depth, invalid_actions=invalid_actions,
        loop_fn=loop_fn, tiebreak_noise=tiebreak_noise, stats=stats)
    return _visit_count_output(search_tree, rng_key, temperature, action_gumbel)


def sampled_muzero_policy(params, rng_key, root, recurrent_fn, num_simulations, num_samples=100,
                          invalid_actions=None, max_depth=None, loop_fn=None, *,
                          qtransform=qtransforms.qtransform_by_parent_and_siblings,
                          dirichlet_fraction=0.25, dirichlet_alpha=0.3, pb_c_init=1.25,
                          pb_c_base=19652, temperature=1.0, sampling_dist_temperature=1.0,
                          dirichlet_noise=None, sample_counts=None, tiebreak_noise=None,
                          action_gumbel=None, stats=None):
    """mctx_dist/_src/policies.py:125-234.

    As in the reference, only the root holds sample-based probabilities: the wrapper
    built by `_make_sampled_recurrent_fn` (policies.py:548-566) discards its empirical
    probabilities, so interior nodes store the recurrent_fn's raw logits and
    sampled_muzero_action_selection reads them as probabilities."""
    _check_floats(dirichlet_alpha=dirichlet_alpha, dirichlet_fraction=dirichlet_fraction)
    lib = nat.lib()
    rng_key, sample_dist_key, dirichlet_rng_key, search_rng_key = prng.split(rng_key, 4)
    logits, noise, inv = _root_muzero_logits(root, invalid_actions, dirichlet_noise,
                                             dirichlet_rng_key, dirichlet_fraction, dirichlet_alpha)
    B, A = logits.shape
    counts = None if sample_counts is None else _f32(sample_counts)
    probs = torch.empty_like(logits)
    tau = float(sampling_dist_temperature)
    exponent = 1 - 1 / tau  # policies.py:546, evaluated in double like the Python expression
    nat.check(lib.az_root_sampled(B, A, nat.ptr(logits), nat.ptr(noise), float(dirichlet_fraction),
                                  nat.ptr(inv), int(num_samples), tau, exponent, nat.ptr(counts),
                                  _key_words(sample_dist_key), nat.ptr(probs), nat.stream_ptr()))
    root = root.replace(prior_logits=probs)

    interior = functools.partial(action_selection.sampled_muzero_action_selection,
                                 pb_c_base=pb_c_base, pb_c_init=pb_c_init, qtransform=qtransform)
    root_fn = functools.partial(interior, depth=0)
    search_tree = search_lib.search(
        params=params, rng_key=search_rng_key, root=root, recurrent_fn=recurrent_fn,
        root_action_selection_fn=root_fn, interior_action_selection_fn=interior,
        num_simulations=num_simulations, max_depth=max_depth, invalid_actions=invalid_actions,
        loop_fn=loop_fn, tiebreak_noise=tiebreak_noise, stats=stats)
    return _visit_count_output(search_tree, rng_key, temperature, action_gumbel)


def _make_stochastic_recurrent_fn(decision_node_fn, chance_node_fn, num_actions, num_chance_outcomes):
    """mctx_dist/_src/policies.py:569-639: one recurrent_fn over A + C "actions".  Both node
    functions are evaluated for every lane and the parent's kind picks the result; the halves
    of the logits that belong to the other kind of node are -inf and never read."""

    def stochastic_recurrent_fn(params, rng, action_or_chance, state):
        B = action_or_chance.shape[0]
        dev = action_or_chance.device
        action = action_or_chance - 0
        chance_outcome = action_or_chance - num_actions
        decision_output, afterstate_embedding = decision_node_fn(params, rng, action, state.state_embedding)
        av = decision_output.afterstate_value
        neg_a = torch.full((B, num_actions), float("-inf"), dtype=torch.float32, device=dev)
        neg_c = torch.full((B, num_chance_outcomes), float("-inf"), dtype=torch.float32, device=dev)
        if_decision = base.RecurrentFnOutput(
            prior_logits=torch.cat([neg_a, _f32(decision_output.chance_logits)], dim=-1),
            value=av, reward=torch.zeros_like(av), discount=torch.ones_like(av))
        chance_output, state_embedding = chance_node_fn(params, rng, chance_outcome,
                                                        state.afterstate_embedding)
        if_chance = base.RecurrentFnOutput(
            prior_logits=torch.cat([_f32(chance_output.action_logits), neg_c], dim=-1),
            value=chance_output.value, reward=chance_output.reward, discount=chance_output.discount)
        new_state = base.StochasticRecurrentState(
            state_embedding=state_embedding, afterstate_embedding=afterstate_embedding,
            is_decision_node=torch.logical_not(state.is_decision_node))

        def pick(decision_leaf, chance_leaf):
            is_decision = state.is_decision_node.reshape([-1] + [1] * (decision_leaf.dim() - 1))
            return torch.where(is_decision, decision_leaf, chance_leaf)
        output = base.RecurrentFnOutput(
            reward=pick(if_decision.reward, if_chance.reward),
            discount=pick(if_decision.discount, if_chance.discount),
            prior_logits=pick(if_decision.prior_logits, if_chance.prior_logits),
            value=pick(if_decision.value, if_chance.value))
        return output, new_state

    return stochastic_recurrent_fn


def _mask_tree(tree, num_actions, mode):
    """mctx_dist/_src/policies.py:642-673: the columns of decision nodes ('decision') or of
    chance nodes ('chance') of every [.., A + C] array, as contiguous copies."""
    if mode == "decision":
        take = lambda x: x[:, :, :num_actions].contiguous()
        take_root = lambda x: x[:, :num_actions].contiguous()
    elif mode == "chance":
        take = lambda x: x[:, :, num_actions:].contiguous()
        take_root = lambda x: x[:, num_actions:].contiguous()
    else:
        raise ValueError(f"Unknown mode: {mode}.")
    return tree.replace(
        children_index=take(tree.children_index),
        children_prior_logits=take(tree.children_prior_logits),
        children_visits=take(tree.children_visits),
        children_rewards=take(tree.children_rewards),
        children_discounts=take(tree.children_discounts),
        children_values=take(tree.children_values),
        root_invalid_actions=take_root(tree.root_invalid_actions))


def stochastic_muzero_policy(params, rng_key, root, decision_recurrent_fn, chance_recurrent_fn,
                             num_simulations, invalid_actions=None, max_depth=None, loop_fn=None, *,
                             qtransform=qtransforms.qtransform_by_parent_and_siblings,
                             dirichlet_fraction=0.25, dirichlet_alpha=0.3, pb_c_init=1.25,
                             pb_c_base=19652, temperature=1.0, dirichlet_noise=None,
                             tiebreak_noise=None, action_gumbel=None, stats=None):
    """mctx_dist/_src/policies.py:346-483.  The two node functions are arbitrary callables, so
    the search runs on the per-simulation kernels (az_simulate / az_expand_backward) around them."""
    _check_floats(dirichlet_alpha=dirichlet_alpha, dirichlet_fraction=dirichlet_fraction)
    lib = nat.lib()
    rng_key, dirichlet_rng_key, search_rng_key = prng.split(rng_key, 3)
    logits, noise, inv = _root_muzero_logits(root, invalid_actions, dirichlet_noise,
                                             dirichlet_rng_key, dirichlet_fraction, dirichlet_alpha)
    B, A = logits.shape
    noisy_logits = torch.empty_like(logits)
    nat.check(lib.az_root_muzero(B, A, nat.ptr(logits), nat.ptr(noise), float(dirichlet_fraction),
                                 nat.ptr(inv), nat.ptr(noisy_logits), nat.stream_ptr()))
    # a dummy afterstate embedding for the root (policies.py:419-424)
    dummy_action = torch.zeros((B,), dtype=torch.int32, device=logits.device)
    dummy_output, dummy_afterstate_embedding = decision_recurrent_fn(params, rng_key, dummy_action,
                                                                     root.embedding)
    C_ = dummy_output.chance_logits.shape[-1]
    root = root.replace(
        prior_logits=torch.cat([noisy_logits, torch.full((B, C_), float("-inf"), dtype=torch.float32,
                                                         device=logits.device)], dim=-1),
        embedding=base.StochasticRecurrentState(
            state_embedding=root.embedding, afterstate_embedding=dummy_afterstate_embedding,
            is_decision_node=torch.ones((B,), dtype=torch.bool, device=logits.device)))
    recurrent_fn = _make_stochastic_recurrent_fn(decision_recurrent_fn, chance_recurrent_fn, A, C_)
    decision_fn = functools.partial(action_selection.muzero_action_selection, pb_c_base=pb_c_base,
                                    pb_c_init=pb_c_init, qtransform=qtransform)
    interior = functools.partial(action_selection.stochastic_muzero_action_selection,
                                 decision_node_selection_fn=decision_fn, num_actions=A)
    root_fn = functools.partial(interior, depth=0)
    search_tree = search_lib.search(
        params=params, rng_key=search_rng_key, root=root, recurrent_fn=recurrent_fn,
        root_action_selection_fn=root_fn, interior_action_selection_fn=interior,
        num_simulations=num_simulations, max_depth=max_depth, invalid_actions=invalid_actions,
        loop_fn=loop_fn, tiebreak_noise=tiebreak_noise, stats=stats)
    search_tree = _mask_tree(search_tree, A, "decision")
    return _visit_count_output(search_tree, rng_key, temperature, action_gumbel)


def _visit_count_output(search_tree, rng_key, temperature, action_gumbel):
    """policies.py:113-122 / 225-234: weights = root visit probabilities, action =
    categorical(log weights / temperature)."""
    d, tp = search_tree._desc_and_ptrs()
    dev = search_tree.node_visits.device
    action = torch.empty((d.B,), dtype=torch.int32, device=dev)
    weights = torch.empty((d.B, d.A), dtype=torch.float32, device=dev)
    g = None if action_gumbel is None else _f32(action_gumbel)
    nat.check(nat.lib().az_epilogue_visits(d, tp, float(temperature), nat.ptr(g),
                                           _key_words(prng.as_key(rng_key)), nat.ptr(action),
                                           nat.ptr(weights), nat.stream_ptr()))
    return base.PolicyOutput(action=action, action_weights=weights, search_tree=search_tree)


def gumbel_muzero_policy(params, rng_key, root, recurrent_fn, num_simulations,
                         invalid_actions=None, max_depth=None, loop_fn=None, *,
                         qtransform=qtransforms.qtransform_completed_by_mix_value,
                         max_num_considered_actions=16, gumbel_scale=1., root_gumbel=None,
                         stats=None):
    """mctx_dist/_src/policies.py:237-343."""
    lib = nat.lib()
    logits = _f32(root.prior_logits)
    B, A = logits.shape
    inv = None if invalid_actions is None else _f32(invalid_actions)
    rng_key, gumbel_rng = prng.split(rng_key, 2)
    masked = torch.empty_like(logits)
    gumbel = torch.empty_like(logits)
    g_in = None if root_gumbel is None else _f32(root_gumbel)
    nat.check(lib.az_root_gumbel(B, A, nat.ptr(logits), nat.ptr(inv), nat.ptr(g_in),
                                 _key_words(gumbel_rng), float(gumbel_scale), nat.ptr(masked),
                                 nat.ptr(gumbel), nat.stream_ptr()))
    root = root.replace(prior_logits=masked)

    extra_data = action_selection.GumbelMuZeroExtraData(root_gumbel=gumbel)
    search_tree = search_lib.search(
        params=params, rng_key=rng_key, root=root, recurrent_fn=recurrent_fn,
        root_action_selection_fn=functools.partial(
            action_selection.gumbel_muzero_root_action_selection,
            num_simulations=num_simulations,
            max_num_considered_actions=max_num_considered_actions, qtransform=qtransform),
        interior_action_selection_fn=functools.partial(
            action_selection.gumbel_muzero_interior_action_selection, qtransform=qtransform),
        num_simulations=num_simulations, max_depth=max_depth, invalid_actions=invalid_actions,
        extra_data=extra_data, loop_fn=loop_fn, stats=stats)

    d, tp = search_tree._desc_and_ptrs()
    qtransforms.apply_to_desc(qtransform, d)
    action = torch.empty((B,), dtype=torch.int32, device=logits.device)
    weights = torch.empty((B, A), dtype=torch.float32, device=logits.device)
    nat.check(lib.az_epilogue_gumbel(d, tp, nat.ptr(masked), nat.ptr(gumbel), nat.ptr(inv),
                                     nat.ptr(action), nat.ptr(weights), nat.stream_ptr()))
    return base.PolicyOutput(action=action, action_weights=weights, search_tree=search_tree)


def actor_post(search_tree, action_weights, sampled_action, prior_logits, deterministic=False):
    """alphazero.py:367-379: (actions, search_value, prior_log_prob) after a policy call."""
    d, tp = search_tree._desc_and_ptrs()
    dev = search_tree.node_visits.device
    action = torch.empty((d.B,), dtype=torch.int32, device=dev)
    sv = torch.empty((d.B, d.K), dtype=torch.float32, device=dev)
    lp = torch.empty((d.B,), dtype=torch.float32, device=dev)
    nat.check(nat.lib().az_actor_post(
        d, tp, nat.ptr(_f32(action_weights)), nat.ptr(sampled_action.to(torch.int32).contiguous()),
        int(bool(deterministic)), nat.ptr(_f32(prior_logits)), nat.ptr(action), nat.ptr(sv),
        nat.ptr(lp), nat.stream_ptr()))
    return action, sv, lp
