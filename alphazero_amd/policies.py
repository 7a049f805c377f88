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
