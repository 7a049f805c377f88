"""Action selection functions (mirrors mctx_dist/_src/action_selection.py).

The four selection rules run inside the HIP search kernels.  As in the
reference, `search()` receives them as ``functools.partial`` objects binding
their keyword options; here the functions are markers that identify the rule
and carry those options to the kernels.
"""
import dataclasses
from typing import Any

from . import _native as nat
from . import qtransforms as qt

_MSG = ("action selection runs inside the search kernels; pass this function (or a "
        "functools.partial of it) to search()")


def muzero_action_selection(rng_key=None, tree=None, node_index=None, depth=None, *,
                            pb_c_init=1.25, pb_c_base=19652.0,
                            qtransform=qt.qtransform_by_parent_and_siblings):
    """mctx_dist/_src/action_selection.py:48-90 (PUCT on softmax(prior logits))."""
    raise NotImplementedError(_MSG)


def sampled_muzero_action_selection(rng_key=None, tree=None, node_index=None, depth=None, *,
                                    pb_c_init=1.25, pb_c_base=19652.0,
                                    qtransform=qt.qtransform_by_parent_and_siblings):
    """mctx_dist/_src/action_selection.py:93-135 (PUCT on stored sample-based probabilities)."""
    raise NotImplementedError(_MSG)


def gumbel_muzero_root_action_selection(rng_key=None, tree=None, node_index=None, *,
                                        num_simulations=None, max_num_considered_actions=None,
                                        qtransform=qt.qtransform_completed_by_mix_value):
    """mctx_dist/_src/action_selection.py:148-200 (Sequential Halving with Gumbel)."""
    raise NotImplementedError(_MSG)


def gumbel_muzero_interior_action_selection(rng_key=None, tree=None, node_index=None, depth=None,
                                            *, qtransform=qt.qtransform_completed_by_mix_value):
    """mctx_dist/_src/action_selection.py:203-240 (deterministic interior selection)."""
    raise NotImplementedError(_MSG)


def stochastic_muzero_action_selection(rng_key=None, tree=None, node_index=None, depth=None, *,
                                       decision_node_selection_fn=None, num_actions=None):
    """mctx_dist/_src/policies.py:676-708 (_make_stochastic_action_selection_fn): decision nodes
    select with `decision_node_selection_fn` (a muzero_action_selection partial) over the first
    `num_actions` columns, chance nodes with argmax(softmax(chance logits) / (visits + 1))."""
    raise NotImplementedError(_MSG)


@dataclasses.dataclass(frozen=True)
class GumbelMuZeroExtraData:
    """mctx_dist/_src/action_selection.py:138-141"""
    root_gumbel: Any


def resolve(root_fn, interior_fn, desc, num_simulations):
    """Fills the native SearchDesc from the (root, interior) selection functions."""
    rf, rkw = qt.unwrap_partial(root_fn)
    inf, ikw = qt.unwrap_partial(interior_fn)
    rkw = dict(rkw)
    rkw.pop("depth", None)  # policies.py:98-100 binds depth=0 into the root function
    if rf is gumbel_muzero_root_action_selection and inf is gumbel_muzero_interior_action_selection:
        desc.variant = nat.VARIANT_GUMBEL
        m = rkw.get("max_num_considered_actions")
        if m is None:
            raise TypeError("gumbel_muzero_root_action_selection needs max_num_considered_actions")
        ns = rkw.get("num_simulations", num_simulations)
        if ns != num_simulations:
            raise ValueError("num_simulations bound into the root selection differs from search()'s")
        desc.max_num_considered = int(m)
        q_root = rkw.get("qtransform", qt.qtransform_completed_by_mix_value)
        q_int = ikw.get("qtransform", qt.qtransform_completed_by_mix_value)
    elif rf is inf and rf is stochastic_muzero_action_selection:
        dec, dkw = qt.unwrap_partial(ikw.get("decision_node_selection_fn"))
        if dec is not muzero_action_selection or rkw != ikw:
            raise NotImplementedError("stochastic selection: decision nodes select with a "
                                      "muzero_action_selection partial, the same at the root")
        desc.variant = nat.VARIANT_STOCHASTIC
        desc.num_decision_actions = int(ikw["num_actions"])
        desc.pb_c_init = float(dkw.get("pb_c_init", 1.25))
        desc.pb_c_base = float(dkw.get("pb_c_base", 19652.0))
        q_root = q_int = dkw.get("qtransform", qt.qtransform_by_parent_and_siblings)
    elif rf is inf and rf in (muzero_action_selection, sampled_muzero_action_selection):
        desc.variant = (nat.VARIANT_MUZERO if rf is muzero_action_selection
                        else nat.VARIANT_SAMPLED)
        for k in ("pb_c_init", "pb_c_base"):
            if rkw.get(k, None) != ikw.get(k, None):
                raise NotImplementedError("root and interior selection must share " + k)
        desc.pb_c_init = float(ikw.get("pb_c_init", 1.25))
        desc.pb_c_base = float(ikw.get("pb_c_base", 19652.0))
        q_root = rkw.get("qtransform", qt.qtransform_by_parent_and_siblings)
        q_int = ikw.get("qtransform", qt.qtransform_by_parent_and_siblings)
    else:
        raise NotImplementedError(
            "the search kernels implement the (muzero, muzero), (sampled, sampled) and "
            "(gumbel root, gumbel interior) selection pairs of the reference; got "
            f"{root_fn!r} / {interior_fn!r}")
    a, b = nat.SearchDesc(), nat.SearchDesc()
    qt.apply_to_desc(q_root, a)
    qt.apply_to_desc(q_int, b)
    if bytes(a) != bytes(b):
        raise NotImplementedError("root and interior selection must use the same qtransform")
    qt.apply_to_desc(q_int, desc)
    return desc
