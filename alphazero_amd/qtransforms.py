"""Q-value transforms (mirrors mctx_dist/_src/qtransforms.py).

In the reference a qtransform is a Python function traced into the search; here
the three transforms are evaluated inside the HIP search kernels, and these
objects only carry their keyword options to the kernels.  As in the reference
they are configured with ``functools.partial(qtransform_..., option=value)``.
A user-defined qtransform cannot be compiled into the kernels and is rejected.
"""
import functools

from . import _native as nat


def qtransform_by_min_max(tree=None, node_index=None, *, min_value, max_value):
    """mctx_dist/_src/qtransforms.py:24-51 (evaluated in-kernel)."""
    raise NotImplementedError("qtransforms are evaluated inside the search kernels; "
                              "pass this function (or a functools.partial of it) to a policy")


def qtransform_by_parent_and_siblings(tree=None, node_index=None, *, epsilon=1e-8):
    """mctx_dist/_src/qtransforms.py:54-87 (evaluated in-kernel)."""
    raise NotImplementedError("qtransforms are evaluated inside the search kernels; "
                              "pass this function (or a functools.partial of it) to a policy")


def qtransform_completed_by_mix_value(tree=None, node_index=None, *, value_scale=0.1,
                                      maxvisit_init=50.0, rescale_values=True,
                                      use_mixed_value=True, epsilon=1e-8):
    """mctx_dist/_src/qtransforms.py:90-149 (evaluated in-kernel)."""
    raise NotImplementedError("qtransforms are evaluated inside the search kernels; "
                              "pass this function (or a functools.partial of it) to a policy")


_KNOWN = {
    qtransform_by_parent_and_siblings: (nat.QT_PARENT_SIBLINGS, dict(epsilon=1e-8)),
    qtransform_completed_by_mix_value: (nat.QT_COMPLETED_MIX, dict(
        value_scale=0.1, maxvisit_init=50.0, rescale_values=True, use_mixed_value=True,
        epsilon=1e-8)),
    qtransform_by_min_max: (nat.QT_MIN_MAX, dict(min_value=None, max_value=None)),
}


def unwrap_partial(fn):
    """(base function, merged keywords) of a possibly nested functools.partial."""
    kwargs = {}
    while isinstance(fn, functools.partial):
        if fn.args:
            raise NotImplementedError("positional arguments bound with functools.partial are not "
                                      "supported; bind options by keyword")
        merged = dict(fn.keywords)
        merged.update(kwargs)
        kwargs = merged
        fn = fn.func
    return fn, kwargs


def apply_to_desc(qtransform, desc):
    """Writes the transform's kind and options into a native SearchDesc."""
    fn, kwargs = unwrap_partial(qtransform)
    if fn not in _KNOWN:
        raise NotImplementedError(
            "only qtransform_by_parent_and_siblings, qtransform_completed_by_mix_value and "
            "qtransform_by_min_max (optionally wrapped in functools.partial) can run inside the "
            f"search kernels; got {qtransform!r}")
    kind, defaults = _KNOWN[fn]
    unknown = set(kwargs) - set(defaults)
    if unknown:
        raise TypeError(f"{fn.__name__} got unexpected options {sorted(unknown)}")
    opts = dict(defaults)
    opts.update(kwargs)
    desc.qtransform = kind
    desc.q_epsilon = float(opts.get("epsilon", 1e-8))
    desc.value_scale = float(opts.get("value_scale", 0.1))
    desc.maxvisit_init = float(opts.get("maxvisit_init", 50.0))
    desc.rescale_values = int(bool(opts.get("rescale_values", True)))
    desc.use_mixed_value = int(bool(opts.get("use_mixed_value", True)))
    if kind == nat.QT_MIN_MAX:
        if opts["min_value"] is None or opts["max_value"] is None:
            raise TypeError("qtransform_by_min_max needs min_value and max_value")
        desc.min_value = float(opts["min_value"])
        desc.max_value = float(opts["max_value"])
    else:
        desc.min_value, desc.max_value = 0.0, 1.0
    return desc
