"""Sequential Halving helpers (mirrors mctx_dist/_src/seq_halving.py).

The considered-visits table is a host-side constant built once per
(max_num_considered_actions, num_simulations) by the native library and cached
on the device; `score_considered` itself runs inside the search kernels.
"""
import ctypes as C

import numpy as np

from . import _native as nat

_TABLE_CACHE = {}


def get_table_of_considered_visits(max_num_considered_actions, num_simulations):
    """mctx_dist/_src/seq_halving.py:73-88 -> tuple of tuples [(m + 1), num_simulations]."""
    tab = _host_table(max_num_considered_actions, num_simulations)
    return tuple(tuple(int(v) for v in row) for row in tab)


def get_sequence_of_considered_visits(max_num_considered_actions, num_simulations):
    """mctx_dist/_src/seq_halving.py:36-70"""
    m = max(int(max_num_considered_actions), 0)
    return tuple(int(v) for v in _host_table(m, num_simulations)[m])


def _host_table(m, S):
    tab = np.zeros((m + 1, S), np.int32)
    nat.check(nat.lib().az_considered_table(m, S, tab.ctypes.data_as(C.c_void_p)))
    return tab


def device_table(m, S, device):
    """The table as an int32 tensor on `device` (cached)."""
    import torch
    key = (int(m), int(S), str(device))
    if key not in _TABLE_CACHE:
        _TABLE_CACHE[key] = torch.from_numpy(_host_table(int(m), int(S))).to(device)
    return _TABLE_CACHE[key]
