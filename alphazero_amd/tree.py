"""The search tree returned by the policies (mirrors mctx_dist/_src/tree.py).

Every array is a torch tensor in HBM with the reference's shape, dtype and
row-major layout; the search kernels write these arrays directly, so the
`Tree` a policy returns is the kernels' working storage, not a copy.
"""
import dataclasses
from typing import Any

import torch

from . import _native as nat

ARRAY_FIELDS = nat.TREE_ARRAYS


@dataclasses.dataclass(frozen=True)
class SearchSummary:
    """mctx_dist/_src/tree.py:133-139"""
    visit_counts: Any
    visit_probs: Any
    value: Any
    qvalues: Any


@dataclasses.dataclass(frozen=True)
class Tree:
    """mctx_dist/_src/tree.py:29-91 (value arrays carry a trailing num_atoms axis) or, when
    the root value was given as [B], mctx/_src/tree.py:29-87 (no atom axis)."""
    node_visits: Any            # [B, N] int32
    raw_values: Any             # [B, N(, K)]
    node_values: Any            # [B, N(, K)]
    parents: Any                # [B, N] int32
    action_from_parent: Any     # [B, N] int32
    children_index: Any         # [B, N, A] int32
    children_prior_logits: Any  # [B, N, A]
    children_visits: Any        # [B, N, A] int32
    children_rewards: Any       # [B, N, A]
    children_discounts: Any     # [B, N, A]
    children_values: Any        # [B, N, A(, K)]
    embeddings: Any             # pytree [B, N, ...]
    root_invalid_actions: Any   # [B, A]
    extra_data: Any             # [B, ...]

    ROOT_INDEX = 0
    NO_PARENT = -1
    UNVISITED = -1

    @property
    def num_actions(self):
        return self.children_index.shape[-1]

    @property
    def scalar_values(self):
        return self.node_values.dim() == 2

    @property
    def num_value_atoms(self):
        return 1 if self.scalar_values else self.node_values.shape[-1]

    @property
    def num_simulations(self):
        return self.node_visits.shape[-1] - 1

    def replace(self, **changes):
        return dataclasses.replace(self, **changes)

    def qvalues(self, indices):
        """tree.py:93-100: reward + discount * child value for the given node indices
        (a scalar index applies to every lane, a [B] tensor gives one index per lane)."""
        B = self.node_visits.shape[0]
        if not torch.is_tensor(indices):
            indices = torch.full((B,), int(indices), dtype=torch.long, device=self.node_visits.device)
        idx = indices.long()
        lanes = torch.arange(B, device=idx.device)
        r = self.children_rewards[lanes, idx]
        d = self.children_discounts[lanes, idx]
        v = self.children_values[lanes, idx]
        if self.scalar_values:
            return r + d * v
        return r.unsqueeze(-1) + d.unsqueeze(-1) * v

    def _desc_and_ptrs(self):
        d = nat.SearchDesc()
        d.B, N = self.node_visits.shape
        d.S = N - 1
        d.A = self.num_actions
        d.K = self.num_value_atoms
        d.max_depth = d.S
        t = nat.TreePtrs()
        for name in ARRAY_FIELDS:
            setattr(t, name, nat.ptr(getattr(self, name)))
        return d, t

    def summary(self):
        """tree.py:102-120: root statistics, computed by the az_summary kernel."""
        if self.node_values.dim() not in (2, 3):
            raise ValueError("Input tree is not batched.")
        d, t = self._desc_and_ptrs()
        B, A, K = d.B, d.A, d.K
        dev = self.node_visits.device
        vc = torch.empty((B, A), dtype=torch.float32, device=dev)
        vp = torch.empty((B, A), dtype=torch.float32, device=dev)
        val = torch.empty((B, K), dtype=torch.float32, device=dev)
        q = torch.empty((B, A, K), dtype=torch.float32, device=dev)
        nat.check(nat.lib().az_summary(d, t, nat.ptr(vc), nat.ptr(vp), nat.ptr(val), nat.ptr(q),
                                       nat.stream_ptr()))
        if self.scalar_values:
            val, q = val.squeeze(-1), q.squeeze(-1)
        return SearchSummary(visit_counts=vc, visit_probs=vp, value=val, qvalues=q)


def infer_batch_size(tree):
    """mctx_dist/_src/tree.py:123-128"""
    if tree.node_values.dim() not in (2, 3) or tree.node_visits.dim() != 2:
        raise ValueError("Input tree is not batched.")
    return tree.node_values.shape[0]


def allocate(B, S, A, K, device, scalar_values=False):
    """Uninitialised tree arrays (az_tree_init fills them)."""
    N = S + 1
    f = dict(dtype=torch.float32, device=device)
    i = dict(dtype=torch.int32, device=device)
    kv = () if scalar_values else (K,)
    return dict(
        node_visits=torch.empty((B, N), **i),
        raw_values=torch.empty((B, N) + kv, **f),
        node_values=torch.empty((B, N) + kv, **f),
        parents=torch.empty((B, N), **i),
        action_from_parent=torch.empty((B, N), **i),
        children_index=torch.empty((B, N, A), **i),
        children_prior_logits=torch.empty((B, N, A), **f),
        children_visits=torch.empty((B, N, A), **i),
        children_rewards=torch.empty((B, N, A), **f),
        children_discounts=torch.empty((B, N, A), **f),
        children_values=torch.empty((B, N, A) + kv, **f),
    )
